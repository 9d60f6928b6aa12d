"""Latent UNet of Stable Diffusion 1.5 (oracle; test infrastructure).  PARITY UNPINNED.

The reference reaches this network only through the un-vendored dependency ``diffusers==0.31.0``
(``pyproject.toml:76``): ``StableDiffusionPipelineUncertainty.__call__`` calls ``self.unet(latent_model_input, t,
encoder_hidden_states=prompt_embeds, ...)`` (pipeline_uncertainty/pipeline_stable_diffusion_uncertainty_guided.py:747-753)
on ``UNet2DConditionModel.from_pretrained('runwayml/stable-diffusion-v1-5')`` (scripts/
generate_with_uncertainty_threshold_stable_diffusion.py:25).  diffusers is not installed here and /root/reference holds no
vector for it, so this file restates the published architecture of that class for the SD-1.5 configuration
(``unet/config.json`` of the checkpoint: block_out_channels (320, 640, 1280, 1280), layers_per_block 2,
CrossAttnDownBlock2D x3 + DownBlock2D, UNetMidBlock2DCrossAttn, UpBlock2D + CrossAttnUpBlock2D x3, attention_head_dim 8
(= number of heads), cross_attention_dim 768, norm_num_groups 32, norm_eps 1e-5, act_fn silu, flip_sin_to_cos True,
freq_shift 0, use_linear_projection False, downsample_padding 1) with diffusers' state-dict key names, so that a real
checkpoint would load strictly.  It is the checker of the HIP implementation (tests, smoke), nothing else.
"""
import math

import torch
import torch.nn.functional as F

SD15 = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
            down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
            up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
            attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5)


def state_shapes(cfg):
    """name -> shape, in diffusers' UNet2DConditionModel key layout."""
    boc, L, ctx = cfg["block_out_channels"], cfg["layers_per_block"], cfg["cross_attention_dim"]
    ted = boc[0] * 4
    s = {"conv_in.weight": (boc[0], cfg["in_channels"], 3, 3), "conv_in.bias": (boc[0],),
         "time_embedding.linear_1.weight": (ted, boc[0]), "time_embedding.linear_1.bias": (ted,),
         "time_embedding.linear_2.weight": (ted, ted), "time_embedding.linear_2.bias": (ted,)}

    def resnet(p, cin, cout):
        s.update({f"{p}.norm1.weight": (cin,), f"{p}.norm1.bias": (cin,), f"{p}.conv1.weight": (cout, cin, 3, 3),
                  f"{p}.conv1.bias": (cout,), f"{p}.time_emb_proj.weight": (cout, ted), f"{p}.time_emb_proj.bias": (cout,),
                  f"{p}.norm2.weight": (cout,), f"{p}.norm2.bias": (cout,), f"{p}.conv2.weight": (cout, cout, 3, 3),
                  f"{p}.conv2.bias": (cout,)})
        if cin != cout:
            s.update({f"{p}.conv_shortcut.weight": (cout, cin, 1, 1), f"{p}.conv_shortcut.bias": (cout,)})

    def attn(p, c):
        t = f"{p}.transformer_blocks.0"
        s.update({f"{p}.norm.weight": (c,), f"{p}.norm.bias": (c,), f"{p}.proj_in.weight": (c, c, 1, 1), f"{p}.proj_in.bias": (c,),
                  f"{p}.proj_out.weight": (c, c, 1, 1), f"{p}.proj_out.bias": (c,)})
        for n in ("norm1", "norm2", "norm3"):
            s.update({f"{t}.{n}.weight": (c,), f"{t}.{n}.bias": (c,)})
        for a, kd in (("attn1", c), ("attn2", ctx)):
            s.update({f"{t}.{a}.to_q.weight": (c, c), f"{t}.{a}.to_k.weight": (c, kd), f"{t}.{a}.to_v.weight": (c, kd),
                      f"{t}.{a}.to_out.0.weight": (c, c), f"{t}.{a}.to_out.0.bias": (c,)})
        s.update({f"{t}.ff.net.0.proj.weight": (8 * c, c), f"{t}.ff.net.0.proj.bias": (8 * c,),
                  f"{t}.ff.net.2.weight": (c, 4 * c), f"{t}.ff.net.2.bias": (c,)})

    ch = boc[0]
    for i, (bt, co) in enumerate(zip(cfg["down_block_types"], boc)):
        for j in range(L):
            resnet(f"down_blocks.{i}.resnets.{j}", ch, co)
            ch = co
            if bt.startswith("CrossAttn"):
                attn(f"down_blocks.{i}.attentions.{j}", co)
        if i != len(boc) - 1:
            s.update({f"down_blocks.{i}.downsamplers.0.conv.weight": (co, co, 3, 3), f"down_blocks.{i}.downsamplers.0.conv.bias": (co,)})
    resnet("mid_block.resnets.0", ch, ch)
    attn("mid_block.attentions.0", ch)
    resnet("mid_block.resnets.1", ch, ch)
    rev = list(reversed(boc))
    for i, bt in enumerate(cfg["up_block_types"]):
        co, prev = rev[i], rev[max(i - 1, 0)] if i else rev[0]
        cin_skip = rev[min(i + 1, len(boc) - 1)]
        for j in range(L + 1):
            skip = cin_skip if j == L else co
            rin = prev if j == 0 else co
            resnet(f"up_blocks.{i}.resnets.{j}", rin + skip, co)
            if bt.startswith("CrossAttn"):
                attn(f"up_blocks.{i}.attentions.{j}", co)
        if i != len(boc) - 1:
            s.update({f"up_blocks.{i}.upsamplers.0.conv.weight": (co, co, 3, 3), f"up_blocks.{i}.upsamplers.0.conv.bias": (co,)})
    s.update({"conv_norm_out.weight": (boc[0],), "conv_norm_out.bias": (boc[0],),
              "conv_out.weight": (cfg["out_channels"], boc[0], 3, 3), "conv_out.bias": (cfg["out_channels"],)})
    return s


def timestep_embedding(t, dim):
    """get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    a = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


def _resnet(sd, p, x, emb, G, eps):
    h = F.conv2d(F.silu(F.group_norm(x, G, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps)), sd[p + ".conv1.weight"],
                 sd[p + ".conv1.bias"], padding=1)
    h = h + F.linear(F.silu(emb), sd[p + ".time_emb_proj.weight"], sd[p + ".time_emb_proj.bias"])[:, :, None, None]
    h = F.conv2d(F.silu(F.group_norm(h, G, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps)), sd[p + ".conv2.weight"],
                 sd[p + ".conv2.bias"], padding=1)
    if p + ".conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def _mha(sd, p, x, ctx, heads):
    q, k, v = F.linear(x, sd[p + ".to_q.weight"]), F.linear(ctx, sd[p + ".to_k.weight"]), F.linear(ctx, sd[p + ".to_v.weight"])
    B, T, C = q.shape
    sp = lambda z: z.reshape(B, z.shape[1], heads, C // heads).transpose(1, 2)
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, T, C)
    return F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])


def _transformer(sd, p, x, ctx, heads, G):
    B, C, H, W = x.shape
    h = F.conv2d(F.group_norm(x, G, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6), sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    t = p + ".transformer_blocks.0"
    ln = lambda z, n: F.layer_norm(z, (C,), sd[f"{t}.{n}.weight"], sd[f"{t}.{n}.bias"], 1e-5)
    n1 = ln(h, "norm1")
    h = h + _mha(sd, t + ".attn1", n1, n1, heads)
    h = h + _mha(sd, t + ".attn2", ln(h, "norm2"), ctx, heads)
    g = F.linear(ln(h, "norm3"), sd[t + ".ff.net.0.proj.weight"], sd[t + ".ff.net.0.proj.bias"])
    a, gate = g.chunk(2, dim=-1)
    h = h + F.linear(a * F.gelu(gate), sd[t + ".ff.net.2.weight"], sd[t + ".ff.net.2.bias"])
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return F.conv2d(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"]) + x


def forward_autograd(sd, cfg, sample, timesteps, encoder_hidden_states):
    """UNet2DConditionModel.forward(sample [B,4,h,w], timestep [B], encoder_hidden_states [B,77,ctx]) -> [B,4,h,w], recorded by
    torch autograd when the caller enables it (gradient-mode guidance, reference uncertainty_guidance.py:83-108)."""
    boc, L, G, eps, heads = (cfg["block_out_channels"], cfg["layers_per_block"], cfg["norm_num_groups"], cfg["norm_eps"],
                             cfg["attention_head_dim"])
    emb = F.linear(timestep_embedding(timesteps, boc[0]).to(sample.dtype), sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    emb = F.linear(F.silu(emb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    h = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    res = [h]
    for i, bt in enumerate(cfg["down_block_types"]):
        for j in range(L):
            h = _resnet(sd, f"down_blocks.{i}.resnets.{j}", h, emb, G, eps)
            if bt.startswith("CrossAttn"):
                h = _transformer(sd, f"down_blocks.{i}.attentions.{j}", h, encoder_hidden_states, heads, G)
            res.append(h)
        if i != len(boc) - 1:
            h = F.conv2d(h, sd[f"down_blocks.{i}.downsamplers.0.conv.weight"], sd[f"down_blocks.{i}.downsamplers.0.conv.bias"], stride=2, padding=1)
            res.append(h)
    h = _resnet(sd, "mid_block.resnets.0", h, emb, G, eps)
    h = _transformer(sd, "mid_block.attentions.0", h, encoder_hidden_states, heads, G)
    h = _resnet(sd, "mid_block.resnets.1", h, emb, G, eps)
    for i, bt in enumerate(cfg["up_block_types"]):
        for j in range(L + 1):
            h = _resnet(sd, f"up_blocks.{i}.resnets.{j}", torch.cat([h, res.pop()], dim=1), emb, G, eps)
            if bt.startswith("CrossAttn"):
                h = _transformer(sd, f"up_blocks.{i}.attentions.{j}", h, encoder_hidden_states, heads, G)
        if i != len(boc) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"up_blocks.{i}.upsamplers.0.conv.weight"], sd[f"up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    h = F.silu(F.group_norm(h, G, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps))
    return F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


@torch.no_grad()
def forward(sd, cfg, sample, timesteps, encoder_hidden_states):
    return forward_autograd(sd, cfg, sample, timesteps, encoder_hidden_states)


def input_vjp(sd, cfg, sample, timesteps, encoder_hidden_states, grad_out):
    """(out, d<grad_out, out>/d sample) by torch autograd: the checker of UNet2DConditionModel.backward_sample."""
    x = sample.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        out = forward_autograd(sd, cfg, x, timesteps, encoder_hidden_states)
        (out * grad_out).sum().backward()
    return out.detach(), x.grad.detach()
