"""ADM (guided-diffusion) UNet forward, functional on a state dict (oracle; test infrastructure).

Follows reference ``guided_diffusion/unet_openai.py``: ``UNetModel.__init__``
:481-617 (block enumeration), ``forward`` :635-665, ``ResBlock._forward``
:237-257, ``AttentionBlock._forward`` :300-306, ``QKVAttentionLegacy`` :338-355,
``QKVAttention`` :371-390; ``guided_diffusion/nn.py`` ``GroupNorm32`` :17-19,
``timestep_embedding`` :103-121.  Hyper-parameters of the two ImageNet models:
``init_model.py`` :21 (ADM-128) and :45-47 (ADM-64).

Written against plain ``torch.nn.functional`` on CPU fp32; the state-dict keys
are the OpenAI checkpoint keys.
"""
import math

import torch
import torch.nn.functional as F

ADM128 = dict(image_size=128, in_channels=3, model_channels=256, out_channels=6, num_res_blocks=2,
              attention_resolutions=(4, 8, 16), dropout=0.0, channel_mult=(1, 1, 2, 3, 4),
              num_classes=1000, num_heads=4, num_head_channels=-1, num_heads_upsample=4,
              use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=False)
ADM64 = dict(image_size=64, in_channels=3, model_channels=192, out_channels=6, num_res_blocks=3,
             attention_resolutions=(2, 4, 8), dropout=0.1, channel_mult=(1, 2, 3, 4),
             num_classes=1000, num_heads=4, num_head_channels=64, num_heads_upsample=4,
             use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True)


def layout(cfg):
    """Enumerate the blocks of UNetModel.__init__ :481-617 as flat tuples.

    Returns (inp, mid, out): lists of lists of ('conv'|'res'|'attn', key_prefix, ...).
    Only the resblock_updown=True / use_scale_shift_norm=True family (both
    ImageNet models) is described.
    """
    mc, mult, nrb = cfg["model_channels"], cfg["channel_mult"], cfg["num_res_blocks"]
    nh, nhc = cfg["num_heads"], cfg["num_head_channels"]
    nhu = cfg["num_heads_upsample"] if cfg.get("num_heads_upsample", -1) != -1 else nh

    def heads(ch, n):
        return n if nhc == -1 else ch // nhc

    ch = int(mult[0] * mc)
    inp = [[("conv", "input_blocks.0.0", cfg["in_channels"], ch)]]
    chans, ds = [ch], 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            i = len(inp)
            blk = [("res", f"input_blocks.{i}.0", ch, int(m * mc), 0)]
            ch = int(m * mc)
            if ds in cfg["attention_resolutions"]:
                blk.append(("attn", f"input_blocks.{i}.1", ch, heads(ch, nh)))
            inp.append(blk)
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append([("res", f"input_blocks.{len(inp)}.0", ch, ch, -1)])
            chans.append(ch)
            ds *= 2
    mid = [("res", "middle_block.0", ch, ch, 0), ("attn", "middle_block.1", ch, heads(ch, nh)),
           ("res", "middle_block.2", ch, ch, 0)]
    out = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            j = len(out)
            blk = [("res", f"output_blocks.{j}.0", ch + ich, int(mc * m), 0)]
            ch = int(mc * m)
            k = 1
            if ds in cfg["attention_resolutions"]:
                blk.append(("attn", f"output_blocks.{j}.{k}", ch, heads(ch, nhu)))
                k += 1
            if level and i == nrb:
                blk.append(("res", f"output_blocks.{j}.{k}", ch, ch, +1))
                ds //= 2
            out.append(blk)
    return inp, mid, out


def timestep_embedding(t, dim, max_period=10000):
    """nn.py:103-121."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn(sd, p, x):
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def resblock(sd, p, x, emb, updown, drop_p=0.0, train=False):
    """ResBlock._forward :237-257 (scale-shift-norm variant :249-253)."""
    h = F.silu(_gn(sd, p + ".in_layers.0", x))
    if updown < 0:      # Downsample(use_conv=False) = 2x2 average pool on both paths :194-196
        h, x = F.avg_pool2d(h, 2, 2), F.avg_pool2d(x, 2, 2)
    elif updown > 0:    # Upsample(use_conv=False) = nearest x2 :191-193
        h, x = F.interpolate(h, scale_factor=2, mode="nearest"), F.interpolate(x, scale_factor=2, mode="nearest")
    h = F.conv2d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])[:, :, None, None]
    scale, shift = torch.chunk(e, 2, dim=1)
    h = _gn(sd, p + ".out_layers.0", h) * (1 + scale) + shift
    h = F.dropout(F.silu(h), drop_p, training=train)                 # nn.Dropout at out_layers[2] :210
    h = F.conv2d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if (p + ".skip_connection.weight") in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def qkv_attention(qkv, n_heads, new_order):
    """QKVAttentionLegacy :338-355 / QKVAttention :371-390; qkv [B, 3*H*ch, T]."""
    bs, width, T = qkv.shape
    ch = width // (3 * n_heads)
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q, k, v = (z.reshape(bs * n_heads, ch, T) for z in (q, k, v))
    else:
        q, k, v = qkv.reshape(bs * n_heads, 3 * ch, T).split(ch, dim=1)
    s = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * s, k * s)
    w = torch.softmax(w.float(), dim=-1)
    return torch.einsum("bts,bcs->bct", w, v).reshape(bs, -1, T)


def attnblock(sd, p, x, n_heads, new_order):
    """AttentionBlock._forward :300-306."""
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn(sd, p + ".norm", xf), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    a = qkv_attention(qkv, n_heads, new_order)
    a = F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (xf + a).reshape(b, c, hh, ww)


def forward(sd, cfg, x, t, y, train=False, drop_p=None, output_hook=None):
    """UNetModel.forward :635-665 -> [B, out_channels, H, W].

    ``output_hook(name, h) -> h`` is applied to the output of every ResBlock / AttentionBlock / input conv, ``name`` being
    the torch module name (``input_blocks.8.0`` ...): the forward hooks of scheduling_ddim_uncertainty.py:520-529."""
    assert (y is not None) == (cfg.get("num_classes") is not None), \
        "must specify y if and only if the model is class-conditional"
    drop_p = cfg.get("dropout", 0.0) if drop_p is None else drop_p
    new_order = cfg.get("use_new_attention_order", False)
    inp, mid, out = layout(cfg)
    emb = F.linear(timestep_embedding(t, cfg["model_channels"]), sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    if y is not None:
        assert y.shape == (x.shape[0],)
        emb = emb + sd["label_emb.weight"][y]

    def run(blk, h):
        for op in blk:
            if op[0] == "conv":
                h = F.conv2d(h, sd[op[1] + ".weight"], sd[op[1] + ".bias"], padding=1)
            elif op[0] == "res":
                h = resblock(sd, op[1], h, emb, op[4], drop_p, train)
            else:
                h = attnblock(sd, op[1], h, op[3], new_order)
            if output_hook is not None:
                h = output_hook(op[1], h)
        return h

    h, hs = x.float(), []
    for blk in inp:
        h = run(blk, h)
        hs.append(h)
    h = run(mid, h)
    for blk in out:
        h = run(blk, torch.cat([h, hs.pop()], dim=1))
    h = F.silu(_gn(sd, "out.0", h))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)
