"""U-ViT forward, functional on a state dict (oracle; test infrastructure).

Follows reference ``uvit/uvit.py``: ``UViT.forward`` :203-231, ``Block._forward``
:115-120, ``Attention.forward`` :66-92 (fp32 SDPA branch :69-73), ``patchify``/
``unpatchify`` :41-51, ``PatchEmbed`` :123-135; ``uvit/timm.py`` ``Mlp`` :96-111.
U-ViT-H/2 hyper-parameters: ``uvit/load_pretrained_models.py`` :34-36.
"""
import math

import torch
import torch.nn.functional as F

from .adm import timestep_embedding

UVIT_H2_256 = dict(img_size=32, patch_size=2, in_chans=4, embed_dim=1152, depth=28, num_heads=16,
                   num_classes=1001, mlp_time_embed=False, conv=False, qkv_bias=False)


def attention(sd, p, x, num_heads):
    B, L, C = x.shape
    qkv = F.linear(x, sd[p + ".qkv.weight"], sd.get(p + ".qkv.bias"))
    qkv = qkv.reshape(B, L, 3, num_heads, C // num_heads).permute(2, 0, 3, 1, 4).float()   # K B H L D
    q, k, v = qkv[0], qkv[1], qkv[2]
    w = torch.softmax((q @ k.transpose(-2, -1)) * (C // num_heads) ** -0.5, dim=-1)
    o = (w @ v).permute(0, 2, 1, 3).reshape(B, L, C)
    return F.linear(o, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def block(sd, p, x, num_heads, skip=None):
    if skip is not None:
        x = F.linear(torch.cat([x, skip], dim=-1), sd[p + ".skip_linear.weight"], sd[p + ".skip_linear.bias"])
    D = x.shape[-1]
    x = x + attention(sd, p + ".attn", F.layer_norm(x, (D,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"]), num_heads)
    h = F.layer_norm(x, (D,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
    h = F.gelu(F.linear(h, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"]))
    return x + F.linear(h, sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])


def forward(sd, cfg, x, t, y=None):
    P, D, H = cfg["patch_size"], cfg["embed_dim"], cfg["num_heads"]
    B, C, hh, ww = x.shape
    tok = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=P).flatten(2).transpose(1, 2)
    L = tok.shape[1]
    tt = timestep_embedding(t, D)
    if cfg.get("mlp_time_embed"):
        tt = F.linear(F.silu(F.linear(tt, sd["time_embed.0.weight"], sd["time_embed.0.bias"])),
                      sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    tok = torch.cat([tt[:, None], tok], dim=1)
    extras = 1
    if y is not None:
        tok = torch.cat([sd["label_emb.weight"][y][:, None], tok], dim=1)
        extras = 2
    tok = tok + sd["pos_embed"]
    half = cfg["depth"] // 2
    skips = []
    for i in range(half):
        tok = block(sd, f"in_blocks.{i}", tok, H)
        skips.append(tok)
    tok = block(sd, "mid_block", tok, H)
    for i in range(half):
        tok = block(sd, f"out_blocks.{i}", tok, H, skips.pop())
    tok = F.layer_norm(tok, (D,), sd["norm.weight"], sd["norm.bias"])
    tok = F.linear(tok, sd["decoder_pred.weight"], sd["decoder_pred.bias"])[:, extras:]
    g = int(math.isqrt(L))
    # unpatchify 'B (h w) (p1 p2 C) -> B C (h p1) (w p2)'  :46-51
    img = tok.reshape(B, g, g, P, P, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, g * P, g * P)
    if cfg.get("conv"):
        img = F.conv2d(img, sd["final_layer.weight"], sd["final_layer.bias"], padding=1)
    return img
