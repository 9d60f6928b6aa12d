"""KL-VAE decoder of the U-ViT latent models, functional on a state dict (oracle; test infrastructure).

Follows reference ``uvit/autoencoder.py``: ``FrozenAutoencoderKL.decode`` :446-450, ``Decoder.forward`` :380-407
(block enumeration ``Decoder.__init__`` :297-378), ``ResnetBlock.forward`` :117-137 (temb is None),
``AttnBlock.forward`` :169-197, ``Upsample.forward`` :37-41, ``Normalize`` :26-27 (GroupNorm 32, eps 1e-6);
hyper-parameters ``get_model`` :463-476.
"""
import torch
import torch.nn.functional as F

KL_F8 = dict(z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2,
             attn_resolutions=(), embed_dim=4, scale_factor=0.18215)


def _gn(sd, p, x):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def resnet_block(sd, p, x):
    h = F.conv2d(F.silu(_gn(sd, p + ".norm1", x)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(sd, p + ".norm2", h)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".nin_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".nin_shortcut.weight"], sd[p + ".nin_shortcut.bias"])
    return x + h


def attn_block(sd, p, x):
    b, c, hh, ww = x.shape
    h = _gn(sd, p + ".norm", x)
    q, k, v = (F.conv2d(h, sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"]).reshape(b, c, hh * ww) for n in ("q", "k", "v"))
    w = torch.softmax(torch.bmm(q.permute(0, 2, 1), k) * (int(c) ** -0.5), dim=2)
    o = torch.bmm(v, w.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + F.conv2d(o, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])


def decode(sd, cfg, z):
    """z [B, z_channels, h, w] -> image [B, out_ch, 8h, 8w]."""
    nres = len(cfg["ch_mult"])
    z = (1.0 / cfg["scale_factor"]) * z
    z = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    d = "decoder."
    h = F.conv2d(z, sd[d + "conv_in.weight"], sd[d + "conv_in.bias"], padding=1)
    h = resnet_block(sd, d + "mid.block_1", h)
    h = attn_block(sd, d + "mid.attn_1", h)
    h = resnet_block(sd, d + "mid.block_2", h)
    for lvl in reversed(range(nres)):
        for j in range(cfg["num_res_blocks"] + 1):
            h = resnet_block(sd, f"{d}up.{lvl}.block.{j}", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"{d}up.{lvl}.upsample.conv.weight"], sd[f"{d}up.{lvl}.upsample.conv.bias"], padding=1)
    h = F.silu(_gn(sd, d + "norm_out", h))
    return F.conv2d(h, sd[d + "conv_out.weight"], sd[d + "conv_out.bias"], padding=1)
