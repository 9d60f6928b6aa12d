#!/usr/bin/env python3
"""2-D Winograd F(4x4, 3x3) with f16x3 operands against the forms the library runs (F(4,3) along y, F(2,3) along y, direct), emulated on
the CPU in torch against an fp64 convolution: is the 2.25-tap form numerically admissible?  (python scripts/diag/wino2d_precision.py)

Emulation: transformed input V = B^T d B in fp32, split into fp16 hi + fp16 remainder at a power-of-two scale; transformed weights
U = G g G^T in fp64, scaled per output channel to fp16's range, split likewise; per frequency hi*hi + lo*hi + hi*lo with fp32 (or fp64)
accumulation over the input channels; inverse transform A^T M A in fp32."""
import math

import torch

torch.manual_seed(0)
N, H, W, Cin, Cout = 2, 32, 32, 256, 64
x = torch.nn.functional.silu(torch.randn(N, Cin, H, W, dtype=torch.float64) * 1.5)          # like GroupNorm + SiLU outputs
w = torch.randn(Cout, Cin, 3, 3, dtype=torch.float64) / math.sqrt(9 * Cin)
ref = torch.nn.functional.conv2d(x, w, padding=1)
x32 = x.float()
BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def split(v, scale):
    s = v.float() * scale
    hi = s.half()
    return hi.float(), (s - hi.float()).half().float(), (s.abs() > 65504).sum().item()


def wino2d(vscale, acc):
    U = torch.einsum("ak,ockl,bl->aboc", G, w, G)                       # [6,6,Cout,Cin]
    sc = U.abs().amax(dim=(0, 1, 3))
    rs = 2.0 ** torch.floor(torch.log2(16384.0 / sc))
    Us = (U * rs.view(1, 1, -1, 1)).float()
    uh = Us.half().float()
    ul = (Us - uh).half().float()
    xp = torch.nn.functional.pad(x32, (1, 1, 1, 1))
    out = torch.zeros(N, Cout, H, W, dtype=torch.float64)
    maxV, sat = 0.0, 0
    for ty in range(H // 4):
        for tx in range(W // 4):
            d = xp[:, :, 4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6]            # [N,Cin,6,6]
            V = torch.einsum("ak,nckl,bl->nabc", BT.float(), d, BT.float())   # fp32
            maxV = max(maxV, V.abs().max().item())
            vh, vl, s_ = split(V, vscale)
            sat += s_
            M = (torch.einsum("nabc,aboc->nabo", vh.to(acc), uh.to(acc)) + torch.einsum("nabc,aboc->nabo", vl.to(acc), uh.to(acc))
                 + torch.einsum("nabc,aboc->nabo", vh.to(acc), ul.to(acc))).float()
            Y = torch.einsum("ra,nabo,sb->nors", AT.float(), M, AT.float()).double() / (vscale * rs.view(1, -1, 1, 1))
            out[:, :, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4] = Y
    return out, maxV, sat


def wino1d(vscale, acc):
    U = torch.einsum("ak,ockx->aocx", G, w)                               # [6,Cout,Cin,3]
    sc = U.abs().amax(dim=(0, 2, 3))
    rs = 2.0 ** torch.floor(torch.log2(16384.0 / sc))
    Us = (U * rs.view(1, -1, 1, 1)).float()
    uh = Us.half().float()
    ul = (Us - uh).half().float()
    xp = torch.nn.functional.pad(x32, (0, 0, 1, 1))
    out = torch.zeros(N, Cout, H, W, dtype=torch.float64)
    c13 = lambda v, u: torch.nn.functional.conv2d(v.unsqueeze(2), u.unsqueeze(2), padding=(0, 1)).squeeze(2)      # noqa: E731
    for j in range(H // 4):
        V = torch.einsum("fk,nckw->nfcw", BT.float(), xp[:, :, 4 * j:4 * j + 6, :])
        vh, vl, _ = split(V, vscale)
        M = torch.stack([(c13(vh[:, f].to(acc), uh[f].to(acc)) + c13(vl[:, f].to(acc), uh[f].to(acc)) + c13(vh[:, f].to(acc), ul[f].to(acc))).float()
                         for f in range(6)], 1)
        out[:, :, 4 * j:4 * j + 4, :] = torch.einsum("rf,nfow->norw", AT.float(), M).double() / (vscale * rs.view(1, -1, 1, 1))
    return out


def direct(acc):
    sc = w.abs().amax(dim=(1, 2, 3))
    rs = 2.0 ** torch.floor(torch.log2(16384.0 / sc))
    ws = (w * rs.view(-1, 1, 1, 1)).float()
    wh = ws.half().float()
    wl = (ws - wh).half().float()
    xh, xl, _ = split(x32, 64.0)
    cv = lambda a, b: torch.nn.functional.conv2d(a.to(acc), b.to(acc), padding=1)      # noqa: E731
    return (cv(xh, wh) + cv(xl, wh) + cv(xh, wl)).double() / (64.0 * rs.view(1, -1, 1, 1))


scale = ref.abs().max().item()
rms = ref.pow(2).mean().sqrt()
err = lambda o: (((o - ref).abs().max() / scale).item(), ((o - ref).pow(2).mean().sqrt() / rms).item())      # noqa: E731
d32 = torch.nn.functional.conv2d(x32, w.float(), padding=1).double()
print("fp32 direct (torch cpu)        max rel %.2e  rms rel %.2e" % err(d32))
print("f16x3 direct, fp32 accumulate  max rel %.2e  rms rel %.2e" % err(direct(torch.float32)))
print("f16x3 F(4,3) along y, scale 8  max rel %.2e  rms rel %.2e" % err(wino1d(8.0, torch.float32)))
for vs in (1.0, 2.0, 4.0):
    for acc in (torch.float32, torch.float64):
        o, mv, sat = wino2d(vs, acc)
        print("f16x3 F(4x4,3x3), V scale %g, %s accumulate  max rel %.2e  rms rel %.2e   max|V| %.0f  saturated %d" % ((vs, str(acc).split(".")[1]) + err(o) + (mv, sat)))
