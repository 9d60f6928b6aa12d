#!/usr/bin/env python3
"""CPU pricing of VERDICT r04 item 5(a): LayerNorm folded into U-ViT's consumer GEMMs.

    python scripts/diag/ln_fold_precision.py [--rows 2048] [--offset 0,2,5,20]

Today:  z = LayerNorm(x) (fp32) -> f16x3 planes of z * 64 -> y = z W^T + b on three fp16 products, fp32 accumulate   (k_layernorm4 + k_hgemm)
Folded: f16x3 planes of the RAW tokens x * 16 (written by the producing GEMM's epilogue) -> acc = x W'^T with W' = diag(gamma) W, then
        y = rstd * acc - rstd * mean * c + d,  c = colsum(W'), d = beta W^T + b                                       (no LayerNorm pass)
The two agree in exact arithmetic; in fp32 the folded form subtracts two numbers of size |mean| * |c| when the row mean is not small
against the row's standard deviation.  This script emulates both on token rows with U-ViT-H/2's widths (D = 1152 -> 3456 / 4608) and
prints their errors against fp64, for rows whose mean is 0, 2, 5, 20 standard deviations (checkpoint tokens carry DC offsets and outlier
channels; seeded random-init tokens do not).  hi / lo planes: fp16(v * S) and fp16(v * S - hi); products exact, sums in fp32."""
import argparse

import torch


def planes(v, scale):
    s = (v * scale).float()
    hi = s.half()
    lo = (s - hi.float()).half()
    return hi.float(), lo.float()


def gemm_f16x3(a, a_scale, w):
    """sum over k of hi*hi + hi*lo + lo*hi in fp32 (weights: per-row power-of-two scale to [2^14, 2^15), as dua_pack_weight_f16x3)."""
    ah, al = planes(a, a_scale)
    e = 14 - torch.floor(torch.log2(w.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)))
    ws = torch.pow(2.0, e)
    wh, wl = planes(w * ws, 1.0)
    acc = ah @ wh.t() + (ah @ wl.t() + al @ wh.t())
    return acc / (a_scale * ws.t())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--offset", default="0,2,5,20")
    a = ap.parse_args()
    g = torch.Generator().manual_seed(5)
    D = 1152
    print("rows", a.rows, "D", D, "(errors relative to the output's column scale: max | rms)")
    for N, name in ((3456, "norm1 -> qkv"), (4608, "norm2 -> fc1")):
        w = torch.randn(N, D, generator=g) / D ** 0.5
        b = torch.randn(N, generator=g) * 0.02
        gamma = 1 + 0.3 * torch.randn(D, generator=g)
        beta = 0.1 * torch.randn(D, generator=g)
        for off in [float(v) for v in a.offset.split(",")]:
            x = torch.randn(a.rows, D, generator=g) * 1.5 + off * 1.5            # row mean = off standard deviations
            x[:, ::97] *= 6.0                                                      # a few outlier channels, as trained tokens have
            x64 = x.double()
            mean, var = x64.mean(1, keepdim=True), x64.var(1, unbiased=False, keepdim=True)
            rstd = (var + 1e-6).rsqrt()
            ref = ((x64 - mean) * rstd * gamma.double() + beta.double()) @ w.double().t() + b.double()
            sc = ref.abs().amax(dim=0, keepdim=True)
            # today: LayerNorm in fp32, planes at scale 64
            z = torch.nn.functional.layer_norm(x, (D,), gamma, beta, eps=1e-6)
            y_cur = gemm_f16x3(z, 64.0, w) + b
            # folded: raw planes at scale 16, fp32 row statistics, fp32 epilogue
            wp = (w.double() * gamma.double()).float()
            c = wp.double().sum(1).float()
            d = (w.double() @ beta.double() + b.double()).float()
            m32, r32 = x.mean(1, keepdim=True), (x.var(1, unbiased=False, keepdim=True) + 1e-6).rsqrt()
            acc = gemm_f16x3(x, 16.0, wp)
            y_fold = r32 * acc - (r32 * m32) * c + d
            # the exact-fp32 engine for scale: LayerNorm fp32, GEMM in fp32
            y_32 = z @ w.t() + b
            e = lambda y: (((y.double() - ref).abs() / sc).max().item(), ((y.double() - ref) / sc).pow(2).mean().sqrt().item())
            (c1, c2), (f1, f2), (g1, g2) = e(y_cur), e(y_fold), e(y_32)
            print(f"{name:14s} row mean = {off:4.0f} sigma   today {c1:.2e} | {c2:.2e}   folded {f1:.2e} | {f2:.2e}   fp32 GEMM {g1:.2e} | {g2:.2e}   "
                  f"folded / today {f1 / c1:5.1f}x")


if __name__ == "__main__":
    main()
