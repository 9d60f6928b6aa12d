#!/usr/bin/env python3
"""K sweep of the one-tap f16x3 GEMM kernel: time per tile = overhead + K steps * step time.

    python scripts/diag/hgemm_sweep.py [--lib alt.so] [--hg nt[,streamk]] [--M 82560]

Cases: N = 1152 with residual (proj / fc2 shape), N = 3456 plain (qkv), N = 4608 GELU -> planes (fc1).  Prints ms, TF, microseconds per
tile and CU, and the least-squares fit over K (overhead per tile, time per 32-deep K step)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--hg", default="")
ap.add_argument("--M", type=int, default=82560)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--splitk", action="store_true", help="register the stream-K workspace")
a = ap.parse_args()
from diffusion_uncertainty_amd import _lib  # noqa: E402
if a.lib:
    _lib.LIB_PATH = a.lib
from diffusion_uncertainty_amd import ops  # noqa: E402
L = _lib.lib()
if a.hg:
    v = [int(t) for t in a.hg.split(",")]
    assert L.dua_set_hgemm_mode(v[0], v[1] if len(v) > 1 else 1) == 0
dev = torch.device("cuda")
if a.splitk:
    ops.set_split_k(True, dev)
M = a.M
for N, kind in ((1152, "r"), (3456, "p"), (4608, "g")):
    pts = []
    for K in (384, 768, 1152, 2304, 4608):
        x = torch.randn(M, K, device=dev) * ops.F16X3_ACT_SCALE
        hi = x.half()
        pl = ops.SplitPlanes(torch.stack([hi, (x - hi.float()).half()]).contiguous().view(torch.int16), (M, K), "f16x3", ops.F16X3_ACT_SCALE)
        del x, hi
        pk = ops.pack_weight_f16x3(torch.randn(N, K, device=dev) / K ** 0.5)
        b = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev) if kind == "r" else None
        if kind == "g":
            fn = lambda: ops.gemm_bf16x3_presplit(pl, pk, N, bias=b, act_out=2, split_out=True)  # noqa: E731
        else:
            fn = lambda: ops.gemm_bf16x3_presplit(pl, pk, N, bias=b, residual=res)  # noqa: E731
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        pts.append((K // 32, ms))
        print(f"N={N} [{kind}] K={K:5d}: {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:6.1f} TF", flush=True)
        del pl, pk
    n = len(pts)
    sx, sy = sum(p[0] for p in pts), sum(p[1] for p in pts)
    sxx, sxy = sum(p[0] ** 2 for p in pts), sum(p[0] * p[1] for p in pts)
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    icpt = (sy - slope * sx) / n
    print(f"  fit N={N} [{kind}]: launch = {icpt * 1e3:.1f} us + {slope * 1e3:.2f} us per K step; overhead = {icpt / slope:.1f} K steps; "
          f"steady rate {2.0 * M * N * 32 / slope / 1e9:.0f} TF")
