#!/usr/bin/env python3
"""Diagnostic: shader-clock stamps of the tile phases of the LDS-DMA conv kernel (build with --build here in the
container, run on the GPU box).  Uses a separate library built with -DDUA_CONV_TIMING; the product .so has no stamps.

    python scripts/conv_tile_timing.py --build
    python scripts/conv_tile_timing.py --shape 128,256,256 [--res]
"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "diffusion-uncertainty_amd")
DIAG = os.path.join(ROOT, "scripts", "diag", "build")          # diagnostic builds stay out of the product package directory
TLIB = os.path.join(DIAG, "libdua_hip_timing.so")


def build(extra_flags=(), out=TLIB):
    objs = []
    os.makedirs(DIAG, exist_ok=True)
    for src, extra in (("sampler.hip", ["-ffp-contract=off"]), ("denoiser.hip", []), ("mfma_gemm.hip", []),
                       ("mfma_bf16x3.hip", ["-DDUA_CONV_TIMING"] + list(extra_flags)), ("attention.hip", [])):
        o = os.path.join(PKG, "csrc", src.replace(".hip", ".o")) if "bf16x3" not in src else os.path.join(DIAG, "mfma_bf16x3.timing.o")
        if "bf16x3" in src or not os.path.exists(o):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"]
                                  + extra + ["-c", os.path.join(PKG, "csrc", src), "-o", o])
        objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    print("built", out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--flags", default="", help="extra hipcc flags for --build, e.g. -DDUA_SCHED=2")
    ap.add_argument("--lib", default=TLIB)
    ap.add_argument("--engine", type=int, default=1, help="3 = zero-record descriptors (prices the memory path)")
    ap.add_argument("--shape", default="128,256,256")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--res", action="store_true")
    ap.add_argument("--fmt", default="bf16x3", choices=["bf16x3", "f16x3"])
    a = ap.parse_args()
    if a.build:
        return build(a.flags.split(), a.lib)
    import numpy as np
    import torch
    from diffusion_uncertainty_amd import _lib
    _lib.LIB_PATH = a.lib
    from diffusion_uncertainty_amd import ops
    L = _lib.lib()
    L.dua_set_dma_conv(a.engine)
    H, Cin, Cout = (int(v) for v in a.shape.split(","))
    N, dev = a.batch, "cuda"
    x = torch.randn(N, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02
    if a.fmt == "bf16x3":
        hi = x.to(torch.bfloat16)
        lo = (x - hi.float()).to(torch.bfloat16)
        xs = ops.SplitPlanes(torch.stack([hi, lo]).view(torch.int16), (N, H, H, Cin))
        planes = ops.pack_weight_bf16x3(w)
    else:
        st = torch.tensor([0.0, 1.0], device=dev).repeat(N * 32, 1).contiguous()
        xs = ops.groupnorm_apply(x, st, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), act=0, split="f16x3")
        planes = ops.pack_weight_f16x3(w)
    bias = torch.randn(Cout, device=dev)
    r = torch.randn(N, H, H, Cout, device=dev) if a.res else None
    for _ in range(3):
        ops.conv3x3_bf16x3(xs, planes, bias, Cout, residual=r, colsum=True)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (256 * 8 * 4))()
    L.dua_debug_tile_times.argtypes = [ctypes.c_void_p]
    assert L.dua_debug_tile_times(buf) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 4).astype(np.int64)
    pro, loop, epi = t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2]
    gap = t[:, 1:, 0] - t[:, :-1, 3]
    print(f"shape {a.shape} fmt={a.fmt} res={int(a.res)} nk={9 * Cin // 32}: median cycles per tile over 256 workgroups")
    for i in range(8):
        print(f" tile {i}: prologue {int(np.median(pro[:, i])):7d}  K loop {int(np.median(loop[:, i])):8d}  epilogue {int(np.median(epi[:, i])):7d}"
              + (f"  gap-to-next {int(np.median(gap[:, i])):6d}" if i < 7 else ""))
    # the counter bases differ between XCDs: spread of the tile starts inside each XCD (workgroup b runs on XCD b % 8)
    for i in (0, 3, 7):
        sp = [int(t[x::8, i, 0].max() - t[x::8, i, 0].min()) for x in range(8)]
        print(f" tile {i}: start-time spread inside an XCD (cycles): median {int(np.median(sp))}, max {max(sp)}")
    tot = t[:, 7, 3] - t[:, 0, 0]
    print(f" 8 tiles: {int(np.median(tot))} cycles; K loop share {np.median(loop.sum(1) / tot):.3f}, epilogue {np.median(epi.sum(1) / tot):.3f}, "
          f"prologue {np.median(pro.sum(1) / tot):.3f}")


if __name__ == "__main__":
    main()
