#!/usr/bin/env python3
"""Diagnostic: where the several-workgroups-per-sample quantile kernel (csrc/sampler.hip: k_quantile_mask_multi) spends its time.

    python scripts/diag/quantile_diag.py --build      # here in the container: -DQM_DIAG library in scripts/diag/build/
    python scripts/diag/quantile_diag.py              # on the GPU box: per-phase times (100-MHz counter, tid 0 of every workgroup)

The diag library stamps phases into the tail of each sample's candidate area; results stay correct."""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "diffusion-uncertainty_amd")
DIAG = os.path.join(ROOT, "scripts", "diag", "build")
VARIANTS = {"stamps": [], "plain_store": ["-DQM_PLAIN_STORE"]}          # name -> extra flags (all with -DQM_DIAG; results stay correct)
PHASES = ["load+keys", "sample bracket", "count+pack", "publish", "meet", "select", "mask"]


def lib_of(v):
    return os.path.join(DIAG, f"libdua_hip_qmdiag_{v}.so")


def build():
    os.makedirs(DIAG, exist_ok=True)
    csrc = os.path.join(PKG, "csrc")
    objs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".o") and f != "sampler.o"]
    for v, flags in VARIANTS.items():
        o = os.path.join(DIAG, f"sampler.qmdiag_{v}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-ffp-contract=off",
                               "-DQM_DIAG", *flags, "-c", os.path.join(csrc, "sampler.hip"), "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [o, "-o", lib_of(v)])
        print("built", lib_of(v))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--shape", default="128,3,128,128")
    ap.add_argument("--p", type=float, default=0.9)
    ap.add_argument("--variant", default="")
    a = ap.parse_args()
    if a.build:
        return build()
    if not a.variant:                        # one process per library
        for v in VARIANTS:
            subprocess.call([sys.executable, os.path.abspath(__file__), "--variant", v, "--shape", a.shape, "--p", str(a.p)])
        return
    LIB = lib_of(a.variant)
    print(f"======== {a.variant} ({' '.join(VARIANTS[a.variant]) or 'default'})", flush=True)
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    os.environ["DUA_HIP_LIB"] = LIB
    L = ctypes.CDLL(LIB)
    L.dua_quantile_diag_workspace.restype = ctypes.c_void_p
    L.dua_quantile_diag_workspace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.dua_quantile_mask.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    shape = tuple(int(t) for t in a.shape.split(","))
    B, n = shape[0], int(np.prod(shape[1:]))
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77)
    base = torch.randn(shape, generator=g)
    pos = np.float32(a.p) * np.float32(n - 1)
    lo = int(np.floor(pos))
    hi = min(lo + 1, n - 1) if pos > lo else lo
    w = float(np.float32(pos) - np.float32(lo))
    CAP, CTR = 12288, 16
    for name, u in (("chi2", base ** 2), ("signed", base)):
        ud = u.contiguous().to(dev)
        mk, qq = torch.empty(shape, device=dev, dtype=torch.uint8), torch.empty(B, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        args = (ctypes.c_void_p(ud.data_ptr()), B, n, lo, hi, w, ctypes.c_void_p(mk.data_ptr()), ctypes.c_void_p(qq.data_ptr()), st)
        for _ in range(5):
            assert L.dua_quantile_mask(*args) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            L.dua_quantile_mask(*args)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        ws = L.dua_quantile_diag_workspace(st, B)
        words = B * (CTR + CAP)
        host = np.empty(words, dtype=np.uint32)
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(ws), ctypes.c_size_t(words * 4), 2)
        host = host.reshape(B, CTR + CAP)
        ctr = host[:, :CTR]
        stamps = host[:, CTR + CAP - 256:].copy().view(np.uint64).reshape(B, 4, 32)[:, :2]          # [B, G, 32]
        t = stamps[:, :, :8].astype(np.int64)
        t0 = t[:, :, 0].min()
        d = np.diff(t, axis=2) / 100.0                                                                # us per phase
        print(f"---- {name} {shape}: {us:.1f} us per call; counters after the last launch (all zero expected): {ctr[:, :4].max()}; "
              f"bracket hit in {int(stamps[:, 0, 8].sum())} of {B} samples")
        print("   workgroup start after the first: median %.1f us, max %.1f us; end: median %.1f, max %.1f" %
              (np.median(t[:, :, 0] - t0) / 100, (t[:, :, 0] - t0).max() / 100, np.median(t[:, :, 7] - t0) / 100, (t[:, :, 7] - t0).max() / 100))
        for i, ph in enumerate(PHASES):
            print(f"   {ph:16s} median {np.median(d[:, :, i]):6.2f} us   max {d[:, :, i].max():6.2f} us")
        st = stamps.astype(np.int64)

        def seg(a_, b_):
            return np.median(st[:, :, b_] - st[:, :, a_]) / 100.0
        print("   sample bracket: min/max %.2f | zero + atomics %.2f | scan + pick %.2f | rest %.2f" % (seg(1, 9), seg(9, 10), seg(10, 11), seg(11, 2)))
        print("   select: candidate load %.2f | min/max %.2f | zero + atomics %.2f | scan + pick %.2f | pack %.2f | rank %.2f | successor + q %.2f" %
              (seg(5, 15) + seg(15, 16), seg(16, 17), seg(17, 18), seg(18, 19), seg(19, 20), seg(20, 21), seg(21, 6)))


if __name__ == "__main__":
    main()
