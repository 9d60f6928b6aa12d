#!/usr/bin/env python3
"""Diagnostic: what a tile of the one-tap f16x3 GEMM kernel (csrc/gemm_f16x3.hip) pays outside its steady K loop.

    python scripts/hgemm_diag.py --build      # here in the container: timing-only libraries in scripts/diag/build/
    python scripts/hgemm_diag.py              # on the GPU box: one K sweep (scripts/diag/hgemm_sweep.py) per library

Variants (-DHG_DIAG bits: 1 no epilogue, 2 no output stores, 4 no prefetch of the next item's first stages, 8 no GELU) produce WRONG
results; they exist only outside the product package."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "diffusion-uncertainty_amd")
DIAG = os.path.join(ROOT, "scripts", "diag", "build")
VARIANTS = (1, 2, 4, 8)


def lib_of(v):
    return os.path.join(DIAG, f"libdua_hip_hgdiag{v}.so")


def build():
    os.makedirs(DIAG, exist_ok=True)
    csrc = os.path.join(PKG, "csrc")
    objs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".o") and f != "gemm_f16x3.o"]
    for v in VARIANTS:
        o = os.path.join(DIAG, f"gemm_f16x3.hgdiag{v}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result",
                               f"-DHG_DIAG={v}", "-c", os.path.join(csrc, "gemm_f16x3.hip"), "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [o, "-o", lib_of(v)])
        print("built", lib_of(v))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--hg", default="")
    ap.add_argument("--variants", default=",".join(str(v) for v in (0,) + VARIANTS))
    a = ap.parse_args()
    if a.build:
        return build()
    sweep = os.path.join(ROOT, "scripts", "diag", "hgemm_sweep.py")
    for v in [int(t) for t in a.variants.split(",")]:
        cmd = [sys.executable, sweep] + (["--lib", lib_of(v)] if v else []) + (["--hg", a.hg] if a.hg else [])
        out = subprocess.run(cmd, capture_output=True, text=True)
        print(f"---- HG_DIAG={v}")
        print("\n".join(l for l in out.stdout.splitlines() if l.startswith("N=") or l.startswith("  fit")), flush=True)
        if out.returncode:
            print(out.stderr[-2000:])


if __name__ == "__main__":
    main()
