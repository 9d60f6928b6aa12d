#!/usr/bin/env python3
"""Diagnostic: what the K loop of the Winograd conv kernel (csrc/conv_wino_f16x3.hip) pays above its matrix-core floor.

    python scripts/wconv_diag.py --build            # here in the container: timing-only libraries in scripts/diag/build/
    python scripts/wconv_diag.py [--only 64x512x512,128x256x256]     # on the GPU box: one bench_conv_mix run per library

The variants drop pieces of the loop (-DWCONV_DIAG bits: 1 barrier, 2 weight DMA, 4 activation DMA, 8 fragment reads; 16 = activations from a cache-resident
set of tiles, 32 = the f = 2 products skipped by their waves, as for an all-zero V2; 64 = no epilogue) and produce
WRONG results; they exist only outside the product package."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "diffusion-uncertainty_amd")
DIAG = os.path.join(ROOT, "scripts", "diag", "build")
VARIANTS = (1, 2, 4, 6, 8, 15, 16, 32, 64)


def lib_of(v):
    return os.path.join(DIAG, f"libdua_hip_wdiag{v}.so")


def build():
    os.makedirs(DIAG, exist_ok=True)
    csrc = os.path.join(PKG, "csrc")
    objs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".o") and f != "conv_wino_f16x3.o"]
    for v in VARIANTS:
        o = os.path.join(DIAG, f"conv_wino_f16x3.wdiag{v}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result",
                               f"-DWCONV_DIAG={v}", "-c", os.path.join(csrc, "conv_wino_f16x3.hip"), "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [o, "-o", lib_of(v)])
        print("built", lib_of(v))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--only", default="128x256x256,64x512x512,16x768x768")
    a = ap.parse_args()
    if a.build:
        return build()
    mix = os.path.join(ROOT, "scripts", "bench_conv_mix.py")
    for v in (0,) + VARIANTS:
        cmd = [sys.executable, mix, "--fmt", "f16x3w", "--only", a.only] + (["--lib", lib_of(v)] if v else [])
        out = subprocess.run(cmd, capture_output=True, text=True).stdout
        print(f"---- WCONV_DIAG={v}")
        print("\n".join(l for l in out.splitlines() if l.startswith("H=") or l.startswith("mix")), flush=True)


if __name__ == "__main__":
    main()
