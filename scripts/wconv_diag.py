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


VARIANTS4 = (2, 16, 32, 48, 128)          # k_wconv4 (csrc/conv_wino4_f16x3.hip, -DWCONV4_DIAG): 1 barrier, 2 epilogue, 4 in-loop DMA, 8 fragment reads


def lib_of(v, k4=False):
    return os.path.join(DIAG, f"libdua_hip_w{'4' if k4 else ''}diag{v}.so")


def build(k4=False):
    os.makedirs(DIAG, exist_ok=True)
    csrc = os.path.join(PKG, "csrc")
    src = "conv_wino4_f16x3" if k4 else "conv_wino_f16x3"
    objs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".o") and f != src + ".o"]
    for v in (VARIANTS4 if k4 else VARIANTS):
        o = os.path.join(DIAG, f"{src}.wdiag{v}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result",
                               f"-DWCONV{'4' if k4 else ''}_DIAG={v}", "-c", os.path.join(csrc, src + ".hip"), "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [o, "-o", lib_of(v, k4)])
        print("built", lib_of(v, k4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--only", default="128x256x256,64x512x512,16x768x768")
    ap.add_argument("--kernel", default="w2", choices=["w2", "w4"], help="w2 = k_wconv (F(2,3)), w4 = k_wconv4 (F(4,3))")
    a = ap.parse_args()
    k4 = a.kernel == "w4"
    if a.build:
        return build(k4)
    mix = os.path.join(ROOT, "scripts", "bench_conv_mix.py")
    for v in (0,) + (VARIANTS4 if k4 else VARIANTS):
        cmd = [sys.executable, mix, "--fmt", "f16x3w4" if k4 else "f16x3w", "--only", a.only] + (["--lib", lib_of(v, k4)] if v else [])
        out = subprocess.run(cmd, capture_output=True, text=True).stdout
        print(f"---- WCONV{'4' if k4 else ''}_DIAG={v}")
        print("\n".join(l for l in out.splitlines() if l.startswith("H=") or l.startswith("mix")), flush=True)


if __name__ == "__main__":
    main()
