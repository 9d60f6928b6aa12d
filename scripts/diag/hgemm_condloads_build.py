#!/usr/bin/env python3
"""A/B library with k_hgemm's round-5 epilogue loads (-DHG_COND_LOADS=1: residual rows, column scales and biases through CONDITIONAL
global loads, behind which the compiler waits vmcnt(0) - draining the output stores in flight - four to seven times per tile) against the
product's always-issued buffer loads.

    python scripts/diag/hgemm_condloads_build.py     # here in the container -> scripts/diag/build/libdua_hip_condloads.so
    python scripts/bench_gemm_mix.py --set uvit64 --lib scripts/diag/build/libdua_hip_condloads.so      # ... and without --lib"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "diffusion-uncertainty_amd", "csrc")
DIAG = os.path.join(ROOT, "scripts", "diag", "build")
sys.path.insert(0, CSRC)
import build as B  # noqa: E402

os.makedirs(DIAG, exist_ok=True)
objs = []
for src, flags in B.SOURCES:
    name = src[:-4]
    if name != "gemm_f16x3":
        objs.append(os.path.join(CSRC, name + ".o"))
        continue
    o = os.path.join(DIAG, name + ".condloads.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-DHG_COND_LOADS=1",
                           *flags, "-c", os.path.join(CSRC, src), "-o", o])
    objs.append(o)
out = os.path.join(DIAG, "libdua_hip_condloads.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
print("built", out)
