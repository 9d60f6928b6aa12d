#!/usr/bin/env python3
"""A/B library without the f16x3 range-check code (-DDUA_NO_RANGE_CHECK: dua_range_ctr() is a constant null pointer, every check is
compiled out): prices what the debug facility costs the plane writers when it is switched off.

    python scripts/diag/norange_build.py            # here in the container -> scripts/diag/build/libdua_hip_norange.so
    ... --lib scripts/diag/build/libdua_hip_norange.so on scripts/bench_gemm_mix.py / bench_conv_mix.py; DUA_HIP_LIB=... for bench.py"""
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
only = set(sys.argv[1:])                  # translation units to rebuild without the check (default: all)
for src, flags in B.SOURCES:
    name = src[:-4]
    if only and name not in only:
        objs.append(os.path.join(CSRC, name + ".o"))
        continue
    o = os.path.join(DIAG, name + ".norange.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-DDUA_NO_RANGE_CHECK",
                           *flags, "-c", os.path.join(CSRC, src), "-o", o])
    objs.append(o)
out = os.path.join(DIAG, "libdua_hip_norange.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
print("built", out)
