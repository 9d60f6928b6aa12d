#!/usr/bin/env python3
"""Diagnostic: where a tile of the raw-operand skip GEMM kernel (csrc/gemm_skip_f16x3.hip: k_sgemm) spends its time.

    python scripts/diag/sgemm_diag.py --build      # here in the container: timing-only libraries in scripts/diag/build/
    python scripts/diag/sgemm_diag.py              # on the GPU box: scripts/bench_gemm_mix.py --set adm on the skip shapes, per library

Variants (-DSG_DIAG bits: 1 no epilogue, 2 no operand split (raw bits as fragments), 4 no prefetch of the next item's first stages)
produce WRONG results; they exist only outside the product package."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "diffusion-uncertainty_amd")
DIAG = os.path.join(ROOT, "scripts", "diag", "build")
VARIANTS = (1, 2, 3, 4)
# named A/B builds (CORRECT results): extra -D flags
NAMED = {"spread": ["-DSG_SPLIT_SPREAD=1"], "nt": ["-DSG_STORE_AUX=2"]}


def lib_of(v):
    return os.path.join(DIAG, f"libdua_hip_sgdiag{v}.so")


def build(extra=()):
    os.makedirs(DIAG, exist_ok=True)
    csrc = os.path.join(PKG, "csrc")
    objs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".o") and f != "gemm_skip_f16x3.o"]
    for v in list(VARIANTS) + list(NAMED):
        o = os.path.join(DIAG, f"gemm_skip_f16x3.sgdiag{v}.o")
        flags = NAMED[v] if v in NAMED else [f"-DSG_DIAG={v}"]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result",
                               *flags, *extra, "-c", os.path.join(csrc, "gemm_skip_f16x3.hip"), "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [o, "-o", lib_of(v)])
        print("built", lib_of(v))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--only", default="skip128,skip64,skip32,skip16,skip8")
    ap.add_argument("--sg", default="0")
    ap.add_argument("--variants", default=",".join(str(v) for v in (0,) + VARIANTS))
    a = ap.parse_args()
    if a.build:
        return build()
    mix = os.path.join(ROOT, "scripts", "bench_gemm_mix.py")
    for v in [(int(t) if t.isdigit() else t) for t in a.variants.split(",")]:
        cmd = [sys.executable, mix, "--set", "adm", "--only", a.only, "--sg", a.sg] + (["--lib", lib_of(v)] if v else [])
        out = subprocess.run(cmd, capture_output=True, text=True)
        print(f"---- SG_DIAG={v}")
        print("\n".join(l for l in out.stdout.splitlines() if l.startswith("skip") or l.startswith("mix")), flush=True)
        if out.returncode:
            print(out.stderr[-2000:])


if __name__ == "__main__":
    main()
