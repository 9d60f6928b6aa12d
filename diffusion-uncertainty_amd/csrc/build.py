"""Build libdua_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libdua_hip.so")
SOURCES = [("sampler.hip", ["-ffp-contract=off"]), ("denoiser.hip", []), ("mfma_gemm.hip", []), ("mfma_bf16x3.hip", []), ("conv_f16x3.hip", []), ("gemm_f16x3.hip", []), ("gemm_skip_f16x3.hip", []), ("conv_wino_f16x3.hip", []), ("conv_wino4_f16x3.hip", []), ("conv_wino_f32.hip", []), ("conv_wino4_f32.hip", []), ("attention.hip", []), ("blocks.hip", []), ("backward.hip", [])]
COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.join(HERE, "common.h"), os.path.join(HERE, "dconv.h"), os.path.join(HERE, "wconv.h"), os.path.join(HERE, "hgemm.h"), os.path.join(os.path.dirname(PKG), "include", "dua_hip.h")]
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(HERE, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


def build_select_variant(verbose=True):
    """libdua_hip_select.so: the product sources with conv_f16x3.hip, conv_wino_f16x3.hip and conv_wino_f32.hip compiled -DDUA_RS_ZERO_SELECT=1 (the
    row-shift convs zero their image-edge fragments by a register select instead of an out-of-range LDS read).  Test artefact (scripts/diag/build/, outside the
    package): tests/test_gpu_ops.py compares the two forms bit for bit."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out_dir = os.path.join(os.path.dirname(PKG), "scripts", "diag", "build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libdua_hip_select.so")
    hdrs = [os.path.join(HERE, "common.h"), os.path.join(HERE, "dconv.h"), os.path.join(HERE, "wconv.h"), os.path.join(HERE, "hgemm.h"), os.path.join(os.path.dirname(PKG), "include", "dua_hip.h")]
    alt = {}
    for name in ("conv_f16x3.hip", "conv_wino_f16x3.hip", "conv_wino4_f16x3.hip", "conv_wino_f32.hip"):   # (conv_wino4_f32.hip has the out-of-range form only)          # the kernels with row-shifted fragment reads
        src = os.path.join(HERE, name)
        obj = os.path.join(out_dir, name.replace(".hip", ".select.o"))
        if _stale(obj, [src] + hdrs):
            cmd = [hipcc] + COMMON + ["-DDUA_RS_ZERO_SELECT=1", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        alt[name] = obj
    objs = [alt.get(s, os.path.join(HERE, s.replace(".hip", ".o"))) for s, _ in SOURCES]
    if _stale(out, objs):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
