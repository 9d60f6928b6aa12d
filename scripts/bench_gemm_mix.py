#!/usr/bin/env python3
"""Per-shape micro-benchmark of the f16x3 one-tap GEMMs (presplit LDS-DMA form and on-the-fly-split form).

    python scripts/bench_gemm_mix.py [--set uvit64|uvit320|adm] [--engine 1] [--reps 5] [--splitk]

Shapes: the linears of one U-ViT-H/2 block at batch 64 / 320 (BASELINE configs[2]: 258 tokens per image; uvit/uvit.py:54-120,
timm.py:96-111) and the 1x1 skip convs / qkv / proj_out GEMMs of one ADM-128 forward at batch 128 (unet_openai.py:219-223,287,295).
Prints per shape the mean launch time and the algorithmic TFLOP/s (2*M*N*K), then the FLOP-weighted mix.  Random operands."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def uvit_shapes(B):
    M = B * 258
    # (name, M, K1, K2, N, kind, count per forward); kind: p = presplit, g = presplit + GELU -> planes, r = presplit + residual, s = on-the-fly split
    return [("qkv", M, 1152, 0, 3456, "p", 29), ("proj", M, 1152, 0, 1152, "r", 29), ("fc1", M, 1152, 0, 4608, "g", 29),
            ("fc2", M, 4608, 0, 1152, "r", 29), ("skip", M, 1152, 1152, 1152, "s", 14)]


def adm_shapes(B=128):
    sk = [(128, 256, 256, 256, 3), (64, 512, 256, 256, 1), (64, 256, 256, 256, 2), (32, 768, 512, 512, 1), (32, 512, 512, 512, 1),
          (32, 512, 256, 512, 1), (32, 256, 0, 512, 1), (16, 1024, 768, 768, 1), (16, 768, 768, 768, 1), (16, 768, 512, 768, 1), (16, 512, 0, 768, 1),
          (8, 1024, 1024, 1024, 2), (8, 1024, 768, 1024, 1), (8, 768, 0, 1024, 1)]
    out = [(f"skip{H}", B * H * H, c1, c2, n, "s", cnt) for H, c1, c2, n, cnt in sk]
    for H, C, cnt in ((32, 512, 5), (16, 768, 5), (8, 1024, 6)):
        out.append((f"qkv{H}", B * H * H, C, 0, 3 * C, "p", cnt))
        out.append((f"proj{H}", B * H * H, C, 0, C, "r", cnt))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="uvit64")
    ap.add_argument("--engine", type=int, default=1)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--splitk", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--lib", default="", help="alternative libdua_hip build to load")
    ap.add_argument("--skip-presplit", action="store_true",
                    help="time the on-the-fly-split shapes ([s]) ALSO as presplit two-source GEMMs on planes, plus the pass that would write those planes")
    ap.add_argument("--hg", default="", help="dua_set_hgemm_mode: nt[,streamk]  (nt -1 = previous engine, 0 = auto, 2 / 3 / 4)")
    ap.add_argument("--sg", type=int, default=0, help="dua_set_sgemm_mode for the on-the-fly-split shapes [s]: -1 = previous engine (k_xtile), 0 = auto, 2 / 4")
    a = ap.parse_args()
    if a.lib:
        from diffusion_uncertainty_amd import _lib as _l
        _l.LIB_PATH = a.lib
    from diffusion_uncertainty_amd import _lib, ops
    _lib.lib().dua_set_dma_conv(a.engine)
    if a.hg:
        v = [int(t) for t in a.hg.split(",")]
        assert _lib.lib().dua_set_hgemm_mode(v[0], v[1] if len(v) > 1 else 1) == 0
    assert _lib.lib().dua_set_sgemm_mode(a.sg) == 0
    dev = torch.device("cuda")
    if a.splitk:
        ops.set_split_k(True, dev)
    shapes = {"uvit64": uvit_shapes(64), "uvit320": uvit_shapes(320), "adm": adm_shapes()}[a.set]
    tf, tt = 0.0, 0.0
    tp = [0.0, 0.0, 0.0]
    for name, M, K1, K2, N, kind, cnt in shapes:
        if a.only and name not in a.only.split(","):
            continue
        K = K1 + K2
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        pk = ops.pack_weight_f16x3(w)
        res = torch.randn(M, N, device=dev) if kind == "r" else None
        if kind == "s":
            a1 = torch.randn(M, K1, device=dev)
            a2 = torch.randn(M, K2, device=dev) if K2 else None
            fn = lambda: ops.gemm_f16x3(a1, pk, N, bias=b, a2=a2)  # noqa: E731   (the ResBlock's call: bias only, blocks.hip / unet_openai.py)
        else:
            x = torch.randn(M, K, device=dev) * ops.F16X3_ACT_SCALE
            hi = x.half()
            pl = ops.SplitPlanes(torch.stack([hi, (x - hi.float()).half()]).contiguous().view(torch.int16), (M, K), "f16x3", ops.F16X3_ACT_SCALE)
            del x, hi
            if kind == "g":
                fn = lambda: ops.gemm_bf16x3_presplit(pl, pk, N, bias=b, act_out=2, split_out=True)  # noqa: E731
            else:
                fn = lambda: ops.gemm_bf16x3_presplit(pl, pk, N, bias=b, residual=res, colsum=(a.set == "adm" and kind == "r"))  # noqa: E731
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        fl = 2.0 * M * N * K
        extra = ""
        if kind == "s" and a.skip_presplit:
            def planes(t):
                v = t * ops.F16X3_RAW_SCALE
                h = v.half()
                return ops.SplitPlanes(torch.stack([h, (v - h.float()).half()]).contiguous().view(torch.int16), tuple(t.shape), "f16x3", ops.F16X3_RAW_SCALE)
            p1, p2 = planes(a1), (planes(a2) if a2 is not None else None)
            f2 = lambda: ops.gemm_f16x3_presplit2(p1, p2, pk, N, bias=b, colsum=(a.set == "adm"))  # noqa: E731
            # the pass that writes the planes: an identity GroupNorm-apply at the raw scale (reads 4 B, writes 4 B per element) - an upper
            # bound of what a second output of the in_layers writer (which reads the operand anyway) would cost: ~ its write half
            cat = a1 if a2 is None else torch.cat([a1, a2], 1)
            st = torch.tensor([0.0, 1.0], device=dev).repeat(32, 1).contiguous()
            Cc = cat.shape[1]
            f3 = lambda: ops.groupnorm_apply(cat.view(1, M, 1, Cc), st, torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev), act=0,  # noqa: E731
                                             split="f16x3", split_scale=ops.F16X3_RAW_SCALE)
            res2 = []
            for f in (f2, f3):
                f(); f()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(a.reps):
                    f()
                e1.record()
                torch.cuda.synchronize()
                res2.append(e0.elapsed_time(e1) / a.reps)
            extra = f" | presplit2 {res2[0]:7.3f} ms {fl / res2[0] / 1e9:6.1f} TF, plane pass {res2[1]:7.3f} ms"
            tp[0] += cnt * res2[0]
            tp[1] += cnt * res2[1]
            tp[2] += cnt * ms
        print(f"{name:8s} M={M:8d} K={K1}+{K2} N={N:5d} [{kind}] x{cnt:2d}: {ms:8.3f} ms  {fl / ms / 1e9:6.1f} TF{extra}", flush=True)
        tf += cnt * fl
        tt += cnt * ms
    if a.skip_presplit:
        print(f"skip GEMMs per forward: on-the-fly split {tp[2]:.2f} ms | presplit two-source {tp[0]:.2f} ms + plane pass (upper bound) {tp[1]:.2f} ms")
    print(f"mix ({a.set}, engine {a.engine}, hgemm mode '{a.hg}'{', split-K' if a.splitk else ''}): {tt:.2f} ms per forward, {tf / tt / 1e9:.1f} TF algorithmic")


if __name__ == "__main__":
    main()
