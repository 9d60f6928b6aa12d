#!/usr/bin/env python3
"""Per-layer micro-benchmark of the split-bf16 LDS-DMA conv3x3 kernel on the ADM-128 layer mix (B = 128).

    python scripts/bench_conv_mix.py [--batch 128] [--reps 5] [--engine 1]

Prints one line per distinct (H, Cin, Cout, residual) conv shape of one ADM-128 forward with its count, mean launch time
and algorithmic TFLOP/s, then the FLOP-weighted mix.  Inputs are random split planes; this prices the kernel alone."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

# (H, Cin, Cout, count without residual, count with residual) - one ADM-128 forward (unet_openai.py:481-617, Appendix A)
MIX = [
    (128, 256, 256, 3, 6), (128, 512, 256, 3, 0),
    (64, 256, 256, 3, 6), (64, 512, 256, 2, 0), (64, 768, 256, 1, 0), (64, 512, 512, 1, 1),
    (32, 256, 256, 1, 1), (32, 256, 512, 1, 0), (32, 512, 512, 1, 5), (32, 768, 512, 1, 0), (32, 1024, 512, 1, 0),
    (32, 1280, 512, 1, 0), (32, 768, 768, 1, 1),
    (16, 512, 512, 1, 1), (16, 512, 768, 1, 0), (16, 768, 768, 1, 5), (16, 1280, 768, 1, 0), (16, 1536, 768, 1, 0),
    (16, 1792, 768, 1, 0), (16, 1024, 1024, 1, 1),
    (8, 768, 768, 1, 1), (8, 768, 1024, 1, 0), (8, 1024, 1024, 3, 7), (8, 1792, 1024, 1, 0), (8, 2048, 1024, 2, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--engine", type=int, default=1)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--fmt", default="bf16x3", choices=["bf16x3", "f16x3", "f16x3w", "f16x3w4", "f16w", "f32w", "fp32"])
    ap.add_argument("--check", action="store_true", help="compare with an fp64 torch convolution (small shapes only)")
    ap.add_argument("--lib", default="", help="alternative libdua_hip build to load")
    ap.add_argument("--stagger", type=str, default="", help="phases,percent,min_tiles")
    a = ap.parse_args()
    from diffusion_uncertainty_amd import _lib
    if a.lib:
        _lib.LIB_PATH = a.lib
    from diffusion_uncertainty_amd import ops
    _lib.lib().dua_set_dma_conv(a.engine)
    if a.stagger:
        _lib.lib().dua_set_conv_stagger(*[int(v) for v in a.stagger.split(",")])
    dev = "cuda"
    tot_f, tot_t = 0.0, 0.0
    for H, Cin, Cout, n0, n1 in MIX:
        if a.only and f"{H}x{Cin}x{Cout}" not in a.only.split(","):
            continue
        for res, cnt in ((False, n0), (True, n1)):
            if cnt == 0:
                continue
            N = a.batch
            x = torch.randn(N, H, H, Cin, device=dev)
            w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02
            if a.fmt == "bf16x3":
                hi = x.to(torch.bfloat16)
                lo = (x - hi.float()).to(torch.bfloat16)
                buf = torch.stack([hi, lo]).view(torch.int16)
                xs = ops.SplitPlanes(buf, (N, H, H, Cin))
                planes = ops.pack_weight_bf16x3(w)
            elif a.fmt == "f16x3":
                st = torch.tensor([0.0, 1.0], device=dev).repeat(N * 32, 1).contiguous()
                xs = ops.groupnorm_apply(x, st, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), act=0, split="f16x3")
                planes = ops.pack_weight_f16x3(w)
                hi = lo = buf = None
            elif a.fmt in ("f32w", "fp32"):            # exact fp32: Winograd form (k_wconv32) / direct fp32 MFMA conv
                st = torch.tensor([0.0, 1.0], device=dev).repeat(N * 32, 1).contiguous()
                if a.fmt == "f32w" and ops.conv3x3_wino_supported(N, H, H, Cin, Cout):
                    xs = ops.groupnorm_apply(x, st, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), act=0, split="f32w")
                    planes = ops.pack_weight_f32_wino(w)
                else:
                    xs = x.clone()
                    planes = ops.pack_conv3x3_weight(w)
                hi = lo = buf = None
            elif a.fmt == "f16x3w4":                    # Winograd F(4,3) along y (k_wconv4)
                st = torch.tensor([0.0, 1.0], device=dev).repeat(N * 32, 1).contiguous()
                if ops.conv3x3_wino4_supported(N, H, H, Cin, Cout):
                    xs = ops.groupnorm_apply(x, st, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), act=0, split="f16x3w4")
                    planes = ops.pack_weight_f16x3_wino4(w)
                else:
                    xs = ops.groupnorm_apply(x, st, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), act=0, split="f16x3")
                    planes = ops.pack_weight_f16x3(w)
                hi = lo = buf = None
            elif a.fmt in ("f16x3w", "f16w"):           # Winograd form: three fp16 products per tap / one (mode "f16w")
                st = torch.tensor([0.0, 1.0], device=dev).repeat(N * 32, 1).contiguous()
                if ops.conv3x3_wino_supported(N, H, H, Cin, Cout):
                    xs = ops.groupnorm_apply(x, st, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), act=0, split=a.fmt)
                    planes = ops.pack_weight_f16x3_wino(w)
                else:
                    xs = ops.groupnorm_apply(x, st, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), act=0, split="f16x3")
                    planes = ops.pack_weight_f16x3(w)
                hi = lo = buf = None
            else:
                raise SystemExit(f"unknown --fmt {a.fmt}")
            bias = torch.randn(Cout, device=dev)
            r = torch.randn(N, H, H, Cout, device=dev) if res else None
            conv = ops.conv3x3_bf16x3 if isinstance(xs, ops.SplitPlanes) else ops.conv3x3
            y = conv(xs, planes, bias, Cout, residual=r, colsum=True)
            torch.cuda.synchronize()
            if a.check:
                ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
                if r is not None:
                    ref = ref + r.double()
                err = (y.double() - ref).abs().max().item()
                print(f"   check: max abs err {err:.3e} (|ref| max {ref.abs().max().item():.2f})", flush=True)
            del x, hi, lo, y
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                conv(xs, planes, bias, Cout, residual=r, colsum=True)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            fl = 2.0 * N * H * H * 9 * Cin * Cout
            print(f"H={H:4d} Cin={Cin:5d} Cout={Cout:5d} res={int(res)} x{cnt:2d}  {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF", flush=True)
            tot_f += cnt * fl
            tot_t += cnt * ms
            del xs, buf, r
    if tot_t:
        print(f"mix: {tot_f / 1e12:.1f} TFLOP in {tot_t:.1f} ms -> {tot_f / tot_t / 1e9:.1f} TF algorithmic")


if __name__ == "__main__":
    main()
