#!/usr/bin/env python3
"""Headline benchmark: images/sec @ ADM ImageNet-128, 50 DDIM steps, M=5, batch 128 per GPU
(BASELINE.json configs[1]), `uncertainty_zigzag_centered`, num_zigzag=3, UC window = README window W10
(--start-step-uc 40 --num-steps-uc 10), eta=0, synthetic X_T/y, seeded random-init weights.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one complete sampling job of one batch of 128 images on each GPU (50 main denoiser forwards +
the UC-window forwards + all sampler kernels + uint8 conversion, inputs resident in HBM).  A warmup step is
a truncated job (3 timesteps, one of them in the UC window): untimed, it only pages in code, packs the
weights and warms the allocator.  Ranks shard independent batches (weak scaling); the only collective is
one RCCL all_gather of the uint8 images at the end of each step.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     - dominant kernel (implicit-GEMM conv3x3 of the selected precision mode): algorithmic FLOPs per launch / mean
                 launch duration, HIP events on the launch stream, over the timed region; peak = 2.5 PFLOP/s (fp16 / bf16 MFMA)
                 for the split modes, 157.3 TFLOP/s (fp32-input MFMA) for --precision fp32.
  cpu_baseline - the CPU oracle (port of the reference path, fp32, all host cores) timed on rank 0 at N=1 on a bounded
                 sample and extrapolated with the reference's exact forward count (200 / image for W10).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ADM128_FLOPS_PER_FWD_IMG = 6.139e11      # BASELINE.md section 3 (2*MAC, hooks on the reference modules)
FP32_MFMA_PEAK_TFLOPS = 157.3            # MI355X_MICROARCH.md chip table (fp32-input MFMA)
BF16_MFMA_PEAK_TFLOPS = 2500.0           # MI355X_MICROARCH.md chip table (dense bf16 MFMA)


class ConvProfile:
    def __init__(self):
        self.items = []

    def record(self, kind, flops, e0, e1):
        self.items.append((flops, e0, e1))

    # HBM-side bytes per conv launch from the PMC passes of profiles/r01d_pmc_traffic_conv.json (FETCH_SIZE doubled per the
    # gfx950 correction + WRITE_SIZE); collected on a truncated job with the same batch and layer shapes.
    PMC_TRAFFIC_BF16X3 = 2.145e9
    PMC_TRAFFIC_F16F8 = 1.999e9       # profiles/r01j_pmc_traffic_conv.json (mean over the conv and 1-tap GEMM launches of the kernel)
    PMC_TRAFFIC_F16F6 = 2.009e9       # profiles/r01l_pmc_traffic_conv.json (same mean)
    PMC_TRAFFIC_F16F6_SOURCE = "profiles/r01l_pmc_traffic_conv.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"

    def summary(self):
        if not self.items:
            return None
        total_ms = sum(e0.elapsed_time(e1) for _, e0, e1 in self.items)
        total_fl = sum(f for f, _, _ in self.items)
        n = len(self.items)
        return dict(launches=n, flops_per_launch=total_fl / n, ms_per_launch=total_ms / n,
                    tflops=total_fl / (total_ms * 1e-3) / 1e12)

    def roofline(self, precision):
        cs = self.summary()
        if cs is None:
            return None
        if precision == "fp32":
            return dict(bound="mfma", achieved=round(cs["tflops"], 2), peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=round(cs["tflops"] / FP32_MFMA_PEAK_TFLOPS, 4), traffic=None,
                        kernel="k_tile<MODE_CONV> (fp32 MFMA implicit-GEMM conv3x3, v_mfma_f32_32x32x2_f32)",
                        launches=cs["launches"], flops_per_launch=cs["flops_per_launch"], ms_per_launch=round(cs["ms_per_launch"], 4))
        if precision == "f16f8":
            return dict(bound="mfma", achieved=round(cs["tflops"], 2), peak=BF16_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=round(cs["tflops"] / BF16_MFMA_PEAK_TFLOPS, 4), traffic=self.PMC_TRAFFIC_F16F8,
                        traffic_source="profiles/r01j_pmc_traffic_conv.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                        kernel="k_xconv_dma<NT, 1> (LDS-DMA implicit-GEMM conv3x3, fp16 hi*hi on v_mfma_f32_32x32x16_f16 + both cross "
                               "terms on one block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 per 32-deep K block)",
                        launches=cs["launches"], flops_per_launch=cs["flops_per_launch"], ms_per_launch=round(cs["ms_per_launch"], 4),
                        executed_mfma_tflops=round(2 * cs["tflops"], 1), executed_frac=round(2 * cs["tflops"] / BF16_MFMA_PEAK_TFLOPS, 4),
                        note="achieved counts ALGORITHMIC fp32 FLOPs; each costs the matrix-core cycles of 2 bf16-rate products (one fp16 "
                             "product + two fp8 products at twice the rate), so frac <= 1/2 against the 2.5 PF bf16/fp16 peak by construction")
        if precision == "f16f6":
            return dict(bound="mfma", achieved=round(cs["tflops"], 2), peak=BF16_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=round(cs["tflops"] / BF16_MFMA_PEAK_TFLOPS, 4), traffic=self.PMC_TRAFFIC_F16F6,
                        traffic_source=self.PMC_TRAFFIC_F16F6_SOURCE,
                        kernel="k_xconv_dma<NT, 2, true> (LDS-DMA row-shift implicit-GEMM conv3x3, fp16 hi*hi on v_mfma_f32_32x32x16_f16 + "
                               "both cross terms on one MXFP6 (e2m3, E8M0 exponent per 32-k block) v_mfma_scale_f32_32x32x64_f8f6f4 per "
                               "32-deep K block)",
                        launches=cs["launches"], flops_per_launch=cs["flops_per_launch"], ms_per_launch=round(cs["ms_per_launch"], 4),
                        executed_mfma_tflops=round(1.5 * cs["tflops"], 1),
                        executed_frac=round(1.5 * cs["tflops"] / BF16_MFMA_PEAK_TFLOPS, 4),
                        note="achieved counts ALGORITHMIC fp32 FLOPs; each costs the matrix-core cycles of 1.5 bf16-rate products (one "
                             "fp16 product + two fp6 products at four times the rate), so frac <= 2/3 against the 2.5 PF bf16/fp16 peak "
                             "by construction")
        return dict(bound="mfma", achieved=round(cs["tflops"], 2), peak=BF16_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(cs["tflops"] / BF16_MFMA_PEAK_TFLOPS, 4), traffic=self.PMC_TRAFFIC_BF16X3,
                    traffic_source="profiles/r01d_pmc_traffic_conv.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                    kernel="k_xconv_dma<NT> (LDS-DMA split-bf16 implicit-GEMM conv3x3, 3 x v_mfma_f32_32x32x16_bf16 per product)",
                    launches=cs["launches"], flops_per_launch=cs["flops_per_launch"], ms_per_launch=round(cs["ms_per_launch"], 4),
                    executed_mfma_tflops=round(3 * cs["tflops"], 1), executed_frac=round(3 * cs["tflops"] / BF16_MFMA_PEAK_TFLOPS, 4),
                    note="achieved counts ALGORITHMIC fp32 FLOPs; each is executed as 3 bf16 MFMA products (hi*hi + hi*lo + lo*hi), "
                         "so frac <= 1/3 by construction")


def build_model(device, seed=128):
    from diffusion_uncertainty_amd.init_model import init_guided_diffusion_imagenet128_and_scheduler
    from diffusion_uncertainty_amd.utils import seeded_state_dict
    model, base = init_guided_diffusion_imagenet128_and_scheduler(load_checkpoint=False)
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed)
    model.load_state_dict(sd)
    return model.eval().to(device), base, sd


def make_scheduler(model, base, M, start, num, nz, seed, faithful):
    from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler
    from diffusion_uncertainty_amd.schedulers_uncertainty.noise import DeviceNoise
    args = argparse.Namespace(scheduler_type="uncertainty_zigzag_centered", M=M, start_step_uc=start, num_steps_uc=num,
                              predict_next=True, eta=0.0, num_zigzag=nz, uncertainty_distance=1)
    s = get_uncertainty_scheduler(args, None, model, base)
    s.noise_source = DeviceNoise(seed=seed)          # inputs/noise generated in HBM; host tape only for parity tests
    s.faithful_zigzag = faithful
    return s


def cpu_baseline(sd, threads):
    """Oracle (CPU port of the reference path) on a bounded sample: bs=2, two DDIM steps, one of them in the UC window
    (M=1, num_zigzag=2, every forward executed like the reference) = 4 forwards x 2 images; about 10-30 s of CPU work."""
    from oracle import adm, driver, schedule
    torch.set_num_threads(threads)
    cfg = dict(adm.ADM128)
    g = torch.Generator().manual_seed(49394)
    X_T, y = torch.randn(2, 3, 128, 128, generator=g), torch.randint(0, 1000, (2,), generator=g)
    nfwd = [0]

    def model(x, t, yy, train=False):
        nfwd[0] += 1
        return adm.forward(sd, cfg, x, t, yy)

    with torch.no_grad():
        model(X_T, torch.tensor([500, 500]), y)          # untimed: first-touch of the 1.7 GB of weights, thread pool start
    nfwd[0] = 0
    torch.manual_seed(0)
    t0 = time.perf_counter()
    driver.generate(X_T, y, 2, model, schedule.Tables(), 2, "zigzag_centered", M=1, after_step=1, num_steps_uc=1, num_zigzag=2)
    dt = time.perf_counter() - t0
    s_per_img_fwd = dt / (nfwd[0] * 2)
    ref_forwards = 50 + 10 * 5 * 3                   # the reference executes every zig-zag forward (W10: 200 / image)
    return dict(value=1.0 / (ref_forwards * s_per_img_fwd), unit="images/sec", cores=threads, kind="port",
                sample=f"oracle ADM-128 fp32, bs=2, 2 DDIM steps (1 in UC window, M=1, nz=2): {nfwd[0]} forwards in {dt:.1f}s "
                       f"-> {s_per_img_fwd:.3f} s/forward/image, extrapolated to the reference's {ref_forwards} forwards/image")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--generation-steps", type=int, default=50)
    ap.add_argument("-M", type=int, default=5)
    ap.add_argument("--num-zigzag", type=int, default=3)
    ap.add_argument("--start-step-uc", type=int, default=40)
    ap.add_argument("--num-steps-uc", type=int, default=10)
    ap.add_argument("--faithful-zigzag", action="store_true", help="execute the dead num_zigzag-1 forwards too")
    ap.add_argument("--precision", choices=["f16f6", "f16f8", "bf16x3", "fp32"], default="f16f6",
                    help="f16f6: as f16f8 with the cross terms of the 3x3 convs in MXFP6 (parity ~7e-5); f16f8: 3x3 convs / qkv on fp16 + block-scaled fp8 matrix-core products (parity ~5e-5); bf16x3: convs / weight "
                         "GEMMs on the bf16 matrix cores with split fp32 operands (parity ~3e-5); fp32: exact fp32 MFMA (parity ~7e-6)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the one-step run of the other precision mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ          # under torch.distributed.run
    if world > 1 or launched:
        dist.init_process_group("nccl", device_id=dev)

    from diffusion_uncertainty_amd import ops
    from diffusion_uncertainty_amd.distributed import gather_images, max_over_ranks, shard_range
    from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_from_tensor as gen

    model, base, sd = build_model(dev)
    # synthetic inputs exactly as scripts/generate_diffusion_starting_data.py:27-49, sharded by rank like the script (:51)
    g = torch.Generator().manual_seed(49394)
    X_all = torch.randn(a.batch * world, 3, 128, 128, generator=g)
    y_all = torch.randint(0, 1000, (a.batch * world,), generator=g)
    lo, hi = shard_range(a.batch * world, rank, world)
    X_T, y = X_all[lo:hi].to(dev), y_all[lo:hi].to(dev)

    def job(n_steps, start, num, seed):
        s = make_scheduler(model, base, a.M, start, num, a.num_zigzag, seed, a.faithful_zigzag)
        s.set_timesteps(n_steps)
        r = gen(X_T, y, a.batch, dev, model, s)
        gather_images(r["gen_images"].to(dev))             # RCCL over xGMI: the only collective of the path
        return s.denoiser_calls

    def sync():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def timed(precision, steps, warmup):
        model.precision = precision
        for w in range(warmup):
            job(3, 1, 1, 1000 + w)
        prof = None if a.no_profile else ConvProfile()
        ops.PROFILE = prof
        sync()
        t0 = time.perf_counter()
        uc = 0
        for k in range(steps):
            uc += job(a.generation_steps, a.start_step_uc, a.num_steps_uc, k)
        sync()
        dt = time.perf_counter() - t0
        ops.PROFILE = None
        return max_over_ranks(dt, dev), uc, prof

    dt, uc_calls, prof = timed(a.precision, a.steps, a.warmup)
    secondary = None
    if world == 1 and not a.no_secondary:
        other = "fp32" if a.precision != "fp32" else "f16f6"
        dt2, uc2, prof2 = timed(other, 1, 1)
        secondary = dict(precision=other, value=round(a.batch / dt2, 4), unit="images/sec", steps=1, ms_per_step=round(dt2 * 1e3, 1),
                         end_to_end_tflops=round((a.generation_steps + uc2) * ADM128_FLOPS_PER_FWD_IMG * a.batch / dt2 / 1e12, 2),
                         roofline=(prof2.roofline(other) if prof2 is not None else None))

    if rank == 0:
        images = a.batch * world * a.steps
        fwd_per_img = a.generation_steps + uc_calls / max(a.steps, 1)
        exec_flops = fwd_per_img * ADM128_FLOPS_PER_FWD_IMG * images
        roof = prof.roofline(a.precision) if prof is not None else None
        line = {
            "metric": "images/sec @ ADM ImageNet-128, 50 steps, M=5, bs=128; 1/2/4/8 GPU",
            "value": round(images / dt, 4), "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f16f6": "f16f6 (fp32 operands split a = fp16 hi + lo; 3x3 convs: hi*hi on fp16 MFMA, both cross terms on one MXFP6 "
                               "(e2m3 + E8M0 exponent per 32 channels) scaled MFMA, fp32 accumulate; qkv / proj_out f16f8; 1x1 / attention "
                               "products split-bf16; norms/softmax/sampler fp32)",
                      "f16f8": "f16f8 (fp32 operands split a = fp16 hi + lo; hi*hi on fp16 MFMA, both cross terms on one block-scaled fp8 "
                               "MFMA, fp32 accumulate: ~2^-15 per product; 1x1 / attention products split-bf16; norms/softmax/sampler fp32)",
                      "bf16x3": "bf16x3 (fp32 operands split into 2 bf16 terms, 3 bf16 MFMA products, fp32 accumulate; norms/softmax/"
                                "attention products/sampler in fp32)", "fp32": "f32"}[a.precision], "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: ADM ImageNet-128, 50 DDIM steps, M=5, uncertainty_zigzag_centered, "
                                   "num_zigzag=3, UC window W10 (start 40, 10 steps), batch 128 per GPU",
                       "global_batch": a.batch * world, "parallelism": f"independent batches x{world} (no data-path collective; "
                                                                        "one RCCL all_gather of uint8 images per job)",
                       "zigzag_mode": "faithful (all forwards executed)" if a.faithful_zigzag else
                                      "skip_dead_zigzag (outputs identical; noise draws still consumed)",
                       "executed_forwards_per_image": fwd_per_img, "noise": "device Philox (dua_randn)",
                       "precision": a.precision,
                       "parity": ("full-size ADM-128 forward vs the reference CPU output: max abs err 6.8e-5 (f16f6), 5.3e-5 (f16f8), 3.1e-5 (bf16x3), "
                                  "7.4e-6 (fp32); tolerance 1e-3 (tests/test_gpu_models.py)"),
                       "warmup_pass": "3-timestep truncated job"},
            "end_to_end_tflops": round(exec_flops / dt / 1e12, 2),
            "roofline": roof,
        }
        if secondary is not None:
            line["other_precision_mode"] = secondary
        if not a.no_cpu_baseline and world == 1:
            try:
                cores = len(os.sched_getaffinity(0))
            except Exception:
                cores = os.cpu_count() or 1
            # torch's intra-op pool stops scaling (and collapses from oversubscription) well before a 256-core host is
            # full at batch 2: use at most 32 threads and report that count.
            line["cpu_baseline"] = cpu_baseline(sd, min(cores, 32))
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
