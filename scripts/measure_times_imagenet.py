#!/usr/bin/env python3
"""Wall time of the sampling driver with and without the uncertainty window (counterpart of the reference's
scripts/measure_times_imagenet.py on the MI355X engine - the script SURVEY.md 8(d) takes the metric's definition from).

Same flags as scripts/generate_dataset_score_uncertainty_imagenet.py (the reference's two scripts share their ``parse_args``, :163-212) and
the reference's flow (:36-138): starting points and denoiser as in the generation script, then the driver call timed twice -
``time_uc`` with the uncertainty scheduler (:88-98: the span bench.py's `value` is quoted on) and ``time_normal`` with the plain scheduler
built from the same config (:103-118: DDIM for the ADM models, DPM-Solver-2 for the U-ViT ones) - and one record
``{dataset, image_size, time_uc, time_normal, model_type, scheduler_type}`` appended to ``RESULTS/times.json``.  Both spans are bracketed by
``torch.cuda.synchronize``; the plain scheduler is the uncertainty scheduler of its family with an EMPTY window behind
``schedulers_uncertainty/plain.py`` (same arithmetic as the diffusers schedulers the reference builds; no map is collected).  The first
call of a fresh model also packs its weights: ``--warmup`` (default 1) runs one untimed batch first, which the reference does not do.
"""
import argparse
import importlib.util
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT_DIR = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT_DIR))

_spec = importlib.util.spec_from_file_location("dua_generate_cli", str(Path(__file__).resolve().parent / "generate_dataset_score_uncertainty_imagenet.py"))
gen_cli = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen_cli)

from diffusion_uncertainty_amd.generate_samples import (  # noqa: E402
    generate_samples_model_scheduler_class_conditioned_from_tensor, generate_samples_model_scheduler_class_conditioned_uvit_from_tensor)
from diffusion_uncertainty_amd.paths import RESULTS  # noqa: E402
from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler  # noqa: E402
from diffusion_uncertainty_amd.schedulers_uncertainty.noise import DeviceNoise  # noqa: E402
from diffusion_uncertainty_amd.schedulers_uncertainty.plain import PlainScheduler  # noqa: E402
from diffusion_uncertainty_amd.utils import seed_everything  # noqa: E402


def plain_scheduler(args, y, model, base):
    """reference :103-107: DPM-Solver-2 for the U-ViT models (image sizes above 128), DDIM otherwise - the uncertainty scheduler of that
    family with an empty window, behind the non-uncertainty wrapper."""
    kind = "dpm_2_uncertainty_centered" if (args.image_size > 128 or (args.image_size == 128 and args.model_type == "uvit")) else "uncertainty_zigzag_centered"
    a = argparse.Namespace(**dict(args.__dict__, scheduler_type=kind, M=1, start_step_uc=args.generation_steps - 1, num_steps_uc=0, num_zigzag=1))
    s = get_uncertainty_scheduler(a, y, model, base)
    s.set_timesteps(args.generation_steps)
    return PlainScheduler(s)


@torch.no_grad()
def measure(gpu_idx, args, num_samples, times_path):
    seed_everything(args.seed)
    device = torch.device("cuda", gpu_idx)
    torch.cuda.set_device(device)
    X_T, y = gen_cli.load_starting_points(gpu_idx, args, num_samples)
    model, base = gen_cli.build_denoiser(args, device)

    def run(scheduler, X, yy):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        if args.model_type == "unet":
            r = generate_samples_model_scheduler_class_conditioned_from_tensor(X, yy, args.batch_size, device, model, scheduler)
        else:
            r = generate_samples_model_scheduler_class_conditioned_uvit_from_tensor(X, yy, args.batch_size, model, scheduler, device)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0, r

    def uc():
        s = get_uncertainty_scheduler(args, y.to(device), model, base)
        if args.noise == "device":
            s.noise_source = DeviceNoise(seed=args.seed + gpu_idx)
        s.set_timesteps(args.generation_steps)
        return s

    for _ in range(args.warmup):                                         # untimed: weight packing, code paging, allocator
        run(uc(), X_T[:args.batch_size], y[:args.batch_size])
    s = uc()
    delta_uc, r = run(s, X_T, y)
    print("Time Uncertainty estimation", delta_uc)
    forwards_uc = args.generation_steps + s.denoiser_calls
    del r
    delta_normal, r = run(plain_scheduler(args, y.to(device), model, base), X_T, y)
    assert "uncertainty" not in r
    print("Time normal", delta_normal)
    model_type = "adm" if args.model_type == "unet" else args.model_type
    results = dict(dataset="imagenet", image_size=args.image_size, time_uc=delta_uc, time_normal=delta_normal, model_type=model_type,
                   scheduler_type=args.scheduler_type, num_samples=int(X_T.shape[0]), batch_size=args.batch_size, generation_steps=args.generation_steps,
                   M=args.M, images_per_second_uc=X_T.shape[0] / delta_uc, images_per_second_normal=X_T.shape[0] / delta_normal,
                   executed_forwards_per_batch_uc=forwards_uc, precision=args.precision, gpu=gpu_idx)
    print("results", results)
    times_path = Path(times_path)
    times = json.load(open(times_path)) if times_path.exists() else []
    times.append(results)
    times_path.parent.mkdir(parents=True, exist_ok=True)
    with open(times_path, "w") as f:
        json.dump(times, f)
    return results


def main():
    extra = argparse.ArgumentParser(add_help=False)
    extra.add_argument("--warmup", type=int, default=1)
    extra.add_argument("--times-file", type=str, default=None)
    ex, rest = extra.parse_known_args()
    args = gen_cli.parse_args(rest)
    args.warmup, args.dataset = ex.warmup, f"imagenet{args.image_size}"
    if args.scheduler_type == "mc_dropout":
        assert args.image_size == 64, "mc_dropout only implemented for image_size=64"
    num_gpus = torch.cuda.device_count()
    if num_gpus == 0:
        raise SystemExit("no GPU visible: the engine has no CPU path")
    times_path = ex.times_file or os.path.join(str(RESULTS), "times.json")
    if num_gpus == 1:
        measure(0, args, args.num_samples, times_path)
    else:                                                                # (one record per rank; the ranks append to files of their own)
        torch.multiprocessing.spawn(fn=_spawned, args=(args, args.num_samples // num_gpus, times_path), nprocs=num_gpus, join=True)
    print("Done")


def _spawned(gpu_idx, args, num_samples, times_path):
    base, ext = os.path.splitext(times_path)
    measure(gpu_idx, args, num_samples, f"{base}_{gpu_idx}{ext}")


if __name__ == "__main__":
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    main()
