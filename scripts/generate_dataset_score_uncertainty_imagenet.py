#!/usr/bin/env python3
"""Generate a dataset of score / uncertainty maps (counterpart of the reference's
scripts/generate_dataset_score_uncertainty_imagenet.py on the MI355X engine).

Same flags and defaults as the reference's ``parse_args`` (:147-196) and the same flow as ``generate_and_save``
(:33-93): one process per GPU (``torch.multiprocessing.spawn``), rank r samples
``X_T[start + r*n : start + (r+1)*n]`` read from ``DIFFUSION_STARTING_POINTS/imagenet{size}{suffix}/{X_T,y}.pth``, per-rank
output files ``timestep_{rank}.pth`` / ``uncertainty_{scheduler}_{rank}.pth`` / ``gen_images_{rank}.pth`` (+ ``args.yaml``), :90-93.
Deviations: the score maps go to ``score_{rank}.pth`` (the reference's shared ``score.pth`` is a last-writer-wins race,
:91; rank 0 also provides ``score.pth`` and ``timesteps.pth``, the shared names, as links to its own files); missing starting-point files are an
ERROR unless ``--synthetic`` asks for generated ones; ``--random-weights`` runs without checkpoints; ``--precision`` and
``--noise`` select the engine modes.  The import block below names the same modules and symbols as the reference's :14-27.
"""
import argparse
import datetime
import os
import sys
from pathlib import Path

import torch
import yaml

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from diffusion_uncertainty_amd.generate_samples import (  # noqa: E402
    generate_samples_model_scheduler_class_conditioned_from_tensor,
    generate_samples_model_scheduler_class_conditioned_uvit_from_tensor)
from diffusion_uncertainty_amd.init_model import (  # noqa: E402
    init_guided_diffusion_imagenet64_and_scheduler, init_guided_diffusion_imagenet128_and_scheduler)
from diffusion_uncertainty_amd.paths import CONFIG, DIFFUSION_STARTING_POINTS, RESULTS  # noqa: E402
from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler  # noqa: E402
from diffusion_uncertainty_amd.schedulers_uncertainty.noise import DeviceNoise  # noqa: E402
from diffusion_uncertainty_amd.utils import (load_config, randomize_parameters_, seed_everything, starting_point_dataset_index,  # noqa: E402
                                             starting_points)
from diffusion_uncertainty_amd.uvit.load_pretrained_models import load_autoencoder_uvit, load_uvit, load_uvit_scheduler  # noqa: E402
from diffusion_uncertainty_amd.uvit.uvit_ae import UViTAE  # noqa: E402


def index_range(gpu_idx, args, num_samples):
    """The rank's slice of the starting points (reference :51): rank r owns ``[start + r*n, start + (r+1)*n)``."""
    return slice(args.start_index + gpu_idx * num_samples, args.start_index + (gpu_idx + 1) * num_samples)


def load_starting_points(gpu_idx, args, num_samples):
    """The rank's X_T / y (reference :45-57): the files under DIFFUSION_STARTING_POINTS, or - with --synthetic - generated the way the
    reference's scripts/generate_diffusion_starting_data.py generates them.  Missing files are an error that names them."""
    suffix = "_uvit" if (args.image_size == 128 and args.model_type == "uvit") else ""
    rng = index_range(gpu_idx, args, num_samples)
    lo, hi = rng.start, rng.stop
    folder = DIFFUSION_STARTING_POINTS / f"imagenet{args.image_size}{suffix}"
    if args.synthetic:
        shape = (4, args.image_size // 8, args.image_size // 8) if args.model_type == "uvit" else (3, args.image_size, args.image_size)
        X_T, y = starting_points(hi, shape, dataset_index=starting_point_dataset_index(f"imagenet{args.image_size}{suffix}"))
        return X_T[lo:hi], y[lo:hi]
    for name in ("y.pth", "X_T.pth"):
        if not (folder / name).exists():
            raise FileNotFoundError(f"{folder / name}: starting points not found (the reference's "
                                    f"scripts/generate_diffusion_starting_data.py writes them); pass --synthetic to "
                                    f"generate them the same way instead")
    y = torch.load(folder / "y.pth", map_location="cpu")[lo:hi]
    X_T = torch.load(folder / "X_T.pth", map_location="cpu")[lo:hi]
    return X_T, y


def build_denoiser(args, device):
    """(denoiser on the device, base scheduler) by --model-type / --image-size (reference :59-78), engine options applied."""
    ckpt = not args.random_weights
    if args.model_type == "unet":
        if args.image_size == 64:
            unet, scheduler = init_guided_diffusion_imagenet64_and_scheduler(load_checkpoint=ckpt, dropout=args.dropout)
        elif args.image_size == 128:
            unet, scheduler = init_guided_diffusion_imagenet128_and_scheduler(load_checkpoint=ckpt)
        else:
            raise ValueError(f"image_size {args.image_size} not supported for the ADM UNet")
        if not ckpt:
            randomize_parameters_(unet, seed=args.image_size)
        unet.eval().to(device)
        unet.precision = args.precision
        unet.on_range_violation = args.on_range_violation
        unet.split_k = args.split_k == "on" or (args.split_k == "auto" and args.batch_size <= 32)
        unet.winograd = args.winograd == "on"
        if args.hip_graph and args.scheduler_type != "mc_dropout":
            unet.enable_hip_graph()
        return unet, scheduler
    uvit = load_uvit(args.image_size, device="cpu", load_checkpoint=ckpt)
    if not ckpt:
        randomize_parameters_(uvit, seed=7)
    ae = load_autoencoder_uvit(device="cpu", load_checkpoint=ckpt)
    if not ckpt:
        randomize_parameters_(ae, seed=9)
    uvit.precision = "f16x3" if args.precision == "f16w" else args.precision             # f16w is a 3x3-conv mode: ADM only
    uvit.on_range_violation = ae.on_range_violation = args.on_range_violation
    uvit.split_k = args.split_k != "off"                                                 # (tail split of under-filled GEMM rounds)
    ae.precision = args.precision if args.precision in ("fp32", "f16x3") else "bf16x3"   # VAE decoder (one call per batch)
    return UViTAE(uvit, ae).to(device), load_uvit_scheduler()


@torch.no_grad()
def generate_and_save(gpu_idx, args, num_samples, dest):
    seed_everything(args.seed)
    device = torch.device("cuda", gpu_idx)
    torch.cuda.set_device(device)
    X_T, y = load_starting_points(gpu_idx, args, num_samples)
    model, scheduler = build_denoiser(args, device)
    uc_scheduler = get_uncertainty_scheduler(args, y.to(device), model, scheduler)
    if args.noise == "device":
        uc_scheduler.noise_source = DeviceNoise(seed=args.seed + gpu_idx)
        if args.model_type == "unet":
            model.dropout_mask_source = DeviceNoise(seed=args.seed + 1000 + gpu_idx).dropout_mask      # mc_dropout masks on the GPU
    if args.mc_group > 1:
        if not hasattr(uc_scheduler, "mc_group"):
            raise SystemExit(f"--mc-group: scheduler type {args.scheduler_type!r} has no stacked Monte-Carlo path")
        uc_scheduler.mc_group = args.mc_group
    uc_scheduler.set_timesteps(args.generation_steps)
    import time
    if args.check_range:
        from diffusion_uncertainty_amd import ops as _ops
        _ops.range_check(True)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    if args.model_type == "unet":
        results = generate_samples_model_scheduler_class_conditioned_from_tensor(X_T, y, args.batch_size, device, model, uc_scheduler)
    else:
        results = generate_samples_model_scheduler_class_conditioned_uvit_from_tensor(X_T, y, args.batch_size, model, uc_scheduler, device)
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if args.check_range:
        sat, sub = _ops.range_check_read()
        _ops.range_check(False)
        print(f"[rank {gpu_idx}] f16x3 range check: {sat} operand(s) beyond fp16's range after scaling (hi half saturated), {sub} below its "
              f"smallest normal (harmless)" + ("" if sat == 0 else " - the fp32-equivalent claim does NOT hold for this run: use --precision fp32"))
    if args.measure_time:
        print(f"[rank {gpu_idx}] {X_T.shape[0]} images in {elapsed:.2f} s -> {X_T.shape[0] / elapsed:.3f} images/s "
              f"({uc_scheduler.denoiser_calls} UC forwards + {args.generation_steps} main forwards per batch)")
    dest = Path(dest)
    torch.save(uc_scheduler.timesteps, dest / f"timestep_{gpu_idx}.pth")
    torch.save(results["score"], dest / f"score_{gpu_idx}.pth")
    if gpu_idx == 0:                    # the reference's shared names, as links to rank 0's files (no second copy of the score maps)
        for shared, own in (("timesteps.pth", f"timestep_{gpu_idx}.pth"), ("score.pth", f"score_{gpu_idx}.pth")):
            link = dest / shared
            if link.is_symlink() or link.exists():
                link.unlink()
            try:
                link.symlink_to(own)
            except OSError:             # a file system without symlinks: a plain second file
                torch.save(uc_scheduler.timesteps if shared.startswith("timesteps") else results["score"], link)
    torch.save(results["uncertainty"], dest / f"uncertainty_{args.scheduler_type}_{gpu_idx}.pth")
    torch.save(results["gen_images"], dest / f"gen_images_{gpu_idx}.pth")


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--num-samples", type=int, default=300, dest="num_samples")
    p.add_argument("--batch-size", type=int, default=32)
    p.add_argument("--generation-steps", type=int, default=20, dest="generation_steps")
    p.add_argument("-M", type=int, default=30, dest="M")
    p.add_argument("--start-step-uc", "--start-step", type=int, default=0, dest="start_step_uc")
    p.add_argument("--num-steps-uc", type=int, default=20, dest="num_steps_uc")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--eta", type=float, default=0.00)
    p.add_argument("--dropout", type=float, default=0.1)
    p.add_argument("--class-condition")
    p.add_argument("--image-size", type=int, choices=[64, 128, 256, 512], dest="image_size", default=None)
    p.add_argument("--scheduler-type", "--scheduler", type=str, default=None, dest="scheduler_type",
                   choices=["mc_dropout", "flip", "uncertainty", "flip_grad", "uncertainty_single", "uncertainty_single_score",
                            "uncertainty_centered", "uncertainty_centered_d", "uncertainty_image", "uncertainty_original",
                            "uncertainty_zigzag_centered", "uncertainty_fisher", "dpm_2_uncertainty_centered"])     # reference :161
    p.add_argument("--start-index", type=int, default=None, dest="start_index")
    p.add_argument("--multi-gpu", action="store_true", default=False)
    p.add_argument("--model-type", type=str, default=None, choices=["unet", "uvit"], dest="model_type")
    p.add_argument("--measure-time", action="store_true", dest="measure_time")
    p.add_argument("--predict-next", action="store_true", dest="predict_next")
    p.add_argument("--uncertainty-distance", type=int, default=20, dest="uncertainty_distance")
    p.add_argument("--num-zigzag", "--num-zigzags", "--num-zig-zag", "--num-zig-zags", type=int, default=3, dest="num_zigzag")
    # engine options (not in the reference)
    p.add_argument("--hip-graph", action="store_true", dest="hip_graph",
                   help="ADM UNet: replay the forward from a captured HIP graph (one launch instead of ~330; for small batches)")
    p.add_argument("--split-k", choices=["auto", "on", "off"], default="auto", dest="split_k",
                   help="f16x3: split the K range of conv / GEMM launches that cannot fill the 256 CUs (deterministic; results then depend on "
                        "the batch size at fp32-rounding level).  auto: on for batches of at most 32 images (2x on batch 4)")
    p.add_argument("--winograd", choices=["on", "off"], default="on",
                   help="f16x3: 3x3 convs in the Winograd F(2,3)-along-y form (6 instead of 9 matrix-core taps per pixel, same operand precision)")
    p.add_argument("--mc-group", type=int, default=1, dest="mc_group",
                   help="uncertainty_zigzag_centered: Monte-Carlo chains of a UC step per denoiser call (stacked batch, identical outputs; "
                        "helps U-ViT at batch 64: +4.6 %% with 5)")
    p.add_argument("--check-range", action="store_true", dest="check_range",
                   help="f16x3: count the operands that leave fp16's range after their static power-of-two scale (dua_range_check); "
                        "a non-zero saturation count means the run is not fp32-equivalent")
    p.add_argument("--on-range-violation", choices=["warn", "raise", "fp32"], default="warn", dest="on_range_violation",
                   help="f16x3: what happens when the first forward after the weight load finds operands beyond fp16's range after their static "
                        "scale (the default-on range check): warn, raise, or switch the denoiser to --precision fp32 and go on")
    p.add_argument("--precision", choices=["f16x3", "fp32", "f16w", "bf16x3"], default="f16x3",
                   help="f16x3 (default): the fp32-equivalent split - trajectories match the reference CPU path like the exact fp32 mode's; "
                        "fp32: exact fp32 MFMA; bf16x3: split-bf16 (16 operand bits); f16w: one fp16 product per tap in the Winograd convs (the "
                        "reference's fp16-autocast arithmetic class: NOT a parity mode)")
    p.add_argument("--noise", choices=["host", "device"], default="host", help="host: CPU-generator noise tape in the reference's draw order")
    p.add_argument("--synthetic", action="store_true", help="generate X_T / y like scripts/generate_diffusion_starting_data.py")
    p.add_argument("--random-weights", action="store_true", help="seeded random weights instead of checkpoints")
    p.add_argument("--dest", type=str, default=None)
    options = [x.basename().replace(".yaml", "") for x in CONFIG.joinpath("generation").files("*.yaml")]
    p.add_argument("--config", type=str, default=None, choices=options, help="name of a config/generation/*.yaml file")
    args = p.parse_args(argv)
    if args.config is not None:
        p.set_defaults(**load_config(CONFIG / "generation", args.config).__dict__)
        args = p.parse_args(argv)
    else:
        assert args.image_size is not None, "--image-size must be specified"
        assert args.scheduler_type is not None, "--scheduler-type must be specified"
        assert args.model_type is not None, "--model-type must be specified"
        assert args.start_index is not None, "--start-index must be specified"
    return args


def main():
    args = parse_args()
    if args.scheduler_type == "mc_dropout":
        assert args.image_size == 64, "mc_dropout only implemented for image_size=64"
    seed_everything(args.seed)
    dest = Path(args.dest) if args.dest else RESULTS / "score-uncertainty" / datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S")
    dest.mkdir(parents=True, exist_ok=True)
    args.__dict__["dataset"] = f"imagenet{args.image_size}"
    with open(dest / "args.yaml", "w") as f:
        yaml.dump(args.__dict__, f)
    num_gpus = torch.cuda.device_count()
    if num_gpus == 0:
        raise SystemExit("no GPU visible: the engine has no CPU path")
    if num_gpus == 1:
        generate_and_save(0, args, args.num_samples, dest)
    else:
        torch.multiprocessing.spawn(fn=generate_and_save, args=(args, args.num_samples // num_gpus, dest), nprocs=num_gpus, join=True)
    print("Done:", dest)


if __name__ == "__main__":
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    main()
