#!/usr/bin/env python3
"""Uncertainty-guided image generation for the class-conditional models (counterpart of the reference's
scripts/generate_images_with_uncertainty_threshold.py on the MI355X engine).

Same sampling flags and the same flow as the reference's ``main`` (:118-306): model and base scheduler by dataset name
(``instantiate_model_scheduler``), thresholds from ``RESULTS/thresholds/<dataset>/thresholds_<scheduler_type>_perc=<percentile>.pth`` (written by
scripts/compute_threshold_pixel_wise.py) or, with ``--use-percentile``, the per-sample quantile of each guided step; starting points
``X_T[start : start + n]`` / ``y`` from ``DIFFUSION_STARTING_POINTS/<dataset>``; one plain DDIM run (``DiffusionClassConditional``) and one
guided run (``DiffusionClassConditionalGuidedPosteriorDistribution``) on the SAME ``num_steps // 4`` timesteps (:213,217 - as upstream);
``gen_images.pth`` / ``gen_images_threshold.pth`` / ``args.yaml`` / ``config.yaml`` under
``RESULTS/uncertainty_guidance/<dataset>_<scheduler_type>_perc=<percentile>/<date>`` and one record appended to ``results.json``.

Outside the accelerated path and therefore not here: the FID scores (InceptionV3 statistics; ``--checkpoint``
re-scoring), and ``--guidance-type gradient`` / ``second_order`` (autograd through the ADM / U-ViT denoiser) - both raise.  The default
``--guidance-type`` is therefore ``posterior``.  ``--synthetic`` / ``--random-weights`` run without the starting-point files / checkpoints.
"""
import argparse
import datetime
import json
import sys
from pathlib import Path

import torch
import yaml

ROOT_DIR = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT_DIR))

from diffusion_uncertainty_amd.init_model import instantiate_model_scheduler  # noqa: E402
from diffusion_uncertainty_amd.paths import CONFIG, DIFFUSION_STARTING_POINTS, RESULTS, THRESHOLD  # noqa: E402
from diffusion_uncertainty_amd.pipeline_uncertainty.pipeline_sampler_class_conditional import DiffusionClassConditional  # noqa: E402
from diffusion_uncertainty_amd.pipeline_uncertainty.pipeline_sampler_class_conditional_uncertainty_guided_posterior_distribution import \
    DiffusionClassConditionalGuidedPosteriorDistribution  # noqa: E402
from diffusion_uncertainty_amd.schedulers_uncertainty.scheduling_ddim import DDIMScheduler  # noqa: E402
from diffusion_uncertainty_amd.utils import (load_config, randomize_parameters_, seed_everything, starting_point_dataset_index,  # noqa: E402
                                             starting_points)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--num-samples", type=int, default=1000, help="number of samples to generate")
    p.add_argument("--batch-size", type=int, default=128)
    p.add_argument("--on-cpu", action="store_true", dest="on_cpu")
    p.add_argument("--scheduler-type", "--scheduler", dest="scheduler_type", type=str, default="uncertainty")
    p.add_argument("--num-steps", default=20, type=int, dest="num_steps", help="diffusion steps (the samplers run on num_steps // 4 of them, as upstream)")
    p.add_argument("--dataset-folder", "--dataset", type=str, required=False, dest="dataset_name", help="dataset name: imagenet64 | imagenet128 | imagenet256 | imagenet512")
    p.add_argument("--start-step-threshold", "--start-step", "--start-step-guidance", type=int, default=0, dest="start_step_guidance")
    p.add_argument("--num-steps-threshold", "--num-steps-uc", "--num-steps-guidance", type=int, default=20, dest="num_steps_guidance")
    p.add_argument("--start-index", type=int, default=0, help="starting index of the samples")
    p.add_argument("--guidance-type", type=str, default="posterior", dest="guidance_type", choices=["gradient", "posterior", "second_order"],
                   help="posterior (built); gradient / second_order need autograd through the denoiser and raise (the reference's default is gradient)")
    p.add_argument("--seed", type=int, default=491, help="random seed")
    p.add_argument("--percentile", type=float, default=0.95, dest="percentile")
    p.add_argument("--skip-save", action="store_true", dest="skip_save")
    p.add_argument("--checkpoint", "--checkpoint-path", type=str, default=None, dest="checkpoint_path")
    p.add_argument("--use-percentile", action="store_true", help="the per-sample percentile instead of saved thresholds")
    p.add_argument("--skip-ddim", action="store_true", help="skip the plain DDIM run")
    p.add_argument("--gradient-type", "--wrt", "--gradient-wrt", type=str, default="input", choices=["input", "score"], dest="gradient_wrt")
    p.add_argument("--gradient-direction", type=str, choices=["ascend", "descend"], default="descend", dest="gradient_direction")
    p.add_argument("--lambda", "--lambda-update", type=float, default=0.1, dest="lambda_update")
    p.add_argument("--threshold-type", type=str, choices=["higher", "lower"], default="higher", dest="threshold_type")
    # engine options (not in the reference)
    p.add_argument("--precision", choices=["f16x3", "fp32"], default="f16x3")
    p.add_argument("--synthetic", action="store_true", help="generate X_T / y like scripts/generate_diffusion_starting_data.py")
    p.add_argument("--random-weights", action="store_true", help="seeded random weights instead of checkpoints")
    p.add_argument("--dest", type=str, default=None)
    p.add_argument("--config", type=str, help="name of a config/guidance/*.yaml file")
    args = p.parse_args(argv)
    if args.config is not None:
        print("Loading config file - terminal arguments take priority")
        p.set_defaults(**load_config(CONFIG / "guidance", args.config).__dict__)
        args = p.parse_args(argv)
    assert args.dataset_name, "--dataset must be given"
    return args


@torch.no_grad()
def main(argv=None):
    args = parse_args(argv)
    print("Arguments:", args.__dict__)
    if args.on_cpu or not torch.cuda.is_available():
        raise SystemExit("the engine has no CPU path: a GPU is needed")
    if args.checkpoint_path is not None:
        raise SystemExit("--checkpoint re-scores saved images with the FID (InceptionV3): outside the accelerated path (SURVEY.md section 2)")
    if args.guidance_type != "posterior":
        raise NotImplementedError(f"--guidance-type {args.guidance_type}: needs autograd through the ADM / U-ViT denoiser (outside the "
                                  "accelerated hot path); the posterior update is built")
    device = torch.device("cuda")
    seed = args.seed
    seed_everything(seed)
    model, scheduler = instantiate_model_scheduler(args.dataset_name, load_checkpoint=not args.random_weights)
    if args.random_weights:
        randomize_parameters_(model, seed=int(args.dataset_name.replace("imagenet", "")))
    model = model.eval().to(device)
    image_size = int(args.dataset_name.replace("imagenet", ""))
    latent = hasattr(model, "uvit")                                      # U-ViT: the samplers run on the KL-VAE's latents
    if latent:
        model.uvit.precision = model.autoencoder.precision = args.precision
    else:
        model.precision = args.precision
    if not args.use_percentile:
        thresholds = torch.load(THRESHOLD / args.dataset_name / f"thresholds_{args.scheduler_type}_perc={args.percentile}.pth", map_location=device)
        with open(THRESHOLD / args.dataset_name / f"config_{args.scheduler_type}_perc={args.percentile}.yaml") as f:
            config_generation = yaml.safe_load(f)["dataset_config"]
        for k, v in config_generation.items():                           # the dataset's own generation arguments fill what the CLI left open (:172-177)
            if k not in args.__dict__:
                setattr(args, k, v)
    else:
        thresholds = args.percentile
    generation_steps = args.num_steps
    lo, hi = args.start_index, args.start_index + args.num_samples
    if args.synthetic:
        x_T, y = starting_points(hi, (4, image_size // 8, image_size // 8) if latent else (3, image_size, image_size),
                                 dataset_index=starting_point_dataset_index(args.dataset_name))
        x_T, y = x_T[lo:hi], y[lo:hi]
    else:
        folder = DIFFUSION_STARTING_POINTS / args.dataset_name
        for name in ("X_T.pth", "y.pth"):
            if not (folder / name).exists():
                raise FileNotFoundError(f"{folder / name}: starting points not found; pass --synthetic to generate them instead")
        x_T = torch.load(folder / "X_T.pth", map_location="cpu")[lo:hi]
        y = torch.load(folder / "y.pth", map_location="cpu")[lo:hi]
    x_T, y = x_T.to(device), y.to(device)
    ddim_sampler = DDIMScheduler.from_config(scheduler.config)
    ddim_sampler.set_timesteps(generation_steps)
    pipeline_sampler_ddim = DiffusionClassConditional(model, ddim_sampler, image_size, device, args.batch_size, seed)
    print("Using posterior distribution guidance")
    pipeline_sampler_threshold = DiffusionClassConditionalGuidedPosteriorDistribution(model, ddim_sampler, thresholds, image_size, device, args.batch_size, seed,
                                                                                     threshold_type=args.threshold_type)
    ddim_sampler.set_timesteps(generation_steps // 4)                    # (:213 - both runs use a quarter of --num-steps, as upstream)
    output = None
    if not args.skip_ddim:
        output = pipeline_sampler_ddim(X_T=x_T, y=y)
    print(f"using {generation_steps // 4} steps for DDIM with guidance")
    ddim_sampler.set_timesteps(generation_steps // 4)
    output_threshold = pipeline_sampler_threshold(X_T=x_T, y=y, start_step=args.start_step_guidance, num_steps=args.num_steps_guidance)
    config = {"dataset": args.dataset_name, "scheduler_type": args.scheduler_type, "num_samples": args.num_samples, "seed": seed,
              "start_step_threshold": args.start_step_guidance, "num_steps_threshold": args.num_steps_guidance, "start_index": args.start_index,
              "percentile": args.percentile, "use_percentile": args.use_percentile, "guidance_type": args.guidance_type}
    storage = Path(args.dest) if args.dest else Path(RESULTS) / "uncertainty_guidance"
    storage.mkdir(parents=True, exist_ok=True)
    results_guidance = dict(config, fid_score=None, fid_score_guidance=None, note="FID is outside the accelerated path: not computed")
    if not args.skip_save and not args.skip_ddim:
        images = storage / f"{args.dataset_name}_{args.scheduler_type}_perc={args.percentile}" / datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S")
        images.mkdir(parents=True, exist_ok=True)
        with open(images / "args.yaml", "w") as f:
            yaml.safe_dump({k: v for k, v in args.__dict__.items() if isinstance(v, (int, float, str, bool, type(None)))}, f)
        with open(images / "config.yaml", "w") as f:
            yaml.safe_dump(config, f)
        torch.save(output_threshold["gen_images"], images / "gen_images_threshold.pth")
        torch.save(output["gen_images"], images / "gen_images.pth")
        print("Saved gen_images_threshold to", images / "gen_images_threshold.pth")
        print("Saved gen_images to", images / "gen_images.pth")
        results_guidance["folder"] = str(images)
    results_path = storage / "results.json"
    results = json.load(open(results_path)) if results_path.exists() else []
    results.append(results_guidance)
    with open(results_path, "w") as f:
        json.dump(results, f, indent=4)
    return output, output_threshold


if __name__ == "__main__":
    main()
