#!/usr/bin/env python3
"""Counterpart of reference scripts/generate_with_uncertainty_threshold_stable_diffusion.py on the HIP library (BASELINE
configs[4]: SD-1.5 latent UNet 512x512, uncertainty-threshold guidance, percentile 0.9, 20 steps, M = 5).

Same sampling flags as the reference script (--num-steps, --start-step-threshold, --num-steps-threshold, --percentile,
--use-posterior, --seed).  What is outside the accelerated path is replaced by inputs: there is no network for the
``runwayml/stable-diffusion-v1-5`` checkpoint and the CLIP text encoder / VAE are third-party models that are not built, so
``--synthetic`` (default) runs seeded random-init UNet weights on seeded random prompt embeddings and returns latents; with
``--unet-state-dict`` a real diffusers UNet state dict is loaded strictly, ``--prompt-embeds`` takes a saved
{'prompt_embeds', 'negative_prompt_embeds'} file.  As in the reference, gradient-mode guidance is the default
(uncertainty_guidance.use_posterior = args.use_posterior, reference script :16-17) and ``--use-posterior`` selects the posterior
update; the gradient is an explicit vector-Jacobian product on the HIP library (UNet2DConditionModel.backward_sample).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--num-steps", type=int, default=20)
    p.add_argument("--seed", type=int, default=491)
    p.add_argument("--start-step-threshold", type=int, default=0)
    p.add_argument("--num-steps-threshold", type=int, default=2)
    p.add_argument("--percentile", "--perc", "-p", type=float, default=0.9, dest="percentile")
    p.add_argument("--use-posterior", action="store_true", help="posterior update instead of the gradient update (reference script :88)")
    p.add_argument("--lr", type=float, default=1.0, help="step of the gradient update (reference pipeline argument lr)")
    p.add_argument("-M", type=int, default=5)
    p.add_argument("--guidance-scale", type=float, default=7.5)
    p.add_argument("--precision", choices=["f16x3", "bf16x3", "fp32"], default="f16x3")
    p.add_argument("--backward-precision", choices=["fp32", "bf16x3"], default="fp32",
                   help="products of the input-gradient pass of the gradient update: exact fp32 MFMA, or split-bf16 convs / linears")
    p.add_argument("--unet-state-dict", type=str, default=None)
    p.add_argument("--prompt-embeds", type=str, default=None)
    p.add_argument("--out", type=str, default=None)
    p.add_argument("--no-graph", action="store_true", help="launch every kernel from the host instead of replaying a captured HIP graph")
    p.add_argument("--repeats", type=int, default=2, help="timed runs after one warm-up run")
    a = p.parse_args()
    from diffusion_uncertainty_amd import uncertainty_guidance
    from diffusion_uncertainty_amd.pipeline_uncertainty.pipeline_stable_diffusion_uncertainty_guided import StableDiffusionPipelineUncertainty
    from diffusion_uncertainty_amd.schedulers_uncertainty.noise import DeviceNoise
    from diffusion_uncertainty_amd.schedulers_uncertainty.scheduling_pndm import PNDMScheduler
    from diffusion_uncertainty_amd.stable_diffusion.unet_2d_condition import UNet2DConditionModel
    from diffusion_uncertainty_amd.utils import randomize_parameters_, seed_everything
    uncertainty_guidance.use_posterior = a.use_posterior
    seed_everything(a.seed)
    unet = UNet2DConditionModel()
    if a.unet_state_dict:
        unet.load_state_dict(torch.load(a.unet_state_dict, map_location="cpu"), strict=True)
    else:
        randomize_parameters_(unet, seed=15)
    unet.eval().to("cuda")
    unet.precision = a.precision
    unet.backward_precision = a.backward_precision
    if not a.no_graph:
        unet.enable_hip_graph()
    if a.prompt_embeds:
        d = torch.load(a.prompt_embeds, map_location="cpu")
        pe, ne = d["prompt_embeds"], d["negative_prompt_embeds"]
    else:
        g = torch.Generator().manual_seed(a.seed)
        pe, ne = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
    pipe = StableDiffusionPipelineUncertainty(unet, PNDMScheduler())
    uncertainty_guidance.noise_source = DeviceNoise(seed=a.seed)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=a.num_steps, guidance_scale=a.guidance_scale,
              start_step_uc=a.start_step_threshold, num_steps_uc=a.num_steps_threshold, percentile=a.percentile,
              num_uncertainty_samples=a.M, height=512, width=512, lr=a.lr)
    times = []
    for r in range(a.repeats + 1):
        g = torch.Generator().manual_seed(a.seed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pipe(generator=g, **kw)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    best = min(times[1:]) if a.repeats else times[0]
    print(f"SD-1.5 latent UNet 512x512, {a.num_steps} PLMS steps (+1 repeated), guided steps {a.num_steps_threshold} x M={a.M}, "
          f"percentile {a.percentile}, {'posterior' if a.use_posterior else 'gradient'} update, precision {a.precision}: {best:.3f} s per image ({1.0 / best:.3f} images/s; "
          f"{a.num_steps + 1} CFG UNet calls of batch 2 + {a.num_steps_threshold} Monte-Carlo calls of batch {2 * a.M}"
          f"{'' if a.use_posterior else ' with their input-gradient passes'}); "
          f"first run incl. weight packing {times[0]:.2f} s")
    if a.out:
        torch.save(out.images.cpu(), a.out)


if __name__ == "__main__":
    main()
