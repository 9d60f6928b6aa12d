#!/usr/bin/env python3
"""BASELINE configs[4] leg alone (SD-1.5 latent UNet, posterior update), eager or graph, for profiling: sd_job.py [precision] [graph]."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from diffusion_uncertainty_amd import uncertainty_guidance  # noqa: E402
from diffusion_uncertainty_amd.pipeline_uncertainty.pipeline_stable_diffusion_uncertainty_guided import StableDiffusionPipelineUncertainty  # noqa: E402
from diffusion_uncertainty_amd.schedulers_uncertainty.noise import DeviceNoise  # noqa: E402
from diffusion_uncertainty_amd.schedulers_uncertainty.scheduling_pndm import PNDMScheduler  # noqa: E402
from diffusion_uncertainty_amd.stable_diffusion.unet_2d_condition import UNet2DConditionModel  # noqa: E402
from diffusion_uncertainty_amd.utils import randomize_parameters_  # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
unet = randomize_parameters_(UNet2DConditionModel(), seed=15).eval().to("cuda")
unet.precision = precision
for kv in sys.argv[3:]:                      # e.g. fold_temb=0 cache_context=0 split_k=0
    k_, v_ = kv.split("=")
    setattr(unet, k_, bool(int(v_)))
if graph:
    unet.enable_hip_graph()
g = torch.Generator().manual_seed(491)
pe, ne = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
pipe = StableDiffusionPipelineUncertainty(unet, PNDMScheduler())
uncertainty_guidance.noise_source = DeviceNoise(seed=491)
kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=20, guidance_scale=7.5, start_step_uc=0, num_steps_uc=2,
          percentile=0.9, num_uncertainty_samples=5, height=512, width=512)
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe(generator=torch.Generator().manual_seed(491), **kw)
    torch.cuda.synchronize()
    print(f"{precision} graph={graph}: {time.perf_counter() - t0:.3f} s", flush=True)
