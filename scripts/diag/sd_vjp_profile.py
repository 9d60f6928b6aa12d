#!/usr/bin/env python3
"""Time one taped forward + input-gradient pass of the SD-1.5 latent UNet at the batch of a guided step (2M = 10 rows)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from diffusion_uncertainty_amd.stable_diffusion.unet_2d_condition import UNet2DConditionModel  # noqa: E402
from diffusion_uncertainty_amd.utils import randomize_parameters_  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
precision = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
unet = randomize_parameters_(UNet2DConditionModel(), seed=15).eval().to("cuda")
unet.precision = precision
unet.backward_precision = sys.argv[3] if len(sys.argv) > 3 else "fp32"
g = torch.Generator().manual_seed(1)
x, enc, go = torch.randn(B, 4, 64, 64, generator=g).cuda(), torch.randn(B, 77, 768, generator=g).cuda(), torch.randn(B, 4, 64, 64, generator=g).cuda()
t = torch.tensor(500)
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, tape = unet.forward_with_tape(x, t, enc)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    grad = unet.backward_sample(tape, go)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"batch {B} {precision} bwd {unet.backward_precision}: taped forward {1e3 * (t1 - t0):.1f} ms, backward {1e3 * (t2 - t1):.1f} ms", flush=True)
