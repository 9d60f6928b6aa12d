"""Posterior-distribution uncertainty guidance for the class-conditional models, on the HIP library.

Mirror of reference pipeline_uncertainty/pipeline_sampler_class_conditional_uncertainty_guided_posterior_distribution.py:
``calculate_threshold_map`` :10-30, ``estimate_score_update_posterior`` :32-66 and
``DiffusionClassConditionalGuidedPosteriorDistribution`` :69-269 (loop :132-154).  Per guided step: M re-noised
denoiser calls around pred_x0, unbiased variance over the M+1 predictions (dua_mc_reduce_var), threshold map - per-sample
quantile (dua_quantile_mask -> dua_threshold_map, bit-exact with torch.quantile) or the dataset-level pixel-wise
thresholds of ``thresholds.py`` - posterior score (dua_batch_sum + dua_posterior_score), masked blend (dua_map_blend)
and a second DDIM step with the blended score.

Reproduced as they are upstream: ``alphas_cumprod[i]`` is indexed by the LOOP index, not the timestep (:146);
``pred_epsilon.sum(dim=0)`` in the posterior mean reduces the LAST re-noised prediction over the BATCH axis (:65); the
guided window ``start_step <= i <= start_step + num_steps`` is inclusive on both ends (:148).
One upstream defect is not reproduced: ``__call__`` invokes the module-level ``calculate_threshold_map`` without its
required ``threshold_type`` (:150, a TypeError) while the script passes ``threshold_type=...`` to a constructor that
does not take it (scripts/generate_images_with_uncertainty_threshold.py:215); here the constructor accepts
``threshold_type`` (default 'higher', the behaviour of the class's own method :230-238) and forwards it.
"""
import math
from typing import Dict, Optional

import numpy as np
import torch

from .. import ops
from ..generate_samples import _require_gpu, predict_model
from ..schedulers_uncertainty.noise import HostNoise
from ..uvit.uvit_ae import UViTAE
from . import _core

noise_source = HostNoise()          # torch.randn_like(prev_noisy_sample) :57; swap for DeviceNoise to keep the draws on the GPU


def calculate_threshold_map(threshold, i, pixel_wise_uncertainty, threshold_type):
    """:10-30 -> float 0/1 map shaped like ``pixel_wise_uncertainty``."""
    assert threshold_type in ("higher", "lower"), threshold_type
    u = ops._f32c(pixel_wise_uncertainty)
    lower = threshold_type == "lower"
    if isinstance(threshold, float):
        _mask, q = ops.quantile_mask(u, threshold)                       # torch.quantile(u.flatten(1).float(), p, dim=1) :15
        return ops.threshold_map(u, q, per_sample=True, lower=lower)
    threshold_i = threshold[i]
    if u.dim() == 4:
        if tuple(threshold_i.shape) != tuple(u.shape[1:]):
            raise RuntimeError(f"threshold[{i}] has shape {tuple(threshold_i.shape)}, uncertainty {tuple(u.shape)}")
        return ops.threshold_map(u, threshold_i.to(u.device, torch.float32).contiguous(), per_sample=False, lower=lower)
    if tuple(threshold_i.shape) != tuple(u.shape):
        raise RuntimeError(f"threshold[{i}] has shape {tuple(threshold_i.shape)}, uncertainty {tuple(u.shape)}")
    return ops.threshold_map(u.unsqueeze(0), threshold_i.to(u.device, torch.float32).contiguous(), per_sample=False, lower=lower)[0]


def estimate_score_update_posterior(M: int, model, scheduler, input, y_slice, t_tensor, noisy_residual, prev_noisy_sample,
                                    alpha_hat_t):
    """:32-66 -> (pixel_wise_uncertainty, new_score)."""
    a32 = np.float32(float(alpha_hat_t))
    sb, sa = math.sqrt(float(np.float32(1) - a32)), math.sqrt(float(a32))       # math.sqrt on 0-dim fp32 tensors :56-57
    eps = ops._f32c(noisy_residual)
    x = ops._f32c(input)
    _, pred_x_0 = ops.ddim_step(eps, x, sb, sa, 0.0, 0.0, clip=False)           # (input - sqrt(1-a) eps) / sqrt(a) :55
    pred_epsilons = []
    pred_epsilon = None
    for _ in range(M):
        z = noise_source.draw(prev_noisy_sample.shape, x.device)
        x_hat_t = ops.axpby(pred_x_0, z, sa, sb)                                # :57
        pred_epsilon = ops._f32c(predict_model(model, x_hat_t, t_tensor, y_slice))
        pred_epsilons.append(pred_epsilon)
    pred_epsilons.append(eps)                                                   # :60
    pixel_wise_uncertainty = ops.mc_reduce_var(torch.stack(pred_epsilons, dim=0))   # torch.var(unbiased) :63
    inv_alpha = float(np.float32(1) / a32)                                      # (1 / alpha_hat_t) :64
    new_score = ops.posterior_score(pixel_wise_uncertainty, ops.batch_sum(pred_epsilon), M, inv_alpha)    # :62-65
    return pixel_wise_uncertainty, new_score


class DiffusionClassConditionalGuidedPosteriorDistribution:
    """Same constructor / call surface as the reference class (:69-175); the loops are the shared skeleton of ``_core`` and the
    guided update is the ``revise`` callback of the timestep loop."""

    def __init__(self, model, scheduler, threshold, image_size: int, device, batch_size: int, init_seed_rng: int,
                 fid_evaluator: Optional[object] = None, M: int = 5, threshold_type: str = "higher"):
        assert isinstance(threshold, (torch.Tensor, float)), "Threshold must be a tensor or a float"
        if isinstance(threshold, float):
            assert 0 <= threshold <= 1, "Threshold percentile must be between 0 and 1"
        _require_gpu(device)
        self.model, self.scheduler, self.threshold, self.device = model, scheduler, threshold, device
        self.image_size, self.batch_size, self.init_seed_rng = image_size, batch_size, init_seed_rng
        self.fid_evaluator, self.M, self.threshold_type = fid_evaluator, M, threshold_type
        self.is_uvit = isinstance(model, UViTAE)
        self.lambda_update = 7

    @torch.no_grad()
    def __call__(self, num_samples: Optional[int] = None, num_classes: Optional[int] = None,
                 X_T: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None, start_step: int = 0,
                 num_steps: Optional[int] = None) -> Dict[str, torch.Tensor]:
        sched = self.scheduler
        n_t = sched.timesteps.shape[0]
        if num_steps is None:
            num_steps = n_t - start_step
        if isinstance(self.threshold, torch.Tensor):
            assert self.threshold.shape[0] == n_t, f"{self.threshold.shape=} {sched.timesteps.shape=}"
        sched.config.after_step, sched.config.num_steps_uc = start_step, num_steps          # :117-119
        sched.set_timesteps(n_t)
        yb_now = [None]

        def guided(i, t, t_tensor, x, eps, out):
            if not (start_step <= i <= start_step + num_steps):                               # inclusive on both ends (:148)
                return None
            eps = ops._f32c(eps)
            u, post = self.estimate_score_update(x, yb_now[0], i, t_tensor, eps, out.prev_sample, sched.alphas_cumprod[i])   # loop index (:146)
            tmap = calculate_threshold_map(self.threshold, i, u, self.threshold_type)
            return sched.step(ops.map_blend(post, eps, tmap), t, x)                           # :152-154

        def chain(xb, yb):
            yb_now[0] = yb
            x, _, _ = _core.sample_chain(self.model, sched, xb, yb, self.device, revise=guided)
            return _core.to_images(x), {}                                                     # (no VAE decode here upstream either)

        return _core.run_pipeline(self, X_T, y, num_samples, num_classes, chain)

    def calculate_threshold_map(self, i, pixel_wise_uncertainty):
        """:230-238 (always 'higher')."""
        return calculate_threshold_map(self.threshold, i, pixel_wise_uncertainty, "higher")

    def estimate_score_update(self, input, y_slice, i, t_tensor, noisy_residual, prev_noisy_sample, alpha_hat_t):
        """:240-267 - the module-level function with M = self.M."""
        return estimate_score_update_posterior(self.M, self.model, self.scheduler, input, y_slice, t_tensor, noisy_residual,
                                               prev_noisy_sample, alpha_hat_t)
