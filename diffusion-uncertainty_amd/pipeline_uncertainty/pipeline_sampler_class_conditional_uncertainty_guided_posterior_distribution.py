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

    def __init__(self, model, scheduler, threshold, image_size: int, device, batch_size: int, init_seed_rng: int,
                 fid_evaluator: Optional[object] = None, M: int = 5, threshold_type: str = "higher"):
        assert isinstance(threshold, (torch.Tensor, float)), "Threshold must be a tensor or a float"
        if isinstance(threshold, float):
            assert 0 <= threshold <= 1, "Threshold percentile must be between 0 and 1"
        _require_gpu(device)
        self.model = model
        self.scheduler = scheduler
        self.threshold = threshold
        self.image_size = image_size
        self.device = device
        self.fid_evaluator = fid_evaluator
        self.batch_size = batch_size
        self.is_uvit = isinstance(model, UViTAE)
        self.is_cifar10 = False
        self.init_seed_rng = init_seed_rng
        self.M = M
        self.lambda_update = 7
        self.threshold_type = threshold_type

    @torch.no_grad()
    def __call__(self, num_samples: Optional[int] = None, num_classes: Optional[int] = None,
                 X_T: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None, start_step: int = 0,
                 num_steps: Optional[int] = None) -> Dict[str, torch.Tensor]:
        assert num_samples is not None or X_T is not None, "Either num_samples or X_T must be provided"
        assert num_classes is not None or y is not None, "Either num_classes or y must be provided"
        num_generated_samples = 0
        samples_x_t, samples_y, samples_gen_images = [], [], []
        if num_steps is None:
            num_steps = self.scheduler.timesteps.shape[0] - start_step
        if num_samples is None:
            num_samples = X_T.shape[0]
        if isinstance(self.threshold, torch.Tensor):
            assert self.threshold.shape[0] == self.scheduler.timesteps.shape[0], \
                f"{self.threshold.shape=} {self.scheduler.timesteps.shape=}"
        self.scheduler.config.after_step = start_step                     # :117-119
        self.scheduler.config.num_steps_uc = num_steps
        self.scheduler.set_timesteps(len(self.scheduler.timesteps))
        generator = torch.Generator(device=self.device)
        i_batch = 0
        while num_samples > num_generated_samples:
            input = self.get_X_T_batch(X_T, num_generated_samples, i_batch, generator).to(self.device)
            samples_x_t.append(input.cpu().clone())
            y_slice = self.get_y_batch(num_classes, y, num_generated_samples, i_batch, generator, input).to(self.device)
            samples_y.append(y_slice)
            self.scheduler.prompt_embeds = y_slice
            for i, t in enumerate(self.scheduler.timesteps):
                t = t.item()
                t_tensor = torch.full((y_slice.shape[0],), t, device=self.device, dtype=torch.long)
                noisy_residual = ops._f32c(predict_model(self.model, input, t_tensor, y_slice))
                output = self.scheduler.step(noisy_residual, t, input)
                prev_noisy_sample = output.prev_sample
                alpha_hat_t = self.scheduler.alphas_cumprod[i]            # loop index, as upstream (:146)
                if (start_step + num_steps) >= i >= start_step:
                    pixel_wise_uncertainty, posterior_score = self.estimate_score_update(
                        input, y_slice, i, t_tensor, noisy_residual, prev_noisy_sample, alpha_hat_t)
                    thresholded_map = calculate_threshold_map(self.threshold, i, pixel_wise_uncertainty, self.threshold_type)
                    noisy_residual = ops.map_blend(posterior_score, noisy_residual, thresholded_map)     # :152
                    output = self.scheduler.step(noisy_residual, t, input)
                    prev_noisy_sample = output.prev_sample
                input = prev_noisy_sample
            gen_images = ops.to_uint8(ops._f32c(input))                   # :157-170 (no VAE decode here upstream either)
            num_generated_samples += gen_images.shape[0]
            if self.fid_evaluator is not None:
                self.fid_evaluator.update(gen_images, real=False)
            samples_gen_images.append(gen_images)
            i_batch += 1
        results = {"y": torch.cat(samples_y, dim=0).cpu(),
                   "x_t": torch.cat(samples_x_t, dim=0).cpu(),
                   "timestep": self.scheduler.timesteps,
                   "gen_images": torch.cat(samples_gen_images, dim=0).cpu()}
        if self.fid_evaluator is not None:
            results["fid"] = self.fid_evaluator.compute()
        return results

    def calculate_threshold_map(self, i, pixel_wise_uncertainty):
        """:230-238 (always 'higher')."""
        return calculate_threshold_map(self.threshold, i, pixel_wise_uncertainty, "higher")

    def estimate_score_update(self, input, y_slice, i, t_tensor, noisy_residual, prev_noisy_sample, alpha_hat_t):
        """:240-267 - the module-level function with M = self.M."""
        return estimate_score_update_posterior(self.M, self.model, self.scheduler, input, y_slice, t_tensor, noisy_residual,
                                               prev_noisy_sample, alpha_hat_t)

    def get_y_batch(self, num_classes, y, num_generated_samples, i_batch, generator, X_t):
        if y is not None:
            return y[num_generated_samples:num_generated_samples + self.batch_size]
        return torch.randint(0, num_classes, (self.batch_size,), device=self.device,
                             generator=generator.manual_seed(self.init_seed_rng + i_batch))

    def get_X_T_batch(self, X_T, num_generated_samples, i_batch, generator):
        if X_T is not None:
            return X_T[num_generated_samples:num_generated_samples + self.batch_size]
        c = 4 if self.is_uvit else 3
        return torch.randn(self.batch_size, c, self.image_size, self.image_size, device=self.device, dtype=torch.float32,
                           generator=generator.manual_seed(self.init_seed_rng + i_batch))

    def predict_score(self, input, y_batch, t_tensor):
        return predict_model(self.model, input, t_tensor, y_batch)
