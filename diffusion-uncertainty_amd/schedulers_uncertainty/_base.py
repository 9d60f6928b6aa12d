"""Shared machinery of the uncertainty DDIM schedulers (host side).

Mirrors the parts the three reference files have in common
(schedulers_uncertainty/scheduling_ddim_uncertainty_zigzag_centered.py): schedule tables ``__init__``
:218-244, ``set_timesteps`` :338-387, ``scale_model_input`` :277-292, ``_get_variance`` :294-301,
DDIM part of ``step`` :445-525, ``add_noise`` :575-604, ``get_velocity`` :607-624, ``__len__`` :626.
Per-step coefficients are evaluated as 0-dim fp32 CPU tensors exactly as the reference writes them; all
tensor arithmetic runs in libdua_hip.so (dua_ddim_step, dua_axpby, dua_mc_*).
"""
import math

import numpy as np
import torch

from .. import ops
from ..configuration import BaseOutput, ConfigMixin, KarrasDiffusionSchedulers, SchedulerMixin
from .noise import HostNoise


class DDIMSchedulerUncertaintyOutput(BaseOutput):
    """prev_sample, pred_original_sample, uncertainty (+ pred_epsilon) - zigzag_centered.py:37-54."""

    def __init__(self, prev_sample=None, pred_original_sample=None, uncertainty=None, **kw):
        super().__init__(prev_sample=prev_sample, pred_original_sample=pred_original_sample, uncertainty=uncertainty, **kw)


def betas_for_alpha_bar(num_diffusion_timesteps, max_beta=0.999, alpha_transform_type="cosine"):
    """zigzag_centered.py:58-99 (Glide cosine schedule) -> fp32 tensor."""
    if alpha_transform_type == "cosine":
        def alpha_bar_fn(t):
            return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    elif alpha_transform_type == "exp":
        def alpha_bar_fn(t):
            return math.exp(t * -12.0)
    else:
        raise ValueError(f"Unsupported alpha_tranform_type: {alpha_transform_type}")
    betas = []
    for i in range(num_diffusion_timesteps):
        t1, t2 = i / num_diffusion_timesteps, (i + 1) / num_diffusion_timesteps
        betas.append(min(1 - alpha_bar_fn(t2) / alpha_bar_fn(t1), max_beta))
    return torch.tensor(betas, dtype=torch.float32)


def make_betas(owner, num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas):
    """fp32 beta table of a scheduler config (zigzag_centered.py:218-233; the same four branches open every reference scheduler)."""
    if trained_betas is not None:
        return torch.tensor(trained_betas, dtype=torch.float32)
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if beta_schedule == "scaled_linear":             # the latent-diffusion schedule: linear in sqrt(beta)
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    if beta_schedule == "squaredcos_cap_v2":
        return betas_for_alpha_bar(num_train_timesteps)
    raise NotImplementedError(f"{beta_schedule} does is not implemented for {owner}")


class DDIMUncertaintyCore(SchedulerMixin, ConfigMixin):
    _compatibles = [e.name for e in KarrasDiffusionSchedulers]
    order = 1

    def _init_tables(self, num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, clip_sample,
                     set_alpha_to_one, steps_offset, prediction_type, thresholding, dynamic_thresholding_ratio,
                     clip_sample_range, sample_max_value, timestep_spacing, rescale_betas_zero_snr):
        self.betas = make_betas(self.__class__, num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas)
        if rescale_betas_zero_snr:
            raise NotImplementedError("rescale_betas_zero_snr is not on the accelerated path")
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        for k, v in dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                         beta_schedule=beta_schedule, trained_betas=trained_betas, clip_sample=clip_sample,
                         set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, prediction_type=prediction_type,
                         thresholding=thresholding, dynamic_thresholding_ratio=dynamic_thresholding_ratio,
                         clip_sample_range=clip_sample_range, sample_max_value=sample_max_value,
                         timestep_spacing=timestep_spacing, rescale_betas_zero_snr=rescale_betas_zero_snr).items():
            setattr(self, k, v)
        self.noise_source = HostNoise()

    # ------------------------------------------------------------------ reference surface
    def scale_model_input(self, sample, timestep=None):
        return sample

    def _get_variance(self, timestep, prev_timestep):
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return ((1 - alpha_prod_t_prev) / (1 - alpha_prod_t)) * (1 - alpha_prod_t / alpha_prod_t_prev)

    def set_timesteps(self, num_inference_steps, device=None):
        self._set_grid(num_inference_steps, device)
        # IndexError when the window exceeds the grid, like the reference (:383-384)
        self.timestep_after_step = self.timesteps[self.config.after_step].item()
        self.timestep_end_step = self.timesteps[self.config.after_step + self.config.num_steps_uc - 1].item()

    def _set_grid(self, num_inference_steps, device=None):
        if num_inference_steps > self.config.num_train_timesteps:
            raise ValueError(
                f"`num_inference_steps`: {num_inference_steps} cannot be larger than `self.config.train_timesteps`:"
                f" {self.config.num_train_timesteps} as the unet model trained with this scheduler can only handle"
                f" maximal {self.config.num_train_timesteps} timesteps.")
        self.num_inference_steps = num_inference_steps
        n_train = self.config.num_train_timesteps
        if self.config.timestep_spacing == "linspace":
            timesteps = np.linspace(0, n_train - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif self.config.timestep_spacing == "leading":
            step_ratio = n_train // self.num_inference_steps
            timesteps = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
            timesteps += self.config.steps_offset
        elif self.config.timestep_spacing == "trailing":
            step_ratio = n_train / self.num_inference_steps
            timesteps = np.round(np.arange(n_train, 0, -step_ratio)).astype(np.int64)
            timesteps -= 1
        else:
            raise ValueError(f"{self.config.timestep_spacing} is not supported. Please make sure to choose one of "
                             "'leading' or 'trailing'.")
        self.timesteps = torch.from_numpy(timesteps).to(device)

    def in_uc_window(self, timestep):
        return self.timestep_end_step <= timestep <= self.timestep_after_step

    def _ddim(self, model_output, timestep, sample, eta, use_clipped_model_output, noise):
        """DDIM algebra of step() :462-525 -> (eps fp32, prev_sample, pred_original_sample)."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self.config.prediction_type != "epsilon":
            raise NotImplementedError("the HIP path implements prediction_type='epsilon' (the only one the reference models use)")
        if self.config.thresholding:
            raise NotImplementedError("dynamic thresholding is not on the accelerated path")
        if use_clipped_model_output:
            raise NotImplementedError("use_clipped_model_output is not on the accelerated path")
        prev_timestep = timestep - self.config.num_train_timesteps // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        variance = self._get_variance(timestep, prev_timestep)
        std_dev_t = eta * variance ** 0.5
        eps = ops._f32c(model_output)
        prev, x0 = ops.ddim_step(eps, ops._f32c(sample), beta_prod_t ** 0.5, alpha_prod_t ** 0.5, alpha_prod_t_prev ** 0.5,
                                 (1 - alpha_prod_t_prev - std_dev_t ** 2) ** 0.5, sigma=float(std_dev_t),
                                 noise=noise if eta > 0 else None, clip=self.config.clip_sample,
                                 clip_range=self.config.clip_sample_range)
        return eps, prev, x0

    def _coefficients(self, timestep, eta=0.0):
        """(sqrt(1-a_t), sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev-sigma^2)) of the DDIM update at ``timestep`` (:462-507)."""
        prev_timestep = timestep - self.config.num_train_timesteps // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        std_dev_t = eta * self._get_variance(timestep, prev_timestep) ** 0.5
        return ((1 - alpha_prod_t) ** 0.5, alpha_prod_t ** 0.5, alpha_prod_t_prev ** 0.5,
                (1 - alpha_prod_t_prev - std_dev_t ** 2) ** 0.5)

    def _renoise_next(self, prev_sample, noise, timestep):
        """x_t_hat = sqrt(1-beta_t) x_{t-1} + sqrt(beta_t) noise  (:538)."""
        return ops.axpby(prev_sample, noise, torch.sqrt(1 - self.betas[timestep]), torch.sqrt(self.betas[timestep]))

    def add_noise(self, original_samples, noise, timesteps):
        """:575-604 for a scalar timestep (the only form the hot path uses)."""
        if isinstance(timesteps, torch.Tensor):
            if timesteps.numel() != 1:
                raise NotImplementedError("add_noise: per-sample timesteps are not on the accelerated path")
            timesteps = int(timesteps.reshape(-1)[0])
        a = self.alphas_cumprod[timesteps]
        return ops.axpby(ops._f32c(original_samples), ops._f32c(noise), a ** 0.5, (1 - a) ** 0.5)

    def get_velocity(self, sample, noise, timesteps):
        """:607-624 for a scalar timestep: sqrt(a) noise - sqrt(1-a) sample."""
        if isinstance(timesteps, torch.Tensor):
            if timesteps.numel() != 1:
                raise NotImplementedError("get_velocity: per-sample timesteps are not on the accelerated path")
            timesteps = int(timesteps.reshape(-1)[0])
        a = self.alphas_cumprod[timesteps]
        return ops.axpby(ops._f32c(noise), ops._f32c(sample), a ** 0.5, -((1 - a) ** 0.5))

    def __len__(self):
        return self.config.num_train_timesteps
