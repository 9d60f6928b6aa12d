"""`dpm_2_uncertainty_centered` scheduler on the HIP library.

Mirror of reference schedulers_uncertainty/scheduling_dpm_2_uncertainty_centered.py.  Despite its name the upstream
class is a multistep DPM-Solver++ (order 2, midpoint) with the centred uncertainty estimator bolted on:
``__init__`` :184-276 (VP tables, sigmas), ``set_timesteps`` :285-367 (note the ``N + 1`` in the 'leading' / 'linspace'
grids, sigmas interpolated at the integer timesteps, final sigma 0), ``convert_model_output`` :523-528,
first-order update :624-632, second-order update :694-727, ``step`` :876-977.

Upstream behaviour kept on purpose:
  * inside the UC window ``model_output`` already is the converted data prediction x0_pred, so
    ``pred_original_sample = (sample - sqrt(1-a_t) x0_pred) / sqrt(a_t)`` (:946), the uncertainty is
    ``mean_m (eps_m - x0_pred)^2`` (:963) and ``pred_epsilon`` carries x0_pred (:970);
  * ``predict_next=True`` raises NotImplementedError (:955);
  * ``return_dict`` is ignored: a record is always returned (``SchedulerOutput`` outside the window).
Only the configuration the factory can produce is on the accelerated path: algorithm_type 'dpmsolver++', solver_type
'midpoint', solver_order <= 2, prediction_type 'epsilon', no thresholding / Karras / Lu sigmas.

Scalars are 0-dim fp32 CPU tensors evaluated exactly as upstream writes them; tensor arithmetic runs in libdua_hip.so
(dua_x0_from_eps, dua_axpby, dua_dpmpp_2m_update, dua_mc_centered_accum).
"""
import numpy as np
import torch

from .. import ops
from ..configuration import BaseOutput, ConfigMixin, KarrasDiffusionSchedulers, SchedulerMixin, register_to_config
from ._base import betas_for_alpha_bar
from .noise import HostNoise
from .traits import PredictorClassConditionedTrait


class SchedulerOutput(BaseOutput):
    """diffusers.schedulers.scheduling_utils.SchedulerOutput: ``prev_sample`` only."""

    def __init__(self, prev_sample=None, **kw):
        super().__init__(prev_sample=prev_sample, **kw)


class SchedulerUncertaintyOutput(BaseOutput):
    """:34-52."""

    def __init__(self, prev_sample=None, pred_original_sample=None, uncertainty=None, pred_epsilon=None, **kw):
        super().__init__(prev_sample=prev_sample, pred_original_sample=pred_original_sample, uncertainty=uncertainty,
                         pred_epsilon=pred_epsilon, **kw)


class KDPM2DiscreteSchedulerUncertainty(SchedulerMixin, ConfigMixin):
    _compatibles = [e.name for e in KarrasDiffusionSchedulers]
    order = 1

    @register_to_config
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, solver_order=2, prediction_type="epsilon", thresholding=False,
                 dynamic_thresholding_ratio=0.995, sample_max_value=1.0, algorithm_type="dpmsolver++",
                 solver_type="midpoint", lower_order_final=True, euler_at_final=False, use_karras_sigmas=False,
                 use_lu_lambdas=False, final_sigmas_type="zero", lambda_min_clipped=-float("inf"), variance_type=None,
                 timestep_spacing="linspace", steps_offset=0, M=30, after_step=10, num_steps_uc=10, unet=None,
                 predict_next=False, prompt_embed=None):
        if trained_betas is not None:
            self.betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "squaredcos_cap_v2":
            self.betas = betas_for_alpha_bar(num_train_timesteps)
        else:
            raise NotImplementedError(f"{beta_schedule} does is not implemented for {self.__class__}")
        if algorithm_type != "dpmsolver++" or solver_type != "midpoint" or solver_order not in (1, 2):
            raise NotImplementedError("the HIP path implements dpmsolver++ / midpoint / order <= 2 (what the factory builds)")
        if prediction_type != "epsilon" or thresholding or use_karras_sigmas or use_lu_lambdas:
            raise NotImplementedError("the HIP path implements prediction_type='epsilon' without thresholding / Karras / Lu sigmas")
        if final_sigmas_type not in ("zero", "sigma_min"):
            raise ValueError(f"`final_sigmas_type` must be one of 'zero', or 'sigma_min', but got {final_sigmas_type}")
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.alpha_t = torch.sqrt(self.alphas_cumprod)
        self.sigma_t = torch.sqrt(1 - self.alphas_cumprod)
        self.lambda_t = torch.log(self.alpha_t) - torch.log(self.sigma_t)
        self.sigmas = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        timesteps = np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=np.float32)[::-1].copy()
        self.timesteps = torch.from_numpy(timesteps)
        self.model_outputs = [None] * solver_order
        self.lower_order_nums = 0
        self._step_index = None
        self.M, self.after_step, self.num_steps_uc = M, after_step, num_steps_uc
        self.unet, self.predict_next, self.prompt_embed = unet, predict_next, prompt_embed
        self.prompt_embeds = None
        self.noise_source = HostNoise()
        self.denoiser_calls = 0

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps=None, device=None):
        """:285-367."""
        clipped_idx = torch.searchsorted(torch.flip(self.lambda_t, [0]), self.config.lambda_min_clipped)
        last_timestep = ((self.config.num_train_timesteps - clipped_idx).numpy()).item()
        if self.config.timestep_spacing == "linspace":
            timesteps = np.linspace(0, last_timestep - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif self.config.timestep_spacing == "leading":
            step_ratio = last_timestep // (num_inference_steps + 1)
            timesteps = (np.arange(0, num_inference_steps + 1) * step_ratio).round()[::-1][:-1].copy().astype(np.int64)
            timesteps += self.config.steps_offset
        elif self.config.timestep_spacing == "trailing":
            step_ratio = self.config.num_train_timesteps / num_inference_steps
            timesteps = np.arange(last_timestep, 0, -step_ratio).round().copy().astype(np.int64)
            timesteps -= 1
        else:
            raise ValueError(f"{self.config.timestep_spacing} is not supported. Please make sure to choose one of "
                             "'linspace', 'leading' or 'trailing'.")
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sigmas = np.interp(timesteps, np.arange(0, len(sigmas)), sigmas)
        if self.config.final_sigmas_type == "sigma_min":
            sigma_last = ((1 - self.alphas_cumprod[0]) / self.alphas_cumprod[0]) ** 0.5
        else:
            sigma_last = 0
        sigmas = np.concatenate([sigmas, [sigma_last]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sigmas)
        self.timesteps = torch.from_numpy(timesteps).to(device=device, dtype=torch.int64)
        self.num_inference_steps = len(timesteps)
        self.model_outputs = [None] * self.config.solver_order
        self.lower_order_nums = 0
        self._step_index = None
        # IndexError when the window exceeds the grid, like the reference (:366-367)
        self.timestep_after_step = int(timesteps[self.config.after_step])
        self.timestep_end_step = int(timesteps[self.config.after_step + self.config.num_steps_uc - 1])

    def _sigma_to_alpha_sigma_t(self, sigma):
        """:436-440."""
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def _init_step_index(self, timestep):
        """:857-874."""
        if isinstance(timestep, torch.Tensor):
            timestep = timestep.to(self.timesteps.device)
        index_candidates = (self.timesteps == timestep).nonzero()
        if len(index_candidates) == 0:
            step_index = len(self.timesteps) - 1
        elif len(index_candidates) > 1:
            step_index = index_candidates[1].item()
        else:
            step_index = index_candidates[0].item()
        self._step_index = step_index

    def convert_model_output(self, model_output, sample=None):
        """:523-528 - data prediction of the epsilon model."""
        if self.config.variance_type in ["learned", "learned_range"]:
            model_output = model_output[:, :3]
        sigma = self.sigmas[self.step_index]
        alpha_t, sigma_t = self._sigma_to_alpha_sigma_t(sigma)
        return ops.x0_from_eps(ops._f32c(model_output), ops._f32c(sample), sigma_t, alpha_t)

    def _lambdas(self, *sigmas):
        out = []
        for s in sigmas:
            a, sg = self._sigma_to_alpha_sigma_t(s)
            out.append((a, sg, torch.log(a) - torch.log(sg)))
        return out

    def dpm_solver_first_order_update(self, model_output, sample=None, noise=None):
        """:624-632."""
        (alpha_t, sigma_t, lambda_t), (_, sigma_s, lambda_s) = self._lambdas(self.sigmas[self.step_index + 1],
                                                                            self.sigmas[self.step_index])
        h = lambda_t - lambda_s
        return ops.axpby(ops._f32c(sample), model_output, sigma_t / sigma_s, -(alpha_t * (torch.exp(-h) - 1.0)))

    def multistep_dpm_solver_second_order_update(self, model_output_list, sample=None, noise=None):
        """:694-727 (dpmsolver++, midpoint)."""
        (alpha_t, sigma_t, lambda_t), (_, sigma_s0, lambda_s0), (_, _, lambda_s1) = self._lambdas(
            self.sigmas[self.step_index + 1], self.sigmas[self.step_index], self.sigmas[self.step_index - 1])
        m0, m1 = model_output_list[-1], model_output_list[-2]
        h, h_0 = lambda_t - lambda_s0, lambda_s0 - lambda_s1
        r0 = h_0 / h
        c2 = alpha_t * (torch.exp(-h) - 1.0)
        return ops.dpmpp_2m_update(ops._f32c(sample), m0, m1, sigma_t / sigma_s0, c2, 0.5 * c2, 1.0 / r0)

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        """:876-977."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self.step_index is None:
            self._init_step_index(timestep)
        n = len(self.timesteps)
        lower_order_final = (self.step_index == n - 1) and (
            self.config.euler_at_final or (self.config.lower_order_final and n < 15) or self.config.final_sigmas_type == "zero")
        model_output = self.convert_model_output(model_output, sample=sample)
        for i in range(self.config.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = model_output
        if self.config.solver_order == 1 or self.lower_order_nums < 1 or lower_order_final:
            prev_sample = self.dpm_solver_first_order_update(model_output, sample=sample)
        else:
            prev_sample = self.multistep_dpm_solver_second_order_update(self.model_outputs, sample=sample)
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1

        is_uncertainty_timestep = self.timestep_end_step <= timestep <= self.timestep_after_step
        if is_uncertainty_timestep:
            alpha_prod_t = self.alphas_cumprod[int(timestep)]
            beta_prod_t = 1 - alpha_prod_t
            # upstream quirk (:946): model_output is the converted x0 prediction here
            pred_original_sample = ops.x0_from_eps(model_output, ops._f32c(sample), beta_prod_t ** 0.5, alpha_prod_t ** 0.5)
            uncertainty = torch.empty_like(model_output)
            for m in range(self.M):
                noise = self.noise_source.draw(sample.shape, sample.device)
                if self.predict_next:
                    raise NotImplementedError("predict_next not implemented yet")        # :955
                x_t_hat = ops.axpby(pred_original_sample, noise, torch.sqrt(alpha_prod_t), torch.sqrt(1 - alpha_prod_t))
                x_t_hat = self.scale_model_input(x_t_hat, timestep)
                model_output_uc = self.predict_model(x_t_hat, timestep)
                self.denoiser_calls += 1
                ops.mc_centered_accum(ops._f32c(model_output_uc), model_output, uncertainty, m, self.M)
        self._step_index += 1
        if is_uncertainty_timestep:
            return SchedulerUncertaintyOutput(prev_sample=prev_sample, uncertainty=uncertainty,
                                              pred_original_sample=pred_original_sample, pred_epsilon=model_output)
        return SchedulerOutput(prev_sample=prev_sample)

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    def add_noise(self, original_samples, noise, timesteps):
        """:994-1027 for a scalar timestep."""
        if isinstance(timesteps, torch.Tensor):
            if timesteps.numel() != 1:
                raise NotImplementedError("add_noise: per-sample timesteps are not on the accelerated path")
            timesteps = int(timesteps.reshape(-1)[0])
        cand = (self.timesteps == timesteps).nonzero()
        idx = len(self.timesteps) - 1 if len(cand) == 0 else (cand[1].item() if len(cand) > 1 else cand[0].item())
        alpha_t, sigma_t = self._sigma_to_alpha_sigma_t(self.sigmas[idx])
        return ops.axpby(ops._f32c(original_samples), ops._f32c(noise), alpha_t, sigma_t)

    def __len__(self):
        return self.config.num_train_timesteps


class KDPM2SchedulerUncertaintyImagenet(KDPM2DiscreteSchedulerUncertainty):

    def predict_model(self, x, t):
        return self.unet(x, t)[:, :3]


class KDPM2SchedulerUncertaintyImagenetClassConditioned(PredictorClassConditionedTrait, KDPM2SchedulerUncertaintyImagenet):
    class_conditioned: bool = True
