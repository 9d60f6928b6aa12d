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
from ._base import make_betas
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


def _half_logsnr(sigma):
    """(alpha, sigma_vp, lambda) of a k-diffusion sigma: alpha = 1/sqrt(sigma^2+1), sigma_vp = sigma*alpha, lambda = log alpha -
    log sigma_vp (:436-440 and the lambda lines of :609-612 / :700-706), on 0-dim fp32 tensors as upstream."""
    alpha = 1 / ((sigma ** 2 + 1) ** 0.5)
    sig = sigma * alpha
    return alpha, sig, torch.log(alpha) - torch.log(sig)


def _solver_grid(spacing, n, last_t, n_train, offset):
    """Integer timestep grid of set_timesteps (:296-321).  Note the N + 1 points of 'linspace' / 'leading' with the last dropped."""
    if spacing == "linspace":
        return np.linspace(0, last_t - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
    if spacing == "leading":
        return (np.arange(0, n + 1) * (last_t // (n + 1))).round()[::-1][:-1].copy().astype(np.int64) + offset
    if spacing == "trailing":
        return np.arange(last_t, 0, -(n_train / n)).round().copy().astype(np.int64) - 1
    raise ValueError(f"{spacing} is not supported. Please make sure to choose one of 'linspace', 'leading' or 'trailing'.")


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
        if algorithm_type != "dpmsolver++" or solver_type != "midpoint" or solver_order not in (1, 2):
            raise NotImplementedError("the HIP path implements dpmsolver++ / midpoint / order <= 2 (what the factory builds)")
        if prediction_type != "epsilon" or thresholding or use_karras_sigmas or use_lu_lambdas:
            raise NotImplementedError("the HIP path implements prediction_type='epsilon' without thresholding / Karras / Lu sigmas")
        if final_sigmas_type not in ("zero", "sigma_min"):
            raise ValueError(f"`final_sigmas_type` must be one of 'zero', or 'sigma_min', but got {final_sigmas_type}")
        # VP tables (:184-225)
        self.betas = make_betas(self.__class__, num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.alpha_t, self.sigma_t = torch.sqrt(self.alphas_cumprod), torch.sqrt(1 - self.alphas_cumprod)
        self.lambda_t = torch.log(self.alpha_t) - torch.log(self.sigma_t)
        self.sigmas = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.init_noise_sigma = 1.0
        self.timesteps = torch.from_numpy(np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=np.float32)[::-1].copy())
        self.num_inference_steps = None
        self._reset_solver(solver_order)
        # estimator state
        self.M, self.after_step, self.num_steps_uc = M, after_step, num_steps_uc
        self.unet, self.predict_next, self.prompt_embed = unet, predict_next, prompt_embed
        self.prompt_embeds = None
        self.noise_source = HostNoise()
        self.denoiser_calls = 0

    def _reset_solver(self, order):
        self.model_outputs = [None] * order          # x0 predictions of the last `order` steps, newest last
        self.lower_order_nums = 0                    # steps taken since the reset, capped at the order
        self._step_index = None

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps=None, device=None):
        """:285-367."""
        cfg = self.config
        clipped = torch.searchsorted(torch.flip(self.lambda_t, [0]), cfg.lambda_min_clipped)
        last_t = int((cfg.num_train_timesteps - clipped).item())
        grid = _solver_grid(cfg.timestep_spacing, num_inference_steps, last_t, cfg.num_train_timesteps, cfg.steps_offset)
        table = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        tail = float(table[0]) if cfg.final_sigmas_type == "sigma_min" else 0.0
        sig = np.concatenate([np.interp(grid, np.arange(len(table)), table), [tail]]).astype(np.float32)      # :333-352
        self.sigmas = torch.from_numpy(sig)
        self.timesteps = torch.from_numpy(grid).to(device=device, dtype=torch.int64)
        self.num_inference_steps = len(grid)
        self._reset_solver(cfg.solver_order)
        # IndexError when the window exceeds the grid, like the reference (:366-367)
        self.timestep_after_step = int(grid[cfg.after_step])
        self.timestep_end_step = int(grid[cfg.after_step + cfg.num_steps_uc - 1])

    def _sigma_to_alpha_sigma_t(self, sigma):
        a, s, _ = _half_logsnr(sigma)
        return a, s

    def _locate(self, timestep):
        """Index of ``timestep`` in the grid (:857-874): the second hit if it occurs twice, the last index if absent."""
        hits = (self.timesteps == (timestep.to(self.timesteps.device) if isinstance(timestep, torch.Tensor) else timestep)).nonzero()
        if len(hits) == 0:
            return len(self.timesteps) - 1
        return hits[min(1, len(hits) - 1)].item()

    def _init_step_index(self, timestep):
        self._step_index = self._locate(timestep)

    def convert_model_output(self, model_output, sample=None):
        """:523-528 - data prediction of the epsilon model."""
        if self.config.variance_type in ["learned", "learned_range"]:
            model_output = model_output[:, :3]
        alpha_t, sigma_t, _ = _half_logsnr(self.sigmas[self.step_index])
        return ops.x0_from_eps(ops._f32c(model_output), ops._f32c(sample), sigma_t, alpha_t)

    def _update_coefficients(self, second_order):
        """Scalars of the data-prediction DPM-Solver++ update from step_index to step_index + 1:
        x_next = c1 * x - c2 * m0 [- c3 * (m0 - m1) / r0], c1 = sigma_t/sigma_s, c2 = alpha_t (e^{-h} - 1), c3 = c2 / 2 (:624-632, :694-727)."""
        i = self.step_index
        alpha_t, sigma_t, lam_t = _half_logsnr(self.sigmas[i + 1])
        _, sigma_s, lam_s = _half_logsnr(self.sigmas[i])
        h = lam_t - lam_s
        c1, c2 = sigma_t / sigma_s, alpha_t * (torch.exp(-h) - 1.0)
        if not second_order:
            return c1, c2, None, None
        lam_p = _half_logsnr(self.sigmas[i - 1])[2]
        return c1, c2, 0.5 * c2, 1.0 / ((lam_s - lam_p) / h)

    def dpm_solver_first_order_update(self, model_output, sample=None, noise=None):
        c1, c2, _, _ = self._update_coefficients(False)
        return ops.axpby(ops._f32c(sample), model_output, c1, -c2)

    def multistep_dpm_solver_second_order_update(self, model_output_list, sample=None, noise=None):
        c1, c2, c3, rinv = self._update_coefficients(True)
        return ops.dpmpp_2m_update(ops._f32c(sample), model_output_list[-1], model_output_list[-2], c1, c2, c3, rinv)

    def _advance(self, x0_pred, sample):
        """Push the new data prediction into the history and take the solver step (:924-941)."""
        n, cfg = len(self.timesteps), self.config
        last = self.step_index == n - 1
        first_order_last = last and (cfg.euler_at_final or (cfg.lower_order_final and n < 15) or cfg.final_sigmas_type == "zero")
        self.model_outputs = self.model_outputs[1:] + [x0_pred]
        if cfg.solver_order == 1 or self.lower_order_nums < 1 or first_order_last:
            prev = self.dpm_solver_first_order_update(x0_pred, sample=sample)
        else:
            prev = self.multistep_dpm_solver_second_order_update(self.model_outputs, sample=sample)
        self.lower_order_nums = min(self.lower_order_nums + 1, cfg.solver_order)
        return prev

    def _centred_uncertainty(self, x0_pred, timestep, sample):
        """The estimator of :943-968 with its upstream quirk: the 'epsilon' it centres on is the converted x0 prediction."""
        a = self.alphas_cumprod[int(timestep)]
        pred_original_sample = ops.x0_from_eps(x0_pred, ops._f32c(sample), (1 - a) ** 0.5, a ** 0.5)           # :946
        uncertainty = torch.empty_like(x0_pred)
        for m in range(self.M):
            noise = self.noise_source.draw(sample.shape, sample.device)
            if self.predict_next:
                raise NotImplementedError("predict_next not implemented yet")                                  # :955
            x_t_hat = self.scale_model_input(ops.axpby(pred_original_sample, noise, torch.sqrt(a), torch.sqrt(1 - a)), timestep)
            eps_m = self.predict_model(x_t_hat, timestep)
            self.denoiser_calls += 1
            ops.mc_centered_accum(ops._f32c(eps_m), x0_pred, uncertainty, m, self.M)
        return pred_original_sample, uncertainty

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        """:876-977 (``return_dict`` is ignored upstream too)."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self.step_index is None:
            self._init_step_index(timestep)
        x0_pred = self.convert_model_output(model_output, sample=sample)
        prev_sample = self._advance(x0_pred, sample)
        in_window = self.timestep_end_step <= timestep <= self.timestep_after_step
        if in_window:
            pred_original_sample, uncertainty = self._centred_uncertainty(x0_pred, timestep, sample)
        self._step_index += 1
        if in_window:
            return SchedulerUncertaintyOutput(prev_sample=prev_sample, uncertainty=uncertainty,
                                              pred_original_sample=pred_original_sample, pred_epsilon=x0_pred)
        return SchedulerOutput(prev_sample=prev_sample)

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    def add_noise(self, original_samples, noise, timesteps):
        """:994-1027 for a scalar timestep."""
        if isinstance(timesteps, torch.Tensor):
            if timesteps.numel() != 1:
                raise NotImplementedError("add_noise: per-sample timesteps are not on the accelerated path")
            timesteps = int(timesteps.reshape(-1)[0])
        alpha_t, sigma_t, _ = _half_logsnr(self.sigmas[self._locate(timesteps)])
        return ops.axpby(ops._f32c(original_samples), ops._f32c(noise), alpha_t, sigma_t)

    def __len__(self):
        return self.config.num_train_timesteps


class KDPM2SchedulerUncertaintyImagenet(KDPM2DiscreteSchedulerUncertainty):

    def predict_model(self, x, t):
        return self.unet(x, t)[:, :3]


class KDPM2SchedulerUncertaintyImagenetClassConditioned(PredictorClassConditionedTrait, KDPM2SchedulerUncertaintyImagenet):
    class_conditioned: bool = True
