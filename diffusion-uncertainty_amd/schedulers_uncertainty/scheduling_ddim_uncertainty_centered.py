"""`uncertainty_centered` scheduler on the HIP library.

Mirror of reference schedulers_uncertainty/scheduling_ddim_uncertainty_centered.py (``__init__`` :198-282,
UC loop :522-540): one denoiser call per MC sample; ``predict_next`` selects re-noising of x_{t-1}
(:531) or ``add_noise(pred_x0, noise, t)`` (:526-528); uncertainty = mean_m (eps_m - eps)^2 (:540).
"""
import torch

from .. import ops
from ..configuration import register_to_config
from ._base import DDIMSchedulerUncertaintyOutput, DDIMUncertaintyCore
from .traits import PredictorClassConditionedTrait


class DDIMSchedulerUncertainty(DDIMUncertaintyCore, PredictorClassConditionedTrait):

    @register_to_config
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False, M=30, after_step=10, num_steps_uc=10, unet=None,
                 predict_next=False, y=None):
        self._init_tables(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, clip_sample,
                          set_alpha_to_one, steps_offset, prediction_type, thresholding, dynamic_thresholding_ratio,
                          clip_sample_range, sample_max_value, timestep_spacing, rescale_betas_zero_snr)
        self.M = M
        self.after_step = after_step
        self.unet = unet
        self.predict_next = predict_next
        self.num_steps_uc = num_steps_uc
        self.prompt_embeds = y
        self.denoiser_calls = 0

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        best_noise = self.noise_source.draw(sample.shape, sample.device)
        pred_epsilon, prev_sample, pred_original_sample = self._ddim(model_output, timestep, sample, eta,
                                                                     use_clipped_model_output, best_noise)
        in_window = self.in_uc_window(timestep)
        if in_window:
            uncertainty = torch.empty_like(pred_epsilon)
            for m in range(self.M):
                noise = self.noise_source.draw(sample.shape, sample.device)
                if not self.predict_next:
                    x_t_hat = self.add_noise(pred_original_sample, noise, timestep)
                else:
                    x_t_hat = self._renoise_next(prev_sample, noise, timestep)
                x_t_hat = self.scale_model_input(x_t_hat, timestep)
                model_output_uc = self.predict_model(x_t_hat, timestep)
                self.denoiser_calls += 1
                ops.mc_centered_accum(ops._f32c(model_output_uc), pred_epsilon, uncertainty, m, self.M)
        if not return_dict:
            return (prev_sample,)
        output = DDIMSchedulerUncertaintyOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample)
        if in_window:
            output.uncertainty = uncertainty
            output.pred_epsilon = pred_epsilon
        return output


class DDIMSchedulerUncertaintyImagenet(DDIMSchedulerUncertainty):

    def predict_model(self, x, t):
        return self.unet(x, t)[:, :3]


class DDIMSchedulerUncertaintyImagenetClassConditioned(DDIMSchedulerUncertainty):
    class_conditioned: bool = True
