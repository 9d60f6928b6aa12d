"""`uncertainty_image` scheduler on the HIP library.

Mirror of reference schedulers_uncertainty/scheduling_ddim_uncertainty_image.py (``__init__`` :193-277, UC loop
:515-533, ``predict_x_t1`` :545-554): per MC sample one re-noised denoiser call, then an UNCLIPPED DDIM update of that
sample; uncertainty = unbiased variance over the M predicted x_{t-1} (``torch.var`` :533, dua_mc_reduce_var).
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
            co = self._coefficients(timestep, eta)
            scores = []
            for m in range(self.M):
                noise = self.noise_source.draw(sample.shape, sample.device)
                if not self.predict_next:
                    x_t_hat = self.add_noise(pred_original_sample, noise, timestep)
                else:
                    x_t_hat = self._renoise_next(prev_sample, noise, timestep)
                x_t_hat = self.scale_model_input(x_t_hat, timestep)
                model_output_uc = ops._f32c(self.predict_model(x_t_hat, timestep))
                self.denoiser_calls += 1
                xt1, _ = ops.ddim_step(model_output_uc, x_t_hat, *co, clip=False)      # predict_x_t1 :545-554
                scores.append(xt1)
            uncertainty = ops.mc_reduce_var(torch.stack(scores, dim=0))
        if not return_dict:
            return (prev_sample,)
        output = DDIMSchedulerUncertaintyOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample)
        if in_window:
            output.uncertainty = uncertainty
            output.pred_epsilon = pred_epsilon
        return output


class DDIMSchedulerUncertaintyImagenetClassConditioned(DDIMSchedulerUncertainty):
    class_conditioned: bool = True
