"""`uncertainty_centered_d` scheduler on the HIP library.

Mirror of reference schedulers_uncertainty/scheduling_ddim_uncertainty_centered_d.py (``__init__`` :193-277,
``set_timesteps`` index tables :390-394, UC loop :524-544): x_t is pulled back over ``uncertainty_distance`` grid steps
with the cumulative alpha ratio and re-noised, one denoiser call per MC sample, uncertainty = mean_m (eps_m - eps)^2.
Kept as written upstream: ``alphas_cumprod`` is indexed with the STEP INDEX + distance (:532) and the denoiser is called
with the step index ``ending_step`` as its timestep (:528,537-539).
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
                 predict_next=False, prompt_embed=None, uncertainty_distance=20):
        self._init_tables(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, clip_sample,
                          set_alpha_to_one, steps_offset, prediction_type, thresholding, dynamic_thresholding_ratio,
                          clip_sample_range, sample_max_value, timestep_spacing, rescale_betas_zero_snr)
        self.M = M
        self.after_step = after_step
        self.unet = unet
        self.predict_next = predict_next
        self.num_steps_uc = num_steps_uc
        self.prompt_embeds = prompt_embed
        self.uncertainty_distance = uncertainty_distance
        self.denoiser_calls = 0

    def set_timesteps(self, num_inference_steps, device=None):
        super().set_timesteps(num_inference_steps, device)
        timesteps_sorted = sorted(self.timesteps.tolist(), reverse=True)                     # :390-394
        self.timestep_index = {timestep: i for i, timestep in enumerate(timesteps_sorted)}
        self.index_timestep = {i: timestep for timestep, i in self.timestep_index.items()}

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        best_noise = self.noise_source.draw(sample.shape, sample.device)
        pred_epsilon, prev_sample, pred_original_sample = self._ddim(model_output, timestep, sample, eta,
                                                                     use_clipped_model_output, best_noise)
        in_window = self.in_uc_window(timestep)
        if in_window:
            uncertainty = torch.empty_like(pred_epsilon)
            x = ops._f32c(sample)
            alpha_prod_t = self.alphas_cumprod[timestep]
            for m in range(self.M):
                noise = self.noise_source.draw(sample.shape, sample.device)
                idx = self.timestep_index[timestep]
                effective_distance = min(self.uncertainty_distance, len(self.timesteps) - idx - 1)   # stay inside the grid
                ending_step = idx + effective_distance - 1
                if self.index_timestep[idx + effective_distance] == 0:
                    end_alpha_prod = 1
                else:
                    end_alpha_prod = self.alphas_cumprod[idx + effective_distance]           # step index, as upstream (:532)
                true_alpha = alpha_prod_t / end_alpha_prod
                sb, sa = (1 - true_alpha) ** 0.5, true_alpha ** 0.5
                _, x_t_next = ops.ddim_step(pred_epsilon, x, sb, sa, 0.0, 0.0, clip=False)   # (x - sb eps) / sa  :534
                sample_hat = ops.axpby(x_t_next, noise, sa, sb)                              # :535
                model_output_uc = self.predict_model(sample_hat, ending_step)
                self.denoiser_calls += 1
                ops.mc_centered_accum(ops._f32c(model_output_uc), pred_epsilon, uncertainty, m, self.M)
        if not return_dict:
            return (prev_sample,)
        output = DDIMSchedulerUncertaintyOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample)
        if in_window:
            output.uncertainty = uncertainty
            output.pred_epsilon = pred_epsilon
        return output


class DDIMSchedulerUncertaintyImagenetClassConditioned(DDIMSchedulerUncertainty):
    class_conditioned: bool = True
