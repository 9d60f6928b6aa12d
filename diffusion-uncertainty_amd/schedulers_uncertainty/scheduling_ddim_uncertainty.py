"""`uncertainty` / `uncertainty_original` scheduler (activation-noise estimator) on the HIP library.

Mirror of reference schedulers_uncertainty/scheduling_ddim_uncertainty.py (``add_gaussian_noise`` :36-40, ``__init__``
:198-282, UC loop :519-542): inside the window the SAME x_t goes through the denoiser M times while forward hooks add
N(0, 0.01^2) noise to the outputs of ``input_blocks.8.0``, ``output_blocks.12.0``, ``output_blocks.1.0`` and
``output_blocks.4.0``; uncertainty = mean_m (eps_m - eps)^2 (:540).  The HIP UNet takes the hooks as
``model.activation_noise = {module name: std}`` (noise drawn in the reference's NCHW order at the reference's points of
the execution order); modules a model does not have are skipped, as ``named_modules`` matching does upstream.
"""
import torch

from .. import ops
from ..configuration import register_to_config
from ._base import DDIMSchedulerUncertaintyOutput, DDIMUncertaintyCore
from .traits import PredictorClassConditionedTrait

HOOKED_MODULES = ("input_blocks.8.0", "output_blocks.12.0", "output_blocks.1.0", "output_blocks.4.0")     # :522-528
NOISE_STD = 0.01                                                                                          # :39


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
            if not hasattr(self.unet, "activation_noise"):
                raise NotImplementedError("scheduler_type 'uncertainty' hooks ADM module outputs; the denoiser must be the "
                                          "HIP UNetModel (activation_noise)")
            uncertainty = torch.empty_like(pred_epsilon)
            saved = self.unet.activation_noise, self.unet.activation_noise_source
            self.unet.activation_noise = {name: NOISE_STD for name in HOOKED_MODULES}
            self.unet.activation_noise_source = self.noise_source
            try:
                x = ops._f32c(sample)
                for m in range(self.M):
                    model_output_uc = self.predict_model(x, timestep)
                    self.denoiser_calls += 1
                    ops.mc_centered_accum(ops._f32c(model_output_uc), pred_epsilon, uncertainty, m, self.M)
            finally:
                self.unet.activation_noise, self.unet.activation_noise_source = saved       # hook.remove() :541-542
        if not return_dict:
            return (prev_sample,)
        output = DDIMSchedulerUncertaintyOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample)
        if in_window:
            output.uncertainty = uncertainty
            output.pred_epsilon = pred_epsilon
        return output


class DDIMSchedulerUncertaintyImagenetClassConditioned(DDIMSchedulerUncertainty):
    class_conditioned: bool = True
