"""Plain DDIM scheduler on the HIP library (no uncertainty window).

The reference's guided-sampling script builds ``diffusers.DDIMScheduler.from_config(scheduler.config)``
(scripts/generate_images_with_uncertainty_threshold.py:200-201; diffusers==0.31.0 is un-vendored, pyproject.toml:76) and hands
it to the pipelines of ``pipeline_uncertainty/``.  This class provides that object: same tables / ``set_timesteps`` /
``step`` algebra as the uncertainty schedulers (the reference's own files restate diffusers' DDIM step line by line,
zigzag_centered.py:462-525), with two differences that follow diffusers: no noise is drawn when ``eta == 0`` and the
output carries only ``prev_sample`` / ``pred_original_sample``.  ``config.after_step`` / ``config.num_steps_uc`` may be
assigned by the pipelines (posterior pipeline :117-118); they have no effect here.
"""
from ..configuration import register_to_config
from ._base import DDIMSchedulerUncertaintyOutput, DDIMUncertaintyCore


class DDIMScheduler(DDIMUncertaintyCore):

    @register_to_config
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False):
        self._init_tables(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, clip_sample,
                          set_alpha_to_one, steps_offset, prediction_type, thresholding, dynamic_thresholding_ratio,
                          clip_sample_range, sample_max_value, timestep_spacing, rescale_betas_zero_snr)
        self.prompt_embeds = None

    def set_timesteps(self, num_inference_steps, device=None):
        self._set_grid(num_inference_steps, device)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        noise = None
        if eta > 0:
            noise = variance_noise if variance_noise is not None else self.noise_source.draw(sample.shape, sample.device)
        _, prev_sample, pred_original_sample = self._ddim(model_output, timestep, sample, eta, use_clipped_model_output, noise)
        if not return_dict:
            return (prev_sample,)
        return DDIMSchedulerUncertaintyOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample)
