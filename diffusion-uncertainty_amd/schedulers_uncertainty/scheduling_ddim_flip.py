"""`flip` scheduler on the HIP library: the mirror-consistency estimator.

Mirror of reference schedulers_uncertainty/scheduling_ddim_flip.py (``DDIMSchedulerUncertainty.__init__`` :194-264 - note
``prompt_embeds=`` and no ``M``; ``step`` :395-530; subclasses :610-620).  Inside the UC window the denoiser is called ONCE
more, on the vertically flipped x0 prediction (``torch.flip(pred_original_sample, dims=[2])`` - the clipped x0, not x_t, at
the same timestep: kept as upstream writes it), its output is flipped back and
``uncertainty = (pred_epsilon - flipped_output)^2`` (:485-492).  Like upstream the record carries ``score`` on every step and
``pred_original_sample`` / ``score`` / ``uncertainty`` / ``pred_epsilon`` are moved to the CPU inside ``step`` (:523-526); one
``randn_like`` is always drawn (:484).

One upstream defect is not reproduced: the class the factory instantiates,
``DDIMSchedulerUncertaintyImagenetClassConditioned`` (:618-620), inherits the diffusers-style ``predict_model`` (:601-608:
``unet(x, t, encoder_hidden_states=...)``), which an ADM / U-ViT denoiser rejects with a TypeError; here it dispatches through
``PredictorClassConditionedTrait`` like every other class-conditioned scheduler.  The unconditional
``DDIMSchedulerUncertaintyImagenet`` (:610-615) is as upstream and is the class the golden vectors (G16) come from.
"""
import torch

from .. import ops
from ..configuration import BaseOutput, register_to_config
from ._base import DDIMUncertaintyCore
from .traits import PredictorClassConditionedTrait


class DDIMSchedulerUncertaintyOutput(BaseOutput):
    """:37-55 - prev_sample, pred_original_sample, score (+ uncertainty, pred_epsilon inside the window)."""

    def __init__(self, prev_sample=None, pred_original_sample=None, score=None, uncertainty=None, **kw):
        super().__init__(prev_sample=prev_sample, pred_original_sample=pred_original_sample, score=score, uncertainty=uncertainty, **kw)


class DDIMSchedulerUncertainty(DDIMUncertaintyCore):
    class_conditioned: bool = False

    @register_to_config
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False, prompt_embeds=None, after_step=10, num_steps_uc=10, unet=None):
        self._init_tables(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, clip_sample,
                          set_alpha_to_one, steps_offset, prediction_type, thresholding, dynamic_thresholding_ratio,
                          clip_sample_range, sample_max_value, timestep_spacing, rescale_betas_zero_snr)
        self.after_step, self.num_steps_uc = after_step, num_steps_uc
        self.unet, self.prompt_embeds = unet, prompt_embeds
        self.denoiser_calls = 0

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        noise = self.noise_source.draw(sample.shape, sample.device)                   # :484, always drawn
        if eta > 0 and variance_noise is not None and generator is not None:
            raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either `generator` or"
                             " `variance_noise` stays `None`.")
        pred_epsilon, prev_sample, pred_original_sample = self._ddim(model_output, timestep, sample, eta,
                                                                     use_clipped_model_output, noise)
        in_window = self.in_uc_window(timestep)
        if in_window:
            mirrored = self.predict_model(ops.flip_h(pred_original_sample), timestep)  # :486-490
            self.denoiser_calls += 1
            uncertainty = torch.empty_like(pred_epsilon)
            ops.mc_centered_accum(ops.flip_h(ops._f32c(mirrored)), pred_epsilon, uncertainty, 0, 1)   # (eps - flip(out))^2 :492
        if not return_dict:
            return (prev_sample,)
        output = DDIMSchedulerUncertaintyOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample.cpu(),
                                                score=pred_epsilon.cpu())
        if in_window:
            output.uncertainty = uncertainty.cpu()
            output.pred_epsilon = pred_epsilon.cpu()
        return output

    def predict_model(self, x, t):
        """:601-608 (diffusers UNet2DConditionModel convention)."""
        return self.unet(x, t, encoder_hidden_states=self.prompt_embeds, cross_attention_kwargs=None, return_dict=False)[0]


class DDIMSchedulerUncertaintyImagenet(DDIMSchedulerUncertainty):

    def predict_model(self, x, t):
        if isinstance(t, int):
            t = torch.zeros(x.shape[0], dtype=torch.int64, device=x.device).fill_(t)
        return self.unet(x, t)[:, :3]


class DDIMSchedulerUncertaintyImagenetClassConditioned(PredictorClassConditionedTrait, DDIMSchedulerUncertainty):
    class_conditioned: bool = True
