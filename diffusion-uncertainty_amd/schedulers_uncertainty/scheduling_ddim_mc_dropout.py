"""`mc_dropout` scheduler on the HIP library.

Mirror of reference schedulers_uncertainty/scheduling_ddim_mc_dropout.py (``__init__`` :197-286, MC loop
:498-519, DDIM part :460-546, outputs :551-554): inside the window the denoiser is switched to train mode
(dropout live), called M times on the SAME ``sample``, and the uncertainty is the unbiased variance of the
stack (``torch.var``, :506; dua_mc_reduce_var); the constructor can overwrite every ``nn.Dropout.p``
(:280-285).  This file draws no ``best_noise``; noise is drawn only when eta > 0 (:536-543).  Like the
reference, ``pred_original_sample`` / ``score`` / ``pred_epsilon`` are returned as CPU tensors (:551-554).
"""
import torch

from .. import ops
from ..configuration import register_to_config
from ._base import DDIMSchedulerUncertaintyOutput, DDIMUncertaintyCore
from .traits import PredictorClassConditionedTrait


class DDIMSchedulerUncertainty(DDIMUncertaintyCore):
    class_conditioned: bool = False

    @register_to_config
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False, y=None, uncertainty_scale=0.9, M=30, after_step=10, num_steps_uc=10,
                 unet=None, dropout=None):
        self._init_tables(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, clip_sample,
                          set_alpha_to_one, steps_offset, prediction_type, thresholding, dynamic_thresholding_ratio,
                          clip_sample_range, sample_max_value, timestep_spacing, rescale_betas_zero_snr)
        self.M = M
        self.uncertainty_scale = uncertainty_scale
        self.after_step = after_step
        self.num_steps_uc = num_steps_uc
        self.unet = unet
        self.prompt_embeds = y
        self.first_step = True
        self.denoiser_calls = 0
        if dropout is not None:
            for module in self.unet.modules():
                if isinstance(module, (torch.nn.Dropout, torch.nn.Dropout1d, torch.nn.Dropout2d, torch.nn.Dropout3d)):
                    module.p = dropout

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        in_window = self.num_inference_steps is not None and self.in_uc_window(timestep)
        if in_window:
            self.unet.train()
            scores = []
            for _ in range(self.M):
                scores.append(ops._f32c(self.predict_model(sample, timestep)))
                self.denoiser_calls += 1
            uncertainty = ops.mc_reduce_var(torch.stack(scores, dim=0))
            if self.first_step:
                self.first_step = False
                if not any(isinstance(m, torch.nn.Dropout) for m in self.unet.modules()):
                    raise ValueError("The model should have a dropout layer for MC Dropout")
                for m in self.unet.modules():
                    if isinstance(m, torch.nn.Dropout):
                        assert m.training is True
                        assert m.p > 0.0, f"Expected dropout rate > 0, got {m.p}"
            self.unet.eval()
        noise = None
        if eta > 0:
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either `generator` or"
                                 " `variance_noise` stays `None`.")
            noise = variance_noise if variance_noise is not None else self.noise_source.draw(sample.shape, sample.device)
        pred_epsilon, prev_sample, pred_original_sample = self._ddim(model_output, timestep, sample, eta,
                                                                     use_clipped_model_output, noise)
        if not return_dict:
            return (prev_sample,)
        output = DDIMSchedulerUncertaintyOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample.cpu(),
                                                score=pred_epsilon.cpu())
        if in_window:
            output.uncertainty = uncertainty
            output.pred_epsilon = pred_epsilon.cpu()
        return output

    def predict_model(self, x, t):
        return self.unet(x, t, encoder_hidden_states=self.prompt_embeds, cross_attention_kwargs=None, return_dict=False)[0]


class DDIMSchedulerUncertaintyImagenet(DDIMSchedulerUncertainty):

    def predict_model(self, x, t):
        if isinstance(t, int):
            t = torch.zeros(x.shape[0], dtype=torch.int64, device=x.device).fill_(t)
        return self.unet(x, t)[:, :3]


class DDIMSchedulerUncertaintyImagenetClassConditioned(PredictorClassConditionedTrait, DDIMSchedulerUncertainty):
    class_conditioned: bool = True
