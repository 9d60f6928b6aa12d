"""`uncertainty_zigzag_centered` scheduler on the HIP library.

Mirror of reference schedulers_uncertainty/scheduling_ddim_uncertainty_zigzag_centered.py
(``DDIMSchedulerUncertainty.__init__`` :194-275, ``step`` :400-559, subclasses :653-661): same class
names, constructor keywords (note ``prompt_embed``, not ``y`` - a ``y=`` passed through ``from_config`` is
dropped exactly as upstream), attributes and ``step`` signature/return.

Estimator (:527-549): for each of M chains, num_zigzag re-noisings x_hat = sqrt(1-beta_t) x_{t-1} +
sqrt(beta_t) z of the SAME prev_sample and a denoiser call each; ``predict_next`` is hard-coded True
upstream (:272), so the zig-zag state x_t1 is never read and only the LAST call of a chain reaches
``uncertainty = mean_m (eps_m - eps)^2`` (:549).  With ``faithful_zigzag=False`` (default) the dead
num_zigzag-1 calls per chain are skipped - their noise draws are still consumed, so every output is
bit-identical to the faithful run; set ``scheduler.faithful_zigzag = True`` to execute them.  ``step`` honours the
``predict_next`` attribute like upstream (:533-535): set to False after construction, every chain walks the x_t1 zig-zag
(``add_noise`` of the running x0 estimate, :545-546) and every forward is live - nothing is skipped then.
"""
import torch

from .. import ops
from ..configuration import register_to_config
from ._base import DDIMSchedulerUncertaintyOutput, DDIMUncertaintyCore, betas_for_alpha_bar  # noqa: F401
from .mixin import SchedulerUncertaintyClassConditionedMixin, SchedulerUncertaintyMixin
from .traits import PredictorClassConditionedTrait


class DDIMSchedulerUncertainty(DDIMUncertaintyCore):

    @register_to_config
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False, M=30, after_step=10, num_steps_uc=10, unet=None,
                 prompt_embed=None, num_zigzag=4):
        self._init_tables(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, clip_sample,
                          set_alpha_to_one, steps_offset, prediction_type, thresholding, dynamic_thresholding_ratio,
                          clip_sample_range, sample_max_value, timestep_spacing, rescale_betas_zero_snr)
        self.M = M
        self.after_step = after_step
        self.unet = unet
        self.predict_next = True          # hard-coded upstream (:272)
        self.num_steps_uc = num_steps_uc
        self.prompt_embeds = prompt_embed
        self.num_zigzag = num_zigzag
        self.faithful_zigzag = False
        # mc_group > 1: the live forwards of that many Monte-Carlo chains of a UC step run as ONE denoiser call on the stacked
        # batch (the chains are independent: same x_{t-1}, different noise).  Same arithmetic per sample; at 288 GB of HBM the
        # 5x activations fit easily and the small-map layers get enough tiles for the 256-channel kernels.  Noise is still
        # drawn in the reference's order.  Only on the predict_next branch (chains do not feed back).
        self.mc_group = 1
        self.denoiser_calls = 0           # UC forwards actually executed (for FLOP accounting; a grouped call counts per chain)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        best_noise = self.noise_source.draw(sample.shape, sample.device)            # :500, always drawn
        if eta > 0 and variance_noise is not None and generator is not None:
            raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either `generator` or"
                             " `variance_noise` stays `None`.")
        pred_epsilon, prev_sample, pred_original_sample = self._ddim(model_output, timestep, sample, eta,
                                                                     use_clipped_model_output, best_noise)
        in_window = self.in_uc_window(timestep)
        if in_window and self.predict_next and not self.faithful_zigzag and self.mc_group > 1 and isinstance(self.prompt_embeds, torch.Tensor):
            uncertainty = self._grouped_chains(prev_sample, pred_epsilon, timestep, sample)
        elif in_window:
            uncertainty = torch.empty_like(pred_epsilon)
            sb_t, sa_t = self._coefficients(timestep)[:2]
            for m in range(self.M):
                x_t1 = pred_original_sample                                          # :530
                for j in range(self.num_zigzag):
                    last = j == self.num_zigzag - 1
                    noise = self.noise_source.draw(sample.shape, sample.device)      # :532
                    if self.predict_next:
                        if not last and not self.faithful_zigzag:
                            continue                                                 # dead forward (F4): x_t1 is never read
                        x_t_hat = self._renoise_next(prev_sample, noise, timestep)   # :538
                    else:
                        x_t_hat = self.add_noise(x_t1, noise, timestep)              # :535
                    x_t_hat = self.scale_model_input(x_t_hat, timestep)
                    model_output_uc = ops._f32c(self.predict_model(x_t_hat, timestep))   # :541-544
                    self.denoiser_calls += 1
                    if not last and not self.predict_next:                           # :545-546, unclipped x0 of the forward
                        x_t1 = ops.ddim_step(model_output_uc, x_t_hat, sb_t, sa_t, 0.0, 0.0, clip=False)[1]
                ops.mc_centered_accum(model_output_uc, pred_epsilon, uncertainty, m, self.M)   # :549
        if not return_dict:
            return (prev_sample,)
        output = DDIMSchedulerUncertaintyOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample)
        if in_window:
            output.uncertainty = uncertainty
            output.pred_epsilon = pred_epsilon
        return output

    def _grouped_chains(self, prev_sample, pred_epsilon, timestep, sample):
        """The M chains of :527-549 on the predict_next branch, `mc_group` chains per denoiser call (only the last re-noising of a
        chain is live).  Draw order as upstream: num_zigzag draws per chain, chain after chain."""
        B = sample.shape[0]
        uncertainty = torch.empty_like(pred_epsilon)
        labels = self.prompt_embeds
        m = 0
        while m < self.M:
            g = min(self.mc_group, self.M - m)
            xs = torch.empty((g * B,) + tuple(sample.shape[1:]), device=sample.device, dtype=torch.float32)
            for c in range(g):
                for j in range(self.num_zigzag):
                    noise = self.noise_source.draw(sample.shape, sample.device)      # :532 (the first num_zigzag-1 draws are dead)
                sb, sn = torch.sqrt(1 - self.betas[timestep]), torch.sqrt(self.betas[timestep])
                ops.axpby(prev_sample, noise, sb, sn, out=xs[c * B:(c + 1) * B])     # :538
            self.prompt_embeds = labels.repeat(g)
            try:
                out = ops._f32c(self.predict_model(self.scale_model_input(xs, timestep), timestep))
            finally:
                self.prompt_embeds = labels
            self.denoiser_calls += g
            for c in range(g):
                ops.mc_centered_accum(out[c * B:(c + 1) * B], pred_epsilon, uncertainty, m + c, self.M)   # :549
            m += g
        return uncertainty

    def predict_model(self, x, t):
        """:561-569 (diffusers UNet2DConditionModel convention)."""
        return self.unet(x, t, encoder_hidden_states=self.prompt_embeds, cross_attention_kwargs=None, return_dict=False)[0]


class DDIMSchedulerUncertaintyImagenet(DDIMSchedulerUncertainty):

    def predict_model(self, x, t):
        return self.unet(x, t)[:, :3]


class DDIMSchedulerUncertaintyImagenetClassConditioned(PredictorClassConditionedTrait, DDIMSchedulerUncertainty):
    class_conditioned: bool = True


assert issubclass(DDIMSchedulerUncertaintyImagenetClassConditioned, object)
_ = (SchedulerUncertaintyMixin, SchedulerUncertaintyClassConditionedMixin)
