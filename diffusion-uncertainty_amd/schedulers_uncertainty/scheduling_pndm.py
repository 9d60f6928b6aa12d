"""PNDM (PLMS) scheduler of Stable Diffusion 1.5 on the HIP library.

The reference's SD pipeline (pipeline_uncertainty/pipeline_stable_diffusion_uncertainty_guided.py:719-720,772-774) steps
with whatever scheduler ``from_pretrained('runwayml/stable-diffusion-v1-5')`` provides: diffusers' ``PNDMScheduler``
(scheduler_config.json: scaled_linear 0.00085..0.012, skip_prk_steps, steps_offset 1, set_alpha_to_one False) - third-party
code from the un-vendored ``diffusers==0.31.0``.  PARITY UNPINNED: restated from the published algorithm (Liu et al.,
pseudo linear multistep; diffusers ``step_plms`` / ``_get_prev_sample``), checked against oracle/sd_pipeline.py only.
Only the skip_prk_steps / epsilon-prediction form is built; the linear combinations run in libdua_hip.so (dua_axpby).
"""
import numpy as np
import torch

from .. import ops
from ..configuration import BaseOutput, ConfigMixin, SchedulerMixin, register_to_config


class SchedulerOutput(BaseOutput):
    def __init__(self, prev_sample=None, **kw):
        super().__init__(prev_sample=prev_sample, **kw)


class PNDMScheduler(SchedulerMixin, ConfigMixin):
    order = 1

    @register_to_config
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 trained_betas=None, skip_prk_steps=True, set_alpha_to_one=False, prediction_type="epsilon",
                 timestep_spacing="leading", steps_offset=1):
        if not skip_prk_steps or prediction_type != "epsilon" or timestep_spacing != "leading":
            raise NotImplementedError("only the SD-1.5 form is built: skip_prk_steps, epsilon prediction, 'leading' spacing")
        if trained_betas is not None:
            self.betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} does is not implemented for {self.__class__}")
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.pndm_order = 4
        self.cur_model_output, self.counter, self.cur_sample, self.ets = 0, 0, None, []
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        step_ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round() + self.config.steps_offset
        # skip_prk_steps: the second timestep is visited twice (the first PLMS step is a two-stage Heun-like start)
        plms = np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].copy()
        self.timesteps = torch.from_numpy(plms.astype(np.int64)).to(device)
        self.ets, self.counter, self.cur_model_output, self.cur_sample = [], 0, 0, None

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    def step(self, model_output, timestep, sample, return_dict=True, **kw):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        timestep = int(timestep)
        eps = ops._f32c(model_output)
        sample = ops._f32c(sample)
        stride = self.config.num_train_timesteps // self.num_inference_steps
        prev_timestep = timestep - stride
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(eps)
        else:
            prev_timestep, timestep = timestep, timestep + stride
        e = self.ets
        if len(e) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(e) == 1 and self.counter == 1:
            eps = ops.axpby(eps, e[-1], 0.5, 0.5)
            sample, self.cur_sample = self.cur_sample, None
        elif len(e) == 2:
            eps = ops.axpby(e[-1], e[-2], 1.5, -0.5)
        elif len(e) == 3:
            eps = ops.axpby(ops.axpby(e[-1], e[-2], 23.0 / 12.0, -16.0 / 12.0), e[-3], 1.0, 5.0 / 12.0)
        else:
            eps = ops.axpby(ops.axpby(ops.axpby(e[-1], e[-2], 55.0 / 24.0, -59.0 / 24.0), e[-3], 1.0, 37.0 / 24.0), e[-4], 1.0, -9.0 / 24.0)
        prev_sample = self._get_prev_sample(sample, timestep, prev_timestep, eps)
        self.counter += 1
        if not return_dict:
            return (prev_sample,)
        return SchedulerOutput(prev_sample=prev_sample)

    def _get_prev_sample(self, sample, timestep, prev_timestep, model_output):
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        beta_prod_t, beta_prod_t_prev = 1 - alpha_prod_t, 1 - alpha_prod_t_prev
        sample_coeff = (alpha_prod_t_prev / alpha_prod_t) ** 0.5
        denom = alpha_prod_t * beta_prod_t_prev ** 0.5 + (alpha_prod_t * beta_prod_t * alpha_prod_t_prev) ** 0.5
        return ops.axpby(sample, model_output, sample_coeff, -(alpha_prod_t_prev - alpha_prod_t) / denom)

    def __len__(self):
        return self.config.num_train_timesteps
