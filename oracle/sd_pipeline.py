"""PNDM (PLMS) stepping and the uncertainty-guided Stable-Diffusion loop (oracle; test infrastructure).  PARITY UNPINNED.

Loop: reference ``pipeline_uncertainty/pipeline_stable_diffusion_uncertainty_guided.py:719-776`` (its own code; the guidance
call is pinned by golden G7).  Scheduler: diffusers' ``PNDMScheduler`` with the SD-1.5 scheduler_config (un-vendored
``diffusers==0.31.0``), restated from the published algorithm: ``set_timesteps`` with skip_prk_steps, ``step_plms``,
``_get_prev_sample``.
"""
import numpy as np
import torch

from . import guidance


class PNDM:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1, set_alpha_to_one=False):
        self.T = num_train_timesteps
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.steps_offset = steps_offset

    def set_timesteps(self, n):
        self.n = n
        ts = (np.arange(0, n) * (self.T // n)).round() + self.steps_offset
        self.timesteps = np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].copy().astype(np.int64)
        self.ets, self.counter, self.cur_sample = [], 0, None

    def step(self, eps, t, sample):
        t = int(t)
        prev_t = t - self.T // self.n
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(eps)
        else:
            prev_t, t = t, t + self.T // self.n
        e = self.ets
        if len(e) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(e) == 1 and self.counter == 1:
            eps = (eps + e[-1]) / 2
            sample, self.cur_sample = self.cur_sample, None
        elif len(e) == 2:
            eps = (3 * e[-1] - e[-2]) / 2
        elif len(e) == 3:
            eps = (23 * e[-1] - 16 * e[-2] + 5 * e[-3]) / 12
        else:
            eps = (1 / 24) * (55 * e[-1] - 59 * e[-2] + 37 * e[-3] - 9 * e[-4])
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_p = 1 - a_t, 1 - a_p
        coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        self.counter += 1
        return coeff * sample - (a_p - a_t) * eps / denom


@torch.no_grad()
def sample(unet, latents, prompt_embeds, negative_prompt_embeds, n_steps, guidance_scale, start_step_uc, num_steps_uc,
           percentile, M, draw=torch.randn_like, use_posterior=True, lr=1.0):
    """``unet(x [2B,4,h,w], t [2B], enc [2B,77,768]) -> eps``; returns the final latents.  use_posterior=False: the gradient
    update (the reference script's default), ``unet`` must then be autograd-traceable."""
    sch = PNDM()
    sch.set_timesteps(n_steps)
    pe = torch.cat([negative_prompt_embeds, prompt_embeds])

    def cfg(x2, t):
        tv = torch.full((x2.shape[0],), int(t), dtype=torch.long)
        un, tx = unet(x2, tv, pe).chunk(2)
        return un + guidance_scale * (tx - un)

    for i, t in enumerate(sch.timesteps):
        x_in = torch.cat([latents] * 2)
        eps = cfg(x_in, t)
        if start_step_uc <= i < start_step_uc + num_steps_uc:
            a = float(sch.alphas_cumprod[int(t)])
            if not use_posterior:
                eps = guidance.guided_score_percentile_gradient(
                    eps, x_in, lambda xh, tt, yy: unet(xh, torch.full((xh.shape[0],), int(tt), dtype=torch.long), yy), t, pe,
                    guidance_scale, a, percentile, M, draw, lr=lr)[0]
                latents = sch.step(eps, t, latents)
                continue
            new, _, _, _ = guidance.guided_score_percentile(
                eps.numpy(), x_in.numpy(), lambda xh: cfg(torch.from_numpy(xh), t).numpy(), a, percentile, M,
                lambda like: draw(torch.from_numpy(like)).numpy())
            eps = torch.from_numpy(new)
        latents = sch.step(eps, t, latents)
    return latents
