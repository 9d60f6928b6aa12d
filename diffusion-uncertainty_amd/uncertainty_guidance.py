"""Percentile-thresholded, uncertainty-guided score update (Stable-Diffusion path).

Mirror of reference ``uncertainty_guidance.py``: ``get_uncertainty_guided_score_with_percentile`` :61-131
(posterior branch :101-120), ``predict_model_stable_diffusion`` :135-147, module global ``use_posterior``
:8.  The M CFG denoiser calls stay with the caller's ``model`` (a third-party diffusers UNet);
everything this file itself computes - re-noising (dua_axpby), the unbiased variance over M(+1) samples
(dua_mc_reduce_var), the exact per-sample quantile mask (dua_quantile_mask, bit-exact with
torch.quantile's fp32 'linear' rule) and the posterior blend (dua_posterior_blend) - runs in libdua_hip.so.
Gradient mode (use_posterior=False) needs a backward pass through the UNet and is out of scope.
"""
from typing import Optional

import torch

from . import ops
from .schedulers_uncertainty.noise import HostNoise

use_posterior = True
noise_source = HostNoise()
# The M Monte-Carlo denoiser calls of one guided step are independent (same timestep, same prompt): run them as ONE call of
# batch 2*M instead of M calls of batch 2 (the reference loops, uncertainty_guidance.py:87-99; same noise draws in the same
# order, same per-sample results, ~3x the GPU efficiency at the CFG batch of 2).  False: one call per sample, as upstream.
batch_mc_samples = True


def predict_model_stable_diffusion(model, sample, t_tensor, y, guidance_scale: float, extra_diffusion_kwargs=None):
    if extra_diffusion_kwargs is None:
        extra_diffusion_kwargs = dict()
    pred_noise = model(sample=sample, timestep=t_tensor, encoder_hidden_states=y, **extra_diffusion_kwargs)[0]
    noise_pred_uncond, noise_pred_text = pred_noise.chunk(2)
    # CFG combine uncond + s*(text - uncond), same two roundings as the reference, on the HIP library
    un, tx = ops._f32c(noise_pred_uncond), ops._f32c(noise_pred_text)
    return ops.axpby(un, ops.axpby(tx, un, 1.0, -1.0), 1.0, guidance_scale)


def get_uncertainty_guided_score_with_percentile(pred_epsilon, input, t_tensor, y, model, alpha_hat_t, percentile: float,
                                                 model_type, num_uncertainty_samples: int = 5,
                                                 guidance_scale: Optional[float] = None, lr: float = 1.0,
                                                 extra_diffusion_kwargs: dict | None = None):
    if not use_posterior:
        raise NotImplementedError("gradient-mode guidance (use_posterior=False) needs UNet backward; only the posterior "
                                  "update is on the accelerated path")
    if model_type != "stable-diffusion":
        raise NotImplementedError(f"model_type={model_type!r}: upstream's 'unet' branch raises UnboundLocalError "
                                  "(:90) and 'flux' has the wrong arity (:96 vs :167); only 'stable-diffusion' is supported")
    assert guidance_scale is not None
    M = num_uncertainty_samples
    a = torch.as_tensor(alpha_hat_t, dtype=torch.float32).cpu()
    sa, sb = torch.sqrt(a), torch.sqrt(1 - a)
    eps = ops._f32c(pred_epsilon)
    B = eps.shape[0]
    assert input.shape[0] in (B, 2 * B) and (input.shape[0] == B or B == 1), \
        "the reference broadcasts eps [B] against the CFG-doubled latent [2B] (:86): B must be 1"
    reps = input.shape[0] // B
    eps_rep = eps if reps == 1 else eps.repeat(reps, *([1] * (eps.ndim - 1)))
    # pred_x_0 = (input - sqrt(1-a) eps) / sqrt(a)   (:86)
    _, x0 = ops.ddim_step(eps_rep, ops._f32c(input), sb, sa, 0.0, 0.0, clip=False)
    samples = []
    x_hats = []
    for _ in range(M):
        z = noise_source.draw(eps.shape, eps.device)                                        # randn_like(pred_epsilon) :88
        z_rep = z if reps == 1 else z.repeat(reps, *([1] * (z.ndim - 1)))
        x_hat = ops.axpby(x0, z_rep, sa, sb)
        if batch_mc_samples and reps == 2 and M > 1:
            x_hats.append(x_hat)
        else:
            samples.append(predict_model_stable_diffusion(model, x_hat, t_tensor, y, guidance_scale, extra_diffusion_kwargs))
    if x_hats:
        # rows [uncond of every sample | text of every sample], prompt embeddings repeated to match: chunk(2) of the output is
        # again (uncond, text), now with M samples in each half
        B1 = eps.shape[0]
        xin = torch.cat([xh[:B1] for xh in x_hats] + [xh[B1:] for xh in x_hats])
        yin = torch.cat([y[:y.shape[0] // 2].repeat(M, *([1] * (y.ndim - 1))), y[y.shape[0] // 2:].repeat(M, *([1] * (y.ndim - 1)))])
        allm = predict_model_stable_diffusion(model, xin, t_tensor, yin, guidance_scale, extra_diffusion_kwargs)      # [M*B1, ...]
        samples.extend(allm[m * B1:(m + 1) * B1] for m in range(M))
    samples.append(eps)                                                                     # :101-102
    u = ops.mc_reduce_var(torch.stack(samples, dim=0))                                      # torch.var :106
    mask, _q = ops.quantile_mask(u, percentile)                                             # :112
    return ops.posterior_blend(eps, u, mask, M, float(a))                                   # :115-120
