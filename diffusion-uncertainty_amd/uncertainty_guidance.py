"""Percentile-thresholded, uncertainty-guided score update (Stable-Diffusion path).

Mirror of reference ``uncertainty_guidance.py``: ``get_uncertainty_guided_score_with_percentile`` :61-131
(posterior branch :101-120), ``predict_model_stable_diffusion`` :135-147, module global ``use_posterior``
:8.  The M CFG denoiser calls stay with the caller's ``model`` (a third-party diffusers UNet);
everything this file itself computes - re-noising (dua_axpby), the unbiased variance over M(+1) samples
(dua_mc_reduce_var), the exact per-sample quantile mask (dua_quantile_mask, bit-exact with
torch.quantile's fp32 'linear' rule) and the posterior blend (dua_posterior_blend) - runs in libdua_hip.so.

Gradient mode (``use_posterior = False``, :107-108,121-129; the default of the reference script) reads
``pred_epsilon.grad`` after ``uncertainty.backward()``: d sum_px mean_b var_m(eps_hat_m) / d pred_epsilon through the M denoiser
calls.  Here that is an explicit vector-Jacobian product: the denoiser must offer ``forward_with_tape`` / ``backward_sample``
(stable_diffusion/unet_2d_condition.py; input gradients on the HIP library, csrc/backward.hip) - there is no autograd graph.
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
    if not use_posterior:
        return _gradient_update(eps, x0, t_tensor, y, model, sa, sb, percentile, M, guidance_scale, lr, reps)
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


def _gradient_update(eps, x0, t_tensor, y, model, sa, sb, percentile, M, guidance_scale, lr, reps):
    """use_posterior = False (:83-108,121-129): final = eps + lr * mask * d U / d eps with U = sum_px mean_b var_m(eps_hat_m)
    (unbiased over the M samples; the original eps is not appended in this mode, :101).  x_hat_m = sqrt(a) x0(eps) + sqrt(1-a) z_m
    depends on eps through x0 with d x_hat / d eps = -sqrt(1-a) on BOTH halves of the CFG-doubled latent, eps_hat_m is the CFG
    combine of the two halves of one denoiser call, so
        d U / d eps = -sqrt(1-a) * sum_m sum_halves J_m^T [ (1-s) G_m | s G_m ],   G_m = 2 (eps_hat_m - mean) / ((M-1) B).
    The M calls run as ONE taped forward of batch 2M and ONE backward (rows [uncond of every sample | text of every sample])."""
    if not (hasattr(model, "forward_with_tape") and hasattr(model, "backward_sample")):
        raise NotImplementedError("gradient-mode guidance (use_posterior=False) needs a denoiser with forward_with_tape / "
                                  "backward_sample (stable_diffusion.UNet2DConditionModel); a third-party model has no HIP backward")
    B = eps.shape[0]
    if not (reps == 2 and B == 1 and M >= 2):
        raise NotImplementedError("gradient-mode guidance: CFG-doubled latent with one prompt (B = 1) and M >= 2, as the reference pipeline calls it")
    x_hats = []
    for _ in range(M):
        z = noise_source.draw(eps.shape, eps.device)                                        # randn_like(pred_epsilon) :88
        x_hats.append(ops.axpby(x0, z.repeat(2, *([1] * (z.ndim - 1))), sa, sb))
    xin = torch.cat([xh[:1] for xh in x_hats] + [xh[1:] for xh in x_hats])
    half = y.shape[0] // 2
    yin = torch.cat([y[:half].repeat(M, *([1] * (y.ndim - 1))), y[half:].repeat(M, *([1] * (y.ndim - 1)))])
    t = torch.as_tensor(t_tensor)
    out, tape = model.forward_with_tape(xin, t.reshape(-1)[0] if t.numel() else t, yin)
    un, tx = ops._f32c(out[:M]), ops._f32c(out[M:])
    e_hat = ops.axpby(un, ops.axpby(tx, un, 1.0, -1.0), 1.0, guidance_scale)                # [M, 4, h, w]: the stack of :103
    u = ops.mc_reduce_var(e_hat).reshape(eps.shape)                                         # torch.var over M samples :106
    G = ops.mc_var_grad(e_hat, 2.0 / ((M - 1) * B))                                         # d U / d eps_hat_m
    g_out = torch.cat([ops.axpby(G, G, 1.0 - guidance_scale, 0.0), ops.axpby(G, G, guidance_scale, 0.0)])
    gx = model.backward_sample(tape, g_out)                                                 # d U / d x_hat, [2M, 4, h, w]
    grad_sum = ops.batch_sum(gx).reshape(eps.shape)
    grad = ops.axpby(grad_sum, grad_sum, -float(sb), 0.0)                                   # pred_epsilon.grad :122
    _mask, q = ops.quantile_mask(u, percentile)                                             # :112
    m = ops.threshold_map(u, q, per_sample=True)
    return ops.map_blend(ops.axpby(eps, grad, 1.0, float(lr)), eps, m)                      # eps + lr * grad * mask :129
