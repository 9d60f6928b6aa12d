"""DDIM update and the three uncertainty estimators (oracle; test infrastructure).

Follows reference ``scheduling_ddim_uncertainty_zigzag_centered.py`` ``step``
:400-559, ``scheduling_ddim_uncertainty_centered.py`` :522-540 and
``scheduling_ddim_mc_dropout.py`` :498-554.  Coefficients are evaluated as
0-dim fp32 tensors exactly as the reference writes them (``** 0.5`` on fp32
tensors, ``torch.sqrt(1 - betas[t])``), not in double.

``denoise(x, t:int) -> eps`` is the callable the reference reaches through
``predict_model`` (traits.py:8-18).  ``draw(like)`` is the noise source; the
default ``torch.randn_like`` consumes the global generator in reference order.
"""
import torch


def step_coefficients(tab, t, n_steps, eta=0.0):
    """zigzag_centered.py:462-468,497-498,507 -> dict of 0-dim fp32 tensors."""
    prev_t = t - tab.num_train_timesteps // n_steps
    a_t = tab.alphas_cumprod[t]
    a_prev = tab.alphas_cumprod[prev_t] if prev_t >= 0 else tab.final_alpha_cumprod
    b_t = 1 - a_t
    variance = ((1 - a_prev) / b_t) * (1 - a_t / a_prev)           # _get_variance :294-301
    sigma = eta * variance ** 0.5
    return dict(a_t=a_t, a_prev=a_prev, b_t=b_t, sigma=sigma,
                sqrt_b_t=b_t ** 0.5, sqrt_a_t=a_t ** 0.5, sqrt_a_prev=a_prev ** 0.5,
                dir_coef=(1 - a_prev - sigma ** 2) ** 0.5)


def ddim_update(tab, eps, t, x, n_steps, eta=0.0, clip_sample=True, clip_range=1.0):
    """zigzag_centered.py:473,490-493,507-510 -> (prev_sample, pred_original_sample, coefs)."""
    c = step_coefficients(tab, t, n_steps, eta)
    x0 = (x - c["sqrt_b_t"] * eps) / c["sqrt_a_t"]
    if clip_sample:
        x0 = x0.clamp(-clip_range, clip_range)
    prev = c["sqrt_a_prev"] * x0 + c["dir_coef"] * eps
    return prev, x0, c


def renoise_next(tab, prev, noise, t):
    """zigzag_centered.py:538 - single-step beta_t re-noising of x_{t-1}."""
    return torch.sqrt(1 - tab.betas[t]) * prev + torch.sqrt(tab.betas[t]) * noise


def add_noise(tab, x0, noise, t):
    """zigzag_centered.py:575-604 (DDPM add_noise)."""
    a = tab.alphas_cumprod[t]
    return a ** 0.5 * x0 + (1 - a) ** 0.5 * noise


def step_zigzag_centered(tab, denoise, eps, t, x, n_steps, window, M, num_zigzag, eta=0.0,
                         clip_sample=True, draw=torch.randn_like, predict_next=True):
    """zigzag_centered.py:462-559.  window = (timestep_after_step, timestep_end_step).

    Returns dict(prev_sample, pred_original_sample[, uncertainty, pred_epsilon]).
    predict_next is hard-coded True in the reference (:272) so every MC chain
    re-noises prev_sample; the x_t1 zig-zag state (:545-546) is dead and only
    the last of the num_zigzag forwards of each chain is used (:548).  With the attribute set to False after
    construction, step() takes its other branch (:533-535): x_hat = add_noise(x_t1, noise, t) with x_t1 starting at
    pred_original_sample and replaced by the unclipped x0 of each forward but the last (:545-546) - every forward live.
    """
    prev, x0, c = ddim_update(tab, eps, t, x, n_steps, eta, clip_sample)
    best_noise = draw(x0)                                            # :500, always drawn
    if eta > 0:
        prev = prev + c["sigma"] * best_noise                        # :523-525
    out = dict(prev_sample=prev, pred_original_sample=x0)
    hi, lo = window
    if lo <= t <= hi:
        acc = torch.zeros_like(eps)
        for _ in range(M):
            last, x_t1 = None, x0
            for j in range(num_zigzag):
                noise = draw(x0)
                x_hat = renoise_next(tab, prev, noise, t) if predict_next else add_noise(tab, x_t1, noise, t)
                last = denoise(x_hat, t)
                if j != num_zigzag - 1:
                    x_t1 = (x_hat - c["sqrt_b_t"] * last) / c["sqrt_a_t"]      # :545-546 (read only when not predict_next)
            acc += (last - eps) ** 2
        out["uncertainty"] = acc / M                                 # :549 mean_m (eps_m - eps)^2
        out["pred_epsilon"] = eps
    return out


def step_centered(tab, denoise, eps, t, x, n_steps, window, M, predict_next=False, eta=0.0,
                  clip_sample=True, draw=torch.randn_like):
    """scheduling_ddim_uncertainty_centered.py:457-556 (one forward per MC sample)."""
    prev, x0, c = ddim_update(tab, eps, t, x, n_steps, eta, clip_sample)
    best_noise = draw(x0)
    if eta > 0:
        prev = prev + c["sigma"] * best_noise
    out = dict(prev_sample=prev, pred_original_sample=x0)
    hi, lo = window
    if lo <= t <= hi:
        acc = torch.zeros_like(eps)
        for _ in range(M):
            noise = draw(x0)
            x_hat = renoise_next(tab, prev, noise, t) if predict_next else add_noise(tab, x0, noise, t)
            acc += (denoise(x_hat, t) - eps) ** 2
        out["uncertainty"] = acc / M
        out["pred_epsilon"] = eps
    return out


def step_mc_dropout(tab, denoise_train, eps, t, x, n_steps, window, M, eta=0.0, clip_sample=True,
                    draw=torch.randn_like):
    """scheduling_ddim_mc_dropout.py:460-556.  ``denoise_train`` runs the denoiser
    in train mode (dropout live) on the SAME input M times; uncertainty is the
    unbiased variance (torch.var, :506).  No best_noise draw in this file;
    noise is drawn only when eta > 0 (:536-543)."""
    hi, lo = window
    out = {}
    if lo <= t <= hi:
        scores = [denoise_train(x, t) for _ in range(M)]
        out["uncertainty"] = torch.var(torch.stack(scores, 0), dim=0)
        out["pred_epsilon"] = eps
    prev, x0, c = ddim_update(tab, eps, t, x, n_steps, eta, clip_sample)
    if eta > 0:
        prev = prev + c["sigma"] * draw(eps)
    out.update(prev_sample=prev, pred_original_sample=x0)
    return out


def step_uncertainty_image(tab, denoise, eps, t, x, n_steps, window, M, predict_next=False, eta=0.0,
                           clip_sample=True, draw=torch.randn_like):
    """scheduling_ddim_uncertainty_image.py:388-543: like ``step_centered`` but every MC prediction is pushed through an
    unclipped DDIM update (``predict_x_t1`` :545-554) and the map is the unbiased variance of those x_{t-1} (:533)."""
    prev, x0, c = ddim_update(tab, eps, t, x, n_steps, eta, clip_sample)
    best_noise = draw(x0)
    if eta > 0:
        prev = prev + c["sigma"] * best_noise
    out = dict(prev_sample=prev, pred_original_sample=x0)
    hi, lo = window
    if lo <= t <= hi:
        xs = []
        for _ in range(M):
            noise = draw(x0)
            x_hat = renoise_next(tab, prev, noise, t) if predict_next else add_noise(tab, x0, noise, t)
            e = denoise(x_hat, t)
            x0_m = (x_hat - c["sqrt_b_t"] * e) / c["sqrt_a_t"]
            xs.append(c["sqrt_a_prev"] * x0_m + c["dir_coef"] * e)
        out["uncertainty"] = torch.var(torch.stack(xs, 0), dim=0)
        out["pred_epsilon"] = eps
    return out


def step_uncertainty_hooks(tab, denoise_noisy, eps, t, x, n_steps, window, M, eta=0.0, clip_sample=True,
                           draw=torch.randn_like):
    """scheduling_ddim_uncertainty.py:393-552 (scheduler_type 'uncertainty' / 'uncertainty_original'): M forwards of the
    SAME x_t with Gaussian noise (std 0.01, :37-40) added to the outputs of four named UNet modules (:520-529);
    ``denoise_noisy(x, t)`` runs the denoiser with those hooks live.  uncertainty = mean_m (eps_m - eps)^2 (:540)."""
    prev, x0, c = ddim_update(tab, eps, t, x, n_steps, eta, clip_sample)
    best_noise = draw(x0)
    if eta > 0:
        prev = prev + c["sigma"] * best_noise
    out = dict(prev_sample=prev, pred_original_sample=x0)
    hi, lo = window
    if lo <= t <= hi:
        acc = torch.zeros_like(eps)
        for _ in range(M):
            acc += (denoise_noisy(x, t) - eps) ** 2
        out["uncertainty"] = acc / M
        out["pred_epsilon"] = eps
    return out


HOOKED_MODULES = ("input_blocks.8.0", "output_blocks.12.0", "output_blocks.1.0", "output_blocks.4.0")     # :522-528


def step_centered_d(tab, denoise, eps, t, x, n_steps, window, M, timesteps, uncertainty_distance, eta=0.0,
                    clip_sample=True, draw=torch.randn_like):
    """scheduling_ddim_uncertainty_centered_d.py:396-551.  Kept as written: ``alphas_cumprod`` is indexed by the STEP
    INDEX + distance (:532), not by a timestep, and the denoiser receives the step index ``ending_step`` as its timestep
    (:528,537-539)."""
    prev, x0, c = ddim_update(tab, eps, t, x, n_steps, eta, clip_sample)
    best_noise = draw(x0)
    if eta > 0:
        prev = prev + c["sigma"] * best_noise
    out = dict(prev_sample=prev, pred_original_sample=x0)
    hi, lo = window
    if lo <= t <= hi:
        ts = sorted([int(v) for v in timesteps], reverse=True)
        idx = ts.index(t)
        acc = torch.zeros_like(eps)
        for _ in range(M):
            noise = draw(x0)
            d = min(uncertainty_distance, len(ts) - idx - 1)
            ending_step = idx + d - 1
            end_alpha = 1 if ts[idx + d] == 0 else tab.alphas_cumprod[idx + d]
            true_alpha = c["a_t"] / end_alpha
            x_next = (x - (1 - true_alpha) ** 0.5 * eps) / true_alpha ** 0.5
            x_hat = x_next * true_alpha ** 0.5 + (1 - true_alpha) ** 0.5 * noise
            acc += (denoise(x_hat, ending_step) - eps) ** 2
        out["uncertainty"] = acc / M
        out["pred_epsilon"] = eps
    return out


def step_flip(tab, denoise, eps, t, x, n_steps, window, eta=0.0, clip_sample=True, draw=torch.randn_like):
    """scheduling_ddim_flip.py:395-530: one extra forward on the vertically flipped (clipped) x0 prediction at the same
    timestep, flipped back; uncertainty = (eps - flipped_output)^2 (:485-492).  One noise draw per step (:484)."""
    prev, x0, c = ddim_update(tab, eps, t, x, n_steps, eta, clip_sample)
    noise = draw(x0)
    if eta > 0:
        prev = prev + c["sigma"] * noise
    out = dict(prev_sample=prev, pred_original_sample=x0, score=eps)
    hi, lo = window
    if lo <= t <= hi:
        back = torch.flip(denoise(torch.flip(x0, dims=[2]), t), dims=[2])
        out["uncertainty"] = (eps - back) ** 2
        out["pred_epsilon"] = eps
    return out
