"""Class-conditional posterior-guided sampling (oracle; test infrastructure - never imported by the product).

CPU restatement (torch fp32) of reference
``pipeline_uncertainty/pipeline_sampler_class_conditional_uncertainty_guided_posterior_distribution.py``:
``calculate_threshold_map`` :10-30, ``estimate_score_update_posterior`` :32-66 and the sampling loop of
``DiffusionClassConditionalGuidedPosteriorDistribution.__call__`` :132-154, plus the plain loop of
``pipeline_sampler_class_conditional.py`` ``generate`` :38-58.  Pinned by tests/golden/g10_posterior_pipeline.npz
(the two functions: outputs of the reference itself; the loops: see make_golden.py g10 for how they were run).
"""
from math import sqrt

import torch

from . import ddim
from .guidance import quantile_linear


def threshold_map(threshold, i, u, threshold_type):
    """:10-30.  float threshold -> per-sample torch.quantile; tensor -> threshold[i] (broadcast over the batch for 4-D u)."""
    if isinstance(threshold, float):
        q = torch.from_numpy(quantile_linear(u.flatten(1).float().numpy(), threshold)).view(u.shape[0], *([1] * (u.dim() - 1)))
        m = u > q if threshold_type == "higher" else u < q
        return m.float()
    thr = threshold[i]
    if u.dim() == 4:
        thr = thr.unsqueeze(0)
    m = u > thr if threshold_type == "higher" else u < thr
    return m.float()


def score_update_posterior(M, denoise, x, eps, alpha_hat_t, draw=torch.randn_like):
    """:32-66.  ``denoise(x_hat) -> eps_hat``; alpha_hat_t a 0-dim fp32 tensor.  Returns (uncertainty, new_score).

    Kept as written: the coefficients are ``math.sqrt`` of 0-dim fp32 tensors (python floats, :55-57) and
    ``pred_epsilon.sum(dim=0)`` reduces the LAST prediction over the batch axis (:65).
    """
    sb, sa = sqrt(1 - alpha_hat_t), sqrt(alpha_hat_t)
    x0 = (x - sb * eps) / sa
    preds = []
    pred = None
    for _ in range(M):
        x_hat = sa * x0 + sb * draw(x)
        pred = denoise(x_hat)
        preds.append(pred)
    preds.append(eps)
    u = torch.var(torch.stack(preds, dim=0), dim=0, unbiased=True)
    inv = 1 / u
    post_precision = 1 / ((M * inv) + (1 / alpha_hat_t))
    return u, post_precision * (inv * pred.sum(dim=0))


def to_uint8(x):
    return ((x / 2 + 0.5).clamp(0, 1) * 255.0).round().to(torch.uint8)


def sample_plain(tab, denoise, X_T, y, batch_size, n_steps, timesteps, clip_sample=True):
    """pipeline_sampler_class_conditional.py :38-58 / :96-134 with a plain DDIM scheduler (eta = 0)."""
    images, scores = [], []
    for s in range(0, X_T.shape[0], batch_size):
        x, yb = X_T[s:s + batch_size].clone(), y[s:s + batch_size]
        sc = []
        for t in timesteps:
            eps = denoise(x, int(t), yb)
            sc.append(eps)
            x, _, _ = ddim.ddim_update(tab, eps, int(t), x, n_steps, clip_sample=clip_sample)
        images.append(to_uint8(x))
        scores.append(torch.stack(sc, dim=1))
    return torch.cat(images), torch.cat(scores)


def sample_posterior_guided(tab, denoise, X_T, y, batch_size, n_steps, timesteps, threshold, M, start_step, num_steps,
                            threshold_type="higher", clip_sample=True, draw=torch.randn_like, step_hook=None):
    """:121-171 with a plain DDIM scheduler: per step a DDIM update; inside start_step <= i <= start_step+num_steps the
    score is replaced by the masked posterior score and the DDIM update is redone.  alphas_cumprod is indexed by the LOOP
    index i (:146), as upstream.  ``step_hook(x, t)`` runs where ``scheduler.step`` is called: a scheduler that consumes
    random numbers inside step() (the reference's uncertainty schedulers draw ``best_noise`` every call) is emulated by
    drawing there."""
    step_hook = step_hook or (lambda x, t: None)
    images = []
    for s in range(0, X_T.shape[0], batch_size):
        x, yb = X_T[s:s + batch_size].clone(), y[s:s + batch_size]
        for i, t in enumerate(timesteps):
            t = int(t)
            eps = denoise(x, t, yb)
            step_hook(x, t)
            prev, _, _ = ddim.ddim_update(tab, eps, t, x, n_steps, clip_sample=clip_sample)
            if (start_step + num_steps) >= i >= start_step:
                u, post = score_update_posterior(M, lambda xh: denoise(xh, t, yb), x, eps, tab.alphas_cumprod[i], draw)
                m = threshold_map(threshold, i, u, threshold_type)
                eps = post * m + eps * (1 - m)
                step_hook(x, t)
                prev, _, _ = ddim.ddim_update(tab, eps, t, x, n_steps, clip_sample=clip_sample)
            x = prev
        images.append(to_uint8(x))
    return torch.cat(images)
