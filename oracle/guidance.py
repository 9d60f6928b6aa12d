"""Quantile mask and percentile-thresholded posterior score update (oracle; test infrastructure).

Follows reference ``uncertainty_guidance.py``
``get_uncertainty_guided_score_with_percentile`` :61-131 (posterior branch
:101-120) and the ``torch.quantile`` 'linear' semantics the reference relies on
at :112 (fp32 rank arithmetic).  numpy only.
"""
import numpy as np


def quantile_linear(u_flat, p):
    """torch.quantile(u, p, dim=1) for fp32 rows, bit-exact (a14).

    rank = float32(p) * (n - 1) in fp32; q = s[lo] + (s[hi] - s[lo]) * w in fp32
    (torch evaluates lerp(a, b, w) as a + w*(b-a) for w < 0.5 and
    b - (b-a)*(1-w) otherwise).
    """
    u = np.ascontiguousarray(u_flat, dtype=np.float32)
    s = np.sort(u, axis=1)
    n = u.shape[1]
    rank = np.float32(p) * np.float32(n - 1)
    lo = np.floor(rank)
    hi = np.ceil(rank)
    w = np.float32(rank - lo)
    a, b = s[:, int(lo)], s[:, int(hi)]
    d = (b - a).astype(np.float32)
    if w < np.float32(0.5):
        return (a + w * d).astype(np.float32)
    return (b - d * (np.float32(1) - w)).astype(np.float32)


def quantile_mask(u, p):
    """mask = u > quantile(u.flatten(1), p) per sample, strict (:112). Returns (mask uint8, q)."""
    B = u.shape[0]
    q = quantile_linear(u.reshape(B, -1), p)
    mask = u > q.reshape((B,) + (1,) * (u.ndim - 1))
    return mask.astype(np.uint8), q


def posterior_blend(eps, u, mask, M, alpha_hat_t):
    """:115-120 with B == 1 semantics kept as written: eps.sum(dim=0) sums the batch axis."""
    eps = eps.astype(np.float32)
    inv = (np.float32(1) / u).astype(np.float32)
    tr = (np.float32(M) * inv + np.float32(1) / np.float32(alpha_hat_t)).astype(np.float32)
    prec = (np.float32(1) / tr).astype(np.float32)
    post = prec * (inv * eps.sum(axis=0))
    m = mask.astype(np.float32)
    return (eps * (np.float32(1) - m) + m * post).astype(np.float32)


def guided_score_percentile(eps, x, denoise_cfg, alpha_hat_t, percentile, M, draw, use_posterior=True):
    """:83-120, model_type='stable-diffusion', posterior mode.

    ``denoise_cfg(x_hat) -> eps_hat`` wraps the CFG UNet call + combine
    (``predict_model_stable_diffusion`` :135-147).  ``x`` may be the CFG-doubled
    latent [2B,...] with eps [B,...] (B must be 1, broadcasting as written :86).
    """
    a = np.float32(alpha_hat_t)
    x0 = (x - np.sqrt(np.float32(1) - a) * eps) / np.sqrt(a)
    samples = []
    for _ in range(M):
        x_hat = (np.sqrt(a) * x0 + np.sqrt(np.float32(1) - a) * draw(eps)).astype(np.float32)
        samples.append(denoise_cfg(x_hat))
    if use_posterior:
        samples.append(eps)
    st = np.stack(samples, 0).astype(np.float32)
    u = st.var(axis=0, ddof=1, dtype=np.float32)
    mask, q = quantile_mask(u, percentile)
    return posterior_blend(eps, u, mask, M, alpha_hat_t), u, mask, q


def guided_score_percentile_gradient(eps, x, unet, t, y, guidance_scale, alpha_hat_t, percentile, M, draw, lr=1.0):
    """:83-108,121-129 with use_posterior = False (the reference script's default, scripts/generate_with_uncertainty_threshold_
    stable_diffusion.py:16-17), model_type='stable-diffusion', as written: torch autograd through the M CFG denoiser calls.

    ``unet(x2 [2,4,h,w], t, y [2,77,ctx]) -> [2,4,h,w]`` must be autograd-traceable (oracle.sd_unet.forward_autograd);
    eps [1,4,h,w], x the CFG-doubled latent [2,4,h,w], ``draw(like)`` = torch.randn_like.  Returns (final, grad, u, mask).
    """
    import torch
    pe = eps.detach().clone().float().requires_grad_(True)
    a = torch.as_tensor(alpha_hat_t, dtype=torch.float32)
    hats = []
    with torch.enable_grad():
        x0 = (x - torch.sqrt(1 - a) * pe) / torch.sqrt(a)                                   # :86
        for _ in range(M):
            x_hat = torch.sqrt(a) * x0 + torch.sqrt(1 - a) * draw(pe)                       # :88
            out = unet(x_hat, t, y)
            un, tx = out.chunk(2)                                                           # :142-145
            hats.append(un + guidance_scale * (tx - un))
        st = torch.stack(hats, dim=0)                                                       # :103 (nothing appended in this mode)
        u = torch.var(st, dim=0)                                                            # :106
        u.mean(dim=0).sum().backward()                                                      # :107-108
    ud = u.detach()
    q = torch.quantile(ud.float().flatten(1), percentile, dim=1, keepdim=True).view(ud.shape[0], *([1] * (ud.ndim - 1)))
    mask = (ud > q).float()                                                                 # :112-113
    grad = pe.grad * 1                                                                      # :122
    return (pe.detach() + lr * grad * mask), grad, ud, mask                                 # :129
