"""Noise-schedule tables and timestep grids (oracle; test infrastructure).

Follows reference ``schedulers_uncertainty/scheduling_ddim_uncertainty_zigzag_centered.py``
``__init__`` :194-275 and ``set_timesteps`` :338-387, and ``init_model.py``
:134-157 (ADM-64 cosine betas).  All tables are fp32 like the reference's.
"""
import math

import numpy as np
import torch


def cosine_trained_betas(n=1000, max_beta=0.999):
    """init_model.py:134-157 - discretised cos^2 alpha-bar, float64 numpy array."""
    def abar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return np.array([min(1 - abar((i + 1) / n) / abar(i / n), max_beta) for i in range(n)])


def make_betas(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, beta_schedule="linear",
               trained_betas=None):
    """zigzag_centered.py:218-231."""
    if trained_betas is not None:
        return torch.tensor(trained_betas, dtype=torch.float32)
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    raise NotImplementedError(beta_schedule)


class Tables:
    """betas / alphas_cumprod / final alpha (zigzag_centered.py:237-244)."""

    def __init__(self, set_alpha_to_one=True, **kw):
        self.betas = make_betas(**kw)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps = self.betas.shape[0]


def leading_timesteps(num_train_timesteps, n, steps_offset=0):
    """zigzag_centered.py:364-369 ('leading' spacing)."""
    ratio = num_train_timesteps // n
    return (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + steps_offset


def uc_window(timesteps, after_step, num_steps_uc):
    """zigzag_centered.py:383-384 - (timestep_after_step, timestep_end_step)."""
    return int(timesteps[after_step]), int(timesteps[after_step + num_steps_uc - 1])
