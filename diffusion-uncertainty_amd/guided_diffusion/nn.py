"""Parameter-holder modules with the reference's names (guided_diffusion/nn.py:17-19,93-100)."""
import torch.nn as nn


class GroupNorm32(nn.GroupNorm):
    """Parameters only; the arithmetic runs in dua_groupnorm_stats / dua_groupnorm_apply."""


def normalization(channels):
    return GroupNorm32(32, channels)
