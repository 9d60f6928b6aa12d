"""Dataset-level pixel-wise uncertainty thresholds.

Mirror of the computation in reference ``scripts/compute_threshold_pixel_wise.py`` (``main`` :86-100 and
``load_uncertainty_datasets`` :143-165): the uncertainty maps of a generated dataset ``[N, S, 3, H, W]`` are cast to
fp16 and, per timestep and pixel, the value at rank ``int(N * perc)`` of the ascending order over the N samples is taken
(``argsort(dim=0)`` + ``gather``; no interpolation) -> thresholds ``[S, 3, H, W]`` fp16.  The selection runs in
libdua_hip.so (dua_pixelwise_order_statistic, exact radix select per pixel column).
"""
import torch

from . import ops


@torch.no_grad()
def compute_threshold_pixel_wise(uncertainties, perc=0.15):
    """uncertainties: [N, S, C, H, W] fp32 (GPU) -> [S, C, H, W] fp16, bit-identical to the reference's result."""
    num_samples = uncertainties.shape[0]
    k = int(num_samples * perc)
    return ops.pixelwise_order_statistic(ops._f32c(uncertainties), k)
