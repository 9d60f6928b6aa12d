"""Dataset-level pixel-wise uncertainty thresholds.

Mirror of the computation in reference ``scripts/compute_threshold_pixel_wise.py`` (``main`` :44-116 and
``load_uncertainty_datasets`` :118-165): the uncertainty maps of a generated dataset ``[N, S, 3, H, W]`` are cast to
fp16 and, per timestep and pixel, the value at rank ``int(N * perc)`` of the ascending order over the N samples is taken
(``argsort(dim=0)`` + ``gather``; no interpolation) -> thresholds ``[S, 3, H, W]`` fp16.  The selection runs in
libdua_hip.so (dua_pixelwise_order_statistic, exact radix select per pixel column); the result is the reference's, bit for
bit (G14: outputs of the reference script's own ``load_uncertainty_datasets``).
"""
import fnmatch
import os

import torch

from . import ops


@torch.no_grad()
def compute_threshold_pixel_wise(uncertainties, perc=0.15):
    """uncertainties: [N, S, C, H, W] fp32 (GPU) -> [S, C, H, W] fp16, bit-identical to the reference's result."""
    num_samples = uncertainties.shape[0]
    k = int(num_samples * perc)
    return ops.pixelwise_order_statistic(ops._f32c(uncertainties), k)


def _files(folder, pattern):
    return sorted(os.path.join(folder, f) for f in os.listdir(folder) if fnmatch.fnmatch(f, pattern) and os.path.isfile(os.path.join(folder, f)))


@torch.no_grad()
def load_uncertainty_datasets(dataset_folders, perc, device="cuda", min_samples=100):
    """reference script :118-165: one ``[1, S, C, H, W]`` fp16 threshold tensor per ``uncertainty_*.pth`` file of the folders; files
    with fewer than 100 samples are skipped (:139-140).  The files are the per-rank outputs of
    scripts/generate_dataset_score_uncertainty_imagenet.py."""
    out = []
    for folder in dataset_folders:
        folder = str(folder)
        unc_files = _files(folder, "uncertainty_*.pth")
        assert len(unc_files) > 0, f"no uncertainty files found in {folder}"
        assert len(_files(folder, "gen_images_*.pth")) > 0, f"no gen_images files found in {folder}"
        for f in unc_files:
            u = torch.load(f, map_location="cpu")
            if u.shape[0] < min_samples:
                continue
            out.append(compute_threshold_pixel_wise(u.to(device), perc).unsqueeze(0).cpu())
    return out


@torch.no_grad()
def thresholds_from_datasets(dataset_folders, perc, device="cuda"):
    """reference ``main`` :78-100: one file -> its thresholds; several -> the same order statistic over the per-file thresholds
    (rank ``int(F * perc)`` over the F files, per timestep and pixel).  -> [S, C, H, W] fp16 (CPU)."""
    per_file = load_uncertainty_datasets(dataset_folders, perc, device)
    if not per_file:
        raise ValueError("no uncertainty file with at least 100 samples in " + ", ".join(str(f) for f in dataset_folders))
    if len(per_file) == 1:
        return per_file[0].squeeze(0)
    stacked = torch.cat(per_file, dim=0)                                  # [F, S, C, H, W] fp16: exact in fp32, selected back exactly
    return compute_threshold_pixel_wise(stacked.to(device).float(), perc).cpu()
