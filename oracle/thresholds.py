"""Dataset-level pixel-wise thresholds (oracle; test infrastructure).

Restates reference ``scripts/compute_threshold_pixel_wise.py`` :86-100 / :143-165 with the same torch ops (fp16 cast,
``argsort(dim=0)``, index at rank ``int(N*perc)``, ``gather``).  Pinned: tests/golden/g14_thresholds.npz holds outputs of the
reference script's own ``load_uncertainty_datasets`` (:118-165), imported from /root/reference through the stand-ins of
tests/golden/refimport.py (``path``, ``diffusion_uncertainty.utils`` / ``.paths``) and run on temporary ``uncertainty_*.pth`` folders.
Files with fewer than 100 samples are skipped by the script (:139-140): ``min_samples``.
"""
import torch


def compute_threshold_pixel_wise(uncertainties, perc, min_samples=0):
    u = uncertainties.cpu().half()
    n, S = u.shape[0], u.shape[1]
    if n < min_samples:
        return None
    out = []
    for i in range(S):
        ut = u[:, i]
        idx = ut.argsort(dim=0)
        sel = idx[int(n * perc)].unsqueeze(0)
        out.append(ut.gather(dim=0, index=sel).squeeze(0))
    return torch.stack(out, dim=0)
