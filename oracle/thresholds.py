"""Dataset-level pixel-wise thresholds (oracle; test infrastructure).

Restates reference ``scripts/compute_threshold_pixel_wise.py`` :86-100 / :143-165 with the same torch ops (fp16 cast,
``argsort(dim=0)``, index at rank ``int(N*perc)``, ``gather``).  The script itself is not importable here (it needs
``path`` / ``omegaconf`` and runs at import), so this row's parity is pinned by construction, not by a reference run.
"""
import torch


def compute_threshold_pixel_wise(uncertainties, perc):
    u = uncertainties.cpu().half()
    n, S = u.shape[0], u.shape[1]
    out = []
    for i in range(S):
        ut = u[:, i]
        idx = ut.argsort(dim=0)
        sel = idx[int(n * perc)].unsqueeze(0)
        out.append(ut.gather(dim=0, index=sel).squeeze(0))
    return torch.stack(out, dim=0)
