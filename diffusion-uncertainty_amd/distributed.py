"""Multi-GPU plumbing: the path shards by independent batches, one process per GPU.

Reference: scripts/generate_dataset_score_uncertainty_imagenet.py:51,137-144 - `mp.spawn`, rank r takes
`X_T[start + r*n : start + (r+1)*n]` with n = N // world (remainder dropped), no collective on the path.
Here the launcher is torch.distributed.run; the only collective is an end-of-job all-gather of the uint8
images (RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests), plus a MAX all-reduce of the wall time.
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world, start=0):
    n = n_total // world
    return start + rank * n, start + (rank + 1) * n


def gather_images(imgs):
    """imgs: uint8 [n, C, H, W] on every rank -> [world, n, C, H, W] on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return imgs.unsqueeze(0)
    w = dist.get_world_size()          # a 1-rank group still goes through the collective (exercises RCCL under torchrun)
    out = torch.empty((w * imgs.shape[0],) + tuple(imgs.shape[1:]), dtype=imgs.dtype, device=imgs.device)
    dist.all_gather_into_tensor(out, imgs.contiguous())          # concatenated along dim 0, rank order
    return out.view((w,) + tuple(imgs.shape))


def max_over_ranks(seconds, device):
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(seconds, device):
    """The wall time of every rank, in rank order (list of floats; a one-element list without a process group)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(seconds)]
    w = dist.get_world_size()
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    out = torch.empty(w, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, t)
    return [float(v) for v in out.tolist()]
