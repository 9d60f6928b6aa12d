"""Noise sources for the sampler.

``HostNoise`` (default) draws from torch's global CPU generator in exactly the order the reference's
``torch.randn_like`` calls consume it (zigzag_centered.py:500,532) and uploads the tensor - the "noise
tape" that makes fixed-seed runs comparable with the reference CPU path.  ``DeviceNoise`` uses the
library's Philox kernel (dua_randn): no host work, a different (but reproducible) stream.
"""
import torch

from .. import ops


class HostNoise:
    def draw(self, shape, device):
        return torch.randn(tuple(shape), dtype=torch.float32).to(device, non_blocking=True)


class DeviceNoise:
    def __init__(self, seed=0):
        self.seed, self.offset = int(seed), 0

    def draw(self, shape, device):
        n = 1
        for s in shape:
            n *= int(s)
        out = ops.randn(tuple(shape), device, self.seed, self.offset)
        self.offset += (n + 3) // 4
        return out

    def dropout_mask(self, shape_nhwc, p):
        n = 1
        for s in shape_nhwc:
            n *= int(s)
        out = ops.dropout_mask(tuple(shape_nhwc), "cuda", p, self.seed, self.offset)
        self.offset += (n + 3) // 4
        return out
