#!/usr/bin/env python3
"""Micro-benchmark of the GroupNorm-apply kernel (HBM-bound: reads fp32, writes the conv operand planes) at ADM-128 shapes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_uncertainty_amd import _lib, ops
if os.environ.get("DUA_DIAG_LIB"):                     # A/B: an alternative build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["DUA_DIAG_LIB"])

dev = "cuda"
L = _lib.lib()
for H, C in ((128, 256), (64, 256), (64, 512), (32, 512), (32, 1024), (16, 768), (16, 1536), (8, 1024), (8, 2048), (128, 512)):
    N = 128
    C1 = C if C not in (512, 1024, 1536, 2048) or (H, C) in ((32, 512),) else C // 2
    x = torch.randn(N, H, H, C1, device=dev)
    x2 = torch.randn(N, H, H, C - C1, device=dev) if C1 != C else None
    st = torch.tensor([0.1, 0.9], device=dev).repeat(N * 32, 1).contiguous()
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    film = torch.randn(N, 2 * C, device=dev)
    for split in ((sys.argv[1],) if len(sys.argv) > 1 else (False, True, "f16x3", "f16x3w", "f16w")):
        for wide in (0, 1):
            L.dua_set_gn_apply_wide(wide)
            ops.groupnorm_apply(x, st, g, b, x2=x2, film=film, act=1, split=split)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.groupnorm_apply(x, st, g, b, x2=x2, film=film, act=1, split=split)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            by = N * H * H * C * (10.0 if split == "f16x3w4" else 8.0)          # F(4,3) images: 4 B read, 6 B written per element
            print(f"H={H:4d} C={C:4d} split={str(split):6s} wide={wide}: {ms:7.3f} ms  {by / ms / 1e9:6.2f} TB/s", flush=True)
