#!/usr/bin/env python3
"""One on-the-fly f16x3 GEMM shape (1x1 skip conv of the ADM up path), a few launches: gemm_one.py M N K1 K2"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import diffusion_uncertainty_amd.ops as O  # noqa: E402

M, N, K1, K2 = (int(v) for v in sys.argv[1:5])
a1 = torch.randn(M, K1, device="cuda")
a2 = torch.randn(M, K2, device="cuda") if K2 else None
w = torch.randn(N, K1 + K2, device="cuda") / (K1 + K2) ** 0.5
packed = O.pack_weight_f16x3(w)
b = torch.randn(N, device="cuda")
for _ in range(3):
    O.gemm_f16x3(a1, packed, N, bias=b, a2=a2, colsum=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    O.gemm_f16x3(a1, packed, N, bias=b, a2=a2, colsum=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"M={M} N={N} K={K1}+{K2}: {ms:.3f} ms  {2.0 * M * N * (K1 + K2) / ms / 1e9:.1f} TF algorithmic")
