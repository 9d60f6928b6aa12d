#!/usr/bin/env python3
"""One ADM-128 level-0 skip GEMM (dua_gemm_f16x3, M = 128 * 128^2, K = 256 + 256, N = 256) a few times: driver for scripts/pmc_kernel.sh."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffusion_uncertainty_amd import ops
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
C1, C2, Cout = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (256, 256, 256)
M = 128 * H * H
a1, a2 = torch.randn(M, C1, device="cuda"), torch.randn(M, C2, device="cuda")
pk = ops.pack_weight_f16x3(torch.randn(Cout, C1 + C2, device="cuda") / (C1 + C2) ** 0.5)
for _ in range(3):
    ops.gemm_f16x3(a1, pk, Cout, a2=a2)
torch.cuda.synchronize()
