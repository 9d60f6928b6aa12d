#!/usr/bin/env python3
"""One presplit f16x3 GEMM shape, a few launches (driver for scripts/pmc_kernel.sh): gemm_presplit_one.py M N K [hg_mode] [res]
hg_mode: dua_set_hgemm_mode nt (-1 = the previous engine k_xconv_dma / k_hconv, 0 = k_hgemm with its own tile choice)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from diffusion_uncertainty_amd import _lib, ops  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
res = len(sys.argv) > 5 and sys.argv[5] == "res"
assert _lib.lib().dua_set_hgemm_mode(mode, 0) == 0
dev = "cuda"
x = torch.randn(M, K, device=dev) * ops.F16X3_ACT_SCALE
hi = x.half()
pl = ops.SplitPlanes(torch.stack([hi, (x - hi.float()).half()]).contiguous().view(torch.int16), (M, K), "f16x3", ops.F16X3_ACT_SCALE)
pk = ops.pack_weight_f16x3(torch.randn(N, K, device=dev) / K ** 0.5)
b = torch.randn(N, device=dev)
r = torch.randn(M, N, device=dev) if res else None
for _ in range(4):
    ops.gemm_bf16x3_presplit(pl, pk, N, bias=b, residual=r)
torch.cuda.synchronize()
