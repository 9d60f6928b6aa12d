#!/usr/bin/env python3
"""One fp32 flash-attention shape, a few launches: attn32_one.py B T heads ch"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import diffusion_uncertainty_amd.ops as O  # noqa: E402

B, T, heads, ch = (int(v) for v in sys.argv[1:5])
qkv = torch.randn(B * T, 3 * heads * ch, device="cuda")
for _ in range(3):
    O.fused_attention(qkv, B, T, heads, ch)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    O.fused_attention(qkv, B, T, heads, ch)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"fp32 flash B={B} T={T} heads={heads} ch={ch}: {ms:.3f} ms  {4.0 * B * heads * T * T * ch / ms / 1e9:.1f} TF")
