"""Same-box A/B of the f16x3 flash attention kernel at the models' shapes: attn_ab.py [alternative libdua_hip.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffusion_uncertainty_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib.LIB_PATH = sys.argv[1]
import diffusion_uncertainty_amd.ops as O
dev = "cuda"
for name, B, T, heads, ch in (("ADM-128 32^2", 128, 1024, 8, 64), ("ADM-128 16^2", 128, 256, 12, 64), ("ADM-128 8^2", 128, 64, 16, 64), ("U-ViT-H/2 b64", 64, 258, 16, 72),
                              ("U-ViT-H/2 b320", 320, 258, 16, 72)):
    C = heads * ch
    qkv = torch.randn(B * T, 3 * C, device=dev)
    fl = 4.0 * B * heads * T * T * ch
    out = O.qkv_attention(qkv, B, T, heads, ch, precision="f16x3")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        O.qkv_attention(qkv, B, T, heads, ch, precision="f16x3")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:16s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF   checksum {out.double().abs().sum().item():.6e}")
