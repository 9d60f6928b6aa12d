#!/usr/bin/env python3
"""Micro-benchmark of the attention paths at the shapes of the models (algorithmic FLOPs = 4*B*heads*T^2*ch)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diffusion_uncertainty_amd.ops as O

dev = "cuda"
for name, B, T, heads, ch in (("ADM-128 32^2", 128, 1024, 4, 128), ("ADM-64 32^2", 128, 1024, 3, 64), ("ADM-64 16^2", 128, 256, 6, 64),
                              ("U-ViT-H/2", 64, 258, 16, 72), ("SD-1.5 64^2", 2, 4096, 8, 40)):
    C = heads * ch
    qkv = torch.randn(B * T, 3 * C, device=dev)
    fl = 4.0 * B * heads * T * T * ch
    ref = None
    for mode, prec in (("fused", "fp32"), ("gemm", "bf16x3"), ("fused_x", "bf16x3")):
        if mode == "gemm" and (ch % 32 or T % 32):
            continue
        O.ATTENTION_MODE = mode
        out = O.qkv_attention(qkv, B, T, heads, ch, precision=prec)
        torch.cuda.synchronize()
        if ref is None:
            ref = out
        err = (out - ref).abs().max().item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            O.qkv_attention(qkv, B, T, heads, ch, precision=prec)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{name:14s} {mode:8s} {prec:7s}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF   max|d vs fp32 flash| {err:.2e}", flush=True)
O.ATTENTION_MODE = "auto"
