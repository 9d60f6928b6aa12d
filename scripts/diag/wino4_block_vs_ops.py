"""Debug aid: ADM ResBlocks in the F(4,3) form, block-level C ABI vs per-op composition, bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from oracle import adm
from test_gpu_models import build_adm
from diffusion_uncertainty_amd.guided_diffusion import unet_openai as U
dev = "cuda"
m, _ = build_adm(dict(adm.ADM64), 1)
m.precision = "f16x3"
m.winograd = 4
N = 2
emb = torch.randn(N, m.time_embed[2].out_features if hasattr(m, "time_embed") else 768, device=dev)
blocks = [(n, b) for n, b in m.named_modules() if isinstance(b, U.ResBlock)]
print(len(blocks), "ResBlocks")
g = torch.Generator(device=dev).manual_seed(0)
for name, blk in blocks:
    cin = blk.channels
    for H in (16, 32):
        x = torch.randn(N, H, H, cin, device=dev, generator=g)
        outs = []
        for abi in (True, False):
            m.use_block_abi = abi
            ctx = U._Ctx(m)
            ctx.emb = emb
            try:
                ctx.prepare_film(emb) if hasattr(ctx, "prepare_film") else None
            except Exception:
                pass
            try:
                y = blk.run(ctx, x)
            except Exception as e:
                print(name, H, "ERR", repr(e)[:200]); outs = None; break
            outs.append(y.clone())
        if outs is None:
            break
        d = (outs[0] - outs[1]).abs().max().item()
        print(f"{name:28s} H={H} cin={cin} cout={blk.out_channels} updown={getattr(blk,'resample',0)}: max|abi - ops| = {d:.3e}")
