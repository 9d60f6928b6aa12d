import sys, os, torch
sys.path.insert(0, os.getcwd())
from diffusion_uncertainty_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = sys.argv[1]
from diffusion_uncertainty_amd import ops
dev="cuda"
for H, C in ((128, 256), (64, 512), (32, 512), (16, 768)):
    N=128
    x = torch.randn(N, H, H, C, device=dev)
    st = torch.tensor([0.1, 0.9], device=dev).repeat(N * 32, 1).contiguous()
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    film = torch.randn(N, 2 * C, device=dev)
    for split in ("f16x3w", "f16x3"):
        for _ in range(2): ops.groupnorm_apply(x, st, g, b, film=film, act=1, split=split)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.groupnorm_apply(x, st, g, b, film=film, act=1, split=split)
        e1.record(); torch.cuda.synchronize()
        print(f"{H}^2 x {C} {split}: {e0.elapsed_time(e1)/10:.4f} ms")
