"""Debug aid: where does the F(4,3) conv differ from fp64?  usage: wino4_debug.py N H W Cin Cout"""
import sys, math, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from diffusion_uncertainty_amd import ops
N, H, W, Cin, Cout = [int(a) for a in sys.argv[1:6]]
dev = "cuda"
g = torch.Generator().manual_seed(0)
x = torch.randn(N, Cin, H, W, generator=g)
w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
ref = F.conv2d(x.double(), w.double(), padding=1)
st = torch.tensor([0.0, 1.0], device=dev).repeat(N * 32, 1).contiguous()
one, zero = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
xv = ops.groupnorm_apply(xd, st, one, zero, act=0, split="f16x3w4")
y = ops.conv3x3_bf16x3(xv, ops.pack_weight_f16x3_wino4(w.to(dev)), None, Cout)
torch.cuda.synchronize()
d = (y.cpu().permute(0, 3, 1, 2).double() - ref).abs()
bad = d > 1e-4 * ref.abs().max()
print("max err", d.max().item(), "bad count", int(bad.sum()), "of", bad.numel())
if bad.any():
    idx = bad.nonzero()
    for name, k in (("n", 0), ("c", 1), ("y", 2), ("x", 3)):
        u = idx[:, k].unique()
        print(name, u.tolist()[:64], "..." if len(u) > 64 else "")
