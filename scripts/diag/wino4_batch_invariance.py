"""Debug aid: is the F(4,3) writer / conv result of an image independent of the batch it runs in?"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffusion_uncertainty_amd import ops
dev = "cuda"
H, Cin, Cout = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (128, 256, 256)))
g = torch.Generator().manual_seed(0)
x9 = torch.randn(9, H, H, Cin, generator=g).to(dev)
w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dev)
b = torch.randn(Cout, generator=g).to(dev)
gamma, beta = torch.randn(Cin, generator=g).to(dev), torch.randn(Cin, generator=g).to(dev)
res9 = torch.randn(9, H, H, Cout, generator=g).to(dev)
wp = ops.pack_weight_f16x3_wino4(w)
def run(x, res):
    st = torch.tensor([0.1, 0.9], device=dev).repeat(x.shape[0] * 32, 1).contiguous()     # fixed statistics: the stand-alone statistics kernel chunks by batch
    v = ops.groupnorm_apply(x, st, gamma, beta, act=1, split="f16x3w4")
    y = ops.conv3x3_bf16x3(v, wp, b, Cout, residual=res, colsum=True)
    return st, v, y
st9, v9, y9 = run(x9, res9)
st1, v1, y1 = run(x9[0:1].contiguous(), res9[0:1].contiguous())
print("stats equal:", torch.equal(st9[0:1], st1))
P = (H // 4) * H
ntile1 = (P + 127) // 128
vb9 = v9.buf[:, :ntile1]; vb1 = v1.buf[:, :ntile1]
print("V planes (image 0 tiles) equal:", torch.equal(vb9, vb1), "max diff", (vb9.view(torch.float16).float() - vb1.view(torch.float16).float()).abs().max().item())
print("conv out equal:", torch.equal(y9[0:1], y1), "max diff", (y9[0:1] - y1).abs().max().item())
cs9, cs1 = y9._dua_colsums, y1._dua_colsums
print("colsums equal:", torch.equal(cs9[: cs1.shape[0]], cs1), (cs9[: cs1.shape[0]] - cs1).abs().max().item())
