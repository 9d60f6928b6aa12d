import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_uncertainty_amd import ops, _lib
dev = "cuda"
torch.manual_seed(0)
N, H, Cin, Cout = 2, 8, 64, 128
x = torch.randn(N, H, H, Cin, device=dev)
st = torch.tensor([0.0, 1.0], device=dev).repeat(N * 32, 1).contiguous()
xs = ops.groupnorm_apply(x, st, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), act=0, split="f16f8")
buf = xs.buf.view(torch.uint8).reshape(2, -1)
hi = buf[0].view(torch.float16).reshape(N, H, H, Cin).float()
x8 = buf[1].reshape(N * H * H, Cin // 32, 2, 32)
lo8 = x8[:, :, 0].reshape(N, H, H, Cin).view(torch.float8_e4m3fn).float() / 4096
h8 = x8[:, :, 1].reshape(N, H, H, Cin).view(torch.float8_e4m3fn).float()
print("A hi err", (hi - x.half().float()).abs().max().item(), "lo err rel", ((x - hi) - lo8).abs().max().item() * 4096,
      "hi8 err", (h8 - x).abs().max().item(), "nan?", torch.isnan(lo8).any().item(), torch.isnan(h8).any().item())
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02
pk = ops.pack_weight_f16f8(w)
K = 9 * Cin
Npad = (Cout + 127) // 128 * 128
plane = Npad * K
whi = pk[:2 * plane].view(torch.float16).reshape(Npad, K).float()
wx = pk[2 * plane:4 * plane].reshape(Npad, K // 32, 2, 32)
sc = pk[4 * plane:4 * plane + Npad].view(torch.int8).float()
print("scales", sc[:8].tolist(), sc.min().item(), sc.max().item())
wk = w.reshape(Cout, Cin // 32, 32, 9).permute(0, 1, 3, 2).reshape(Cout, K)      # k = cb*288 + tap*32 + c
print("W hi err", (whi[:Cout] - wk.half().float()).abs().max().item())
w8 = wx[:, :, 0].reshape(Npad, K).view(torch.float8_e4m3fn).float()[:Cout] / torch.exp2(sc[:Cout, None])
wl8 = wx[:, :, 1].reshape(Npad, K).view(torch.float8_e4m3fn).float()[:Cout] / torch.exp2(sc[:Cout, None] + 12)
print("w8 rel err", ((w8 - wk).abs() / wk.abs().clamp_min(1e-3)).max().item(), "wlo8 err", (wl8 - (wk - wk.half().float())).abs().max().item(),
      "nan?", torch.isnan(w8).any().item(), torch.isnan(wl8).any().item())
y = ops.conv3x3_bf16x3(xs, pk, None, Cout)
print("conv nan count", torch.isnan(y).sum().item(), "of", y.numel())
ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, padding=1).permute(0, 2, 3, 1)
# emulate the exact decomposition
def c(a, b): return torch.nn.functional.conv2d(a.permute(0, 3, 1, 2).double(), b.double(), None, padding=1).permute(0, 2, 3, 1)
whi4 = w.half().float(); wl4 = w - whi4
print("hi*hi only vs y:", (c(hi, whi4) - y.double()).abs().max().item(), " full emu vs ref", (c(hi, whi4) + c(lo8, w) + c(h8, wl4) - ref).abs().max().item())
print("y vs ref", (y.double() - ref).abs().max().item())
