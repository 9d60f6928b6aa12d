"""Debug aid: first block whose output for sample 0 differs between a batch-9 and a batch-1 forward (full-size ADM-128, f16x3)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from oracle import adm
from test_gpu_models import build_adm
from diffusion_uncertainty_amd.guided_diffusion import unet_openai as U
dev = "cuda"
m, _ = build_adm(dict(adm.ADM128), 128)
m.precision = "f16x3"
m.winograd = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = torch.Generator().manual_seed(12)
B = 9
x = torch.randn(B, 3, 128, 128, generator=g).to(dev)
t = torch.tensor([999, 980, 700, 500, 333, 200, 60, 20, 0]).to(dev)
y = torch.randint(0, 1000, (B,), generator=g).to(dev)
names = {id(b): n for n, b in m.named_modules()}
trace = {}
cur = [None]
def wrap(cls):
    orig = cls.run
    def run(self, ctx, *a, **k):
        out = orig(self, ctx, *a, **k)
        trace.setdefault(cur[0], []).append((names[id(self)], out[0:1].detach().clone()))
        return out
    cls.run = run
for cls in (U.ResBlock, U.AttentionBlock):
    wrap(cls)
cur[0] = 9
m(x, t, y=y)
cur[0] = 1
m(x[0:1].contiguous(), t[0:1], y=y[0:1])
for (n1, o1), (n2, o2) in zip(trace[9], trace[1]):
    d = (o1 - o2).abs().max().item()
    print(f"{n1:28s} {tuple(o1.shape)} max|b9 - b1| = {d:.3e}")
    if d != 0:
        break
