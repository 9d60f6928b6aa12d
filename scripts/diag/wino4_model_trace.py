"""Debug aid: first block whose output differs between the block-level C ABI and the per-op composition (full-size ADM-64, winograd = 4)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from oracle import adm
from test_gpu_models import build_adm
from diffusion_uncertainty_amd.guided_diffusion import unet_openai as U
dev = "cuda"
m, _ = build_adm(dict(adm.ADM64), 1)
m.precision = "f16x3"
m.winograd = int(sys.argv[1]) if len(sys.argv) > 1 else 4
x = torch.randn(2, 3, 64, 64, device=dev); t = torch.tensor([500, 20], device=dev); y = torch.tensor([1, 2], device=dev)
names = {id(b): n for n, b in m.named_modules()}
trace = {}
def wrap(cls):
    orig = cls.run
    def run(self, ctx, *a, **k):
        out = orig(self, ctx, *a, **k)
        cs = getattr(out, "_dua_colsums", None)
        trace.setdefault(cur[0], []).append((names[id(self)], out.detach().clone(), None if cs is None else cs.detach().clone(),
                                            None if cs is None else getattr(out, "_dua_colsums_rows", None)))
        return out
    cls.run = run
cur = [None]
for cls in (U.ResBlock, U.AttentionBlock):
    wrap(cls)
for abi in (True, False):
    m.use_block_abi = abi
    cur[0] = abi
    m(x, t, y=y)
for (n1, o1, c1, _), (n2, o2, c2, _) in zip(trace[True], trace[False]):
    d = (o1 - o2).abs().max().item()
    dc = None
    if c1 is not None and c2 is not None and c1.shape == c2.shape:
        dc = (c1 - c2).abs().max().item()
    print(f"{n1:28s} {tuple(o1.shape)} max|abi - ops| = {d:.3e}  colsums: {'-' if c1 is None else tuple(c1.shape)} vs {'-' if c2 is None else tuple(c2.shape)} diff {dc}")
    if d != 0:
        break
