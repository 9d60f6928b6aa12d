"""Diagnostic: when do the 256 persistent workgroups of k_wconv4 reach their epilogues?  Needs the timing build -DWCONV4_DIAG=64
(scripts/wconv_diag.py --build --kernel w4): wave 0 of every workgroup writes the cycle counter at the start and the end (stores drained)
of each tile's epilogue into the column-sum buffer.  usage: wconv4_timeline.py <lib> [H Cin Cout res]"""
import ctypes, sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffusion_uncertainty_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from diffusion_uncertainty_amd import ops
H, Cin, Cout, res = (int(v) for v in (sys.argv[2:6] if len(sys.argv) > 5 else (128, 256, 256, 1)))
N, dev = 128, "cuda"
x = torch.randn(N, H, H, Cin, device=dev)
st = torch.tensor([0.0, 1.0], device=dev).repeat(N * 32, 1).contiguous()
xv = ops.groupnorm_apply(x, st, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), act=0, split="f16x3w4")
wp = ops.pack_weight_f16x3_wino4(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02)
r = torch.randn(N, H, H, Cout, device=dev) if res else None
y = torch.empty(N, H, H, Cout, device=dev)
ws = torch.zeros(max(256 * 64 * 4 * 2, (N * H * H // 64) * 2 * Cout), device=dev, dtype=torch.float32)
L = _lib.lib()
for _ in range(2):
    rc = L.dua_conv3x3_f16x3_wino4(xv.buf.data_ptr(), wp.data_ptr(), None, r.data_ptr() if res else None, 0, y.data_ptr(), N, H, H, Cin, Cout, ws.data_ptr(), xv.scale,
                                   torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
t = ws[:256 * 64 * 4 * 2].view(torch.int64).reshape(256, 64, 4).cpu().double()
ntile = int((t[0, :, 0] > 0).sum())
t0 = t[:, 0, 0].min()
t = (t - t0) / 100.0        # s_memtime: 100 MHz -> us
for k in list(range(0, min(ntile, 6))) + list(range(max(6, ntile - 3), ntile)):
    s, e, mid = t[:, k, 0], t[:, k, 1], t[:, k, 2]
    print(f"tile {k:2d}: epilogue start min {s.min():8.1f} median {s.median():8.1f} max {s.max():8.1f} us | spread (p10..p90) {s.kthvalue(26).values:8.1f} .. {s.kthvalue(230).values:8.1f}"
          f" | duration median {(e - s).median():6.1f} us (min {(e - s).min():5.1f}, max {(e - s).max():5.1f}), of which issue {(mid - s).median():6.1f} us, drain {(e - mid).median():6.1f} us")
