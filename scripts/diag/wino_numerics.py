"""CPU numerics prototype: F(2,3) Winograd along y in emulated f16x3 arithmetic vs direct f16x3 vs fp32, against fp64."""
import torch
torch.manual_seed(0)
N, H, W, Cin, Cout = 2, 16, 16, 256, 64
x = torch.nn.functional.silu(torch.randn(N, Cin, H, W) * 1.5 + 0.3)          # post GN+SiLU-like
w = torch.randn(Cout, Cin, 3, 3) * 0.02
ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)


def split(t, scale):
    s = (t * scale).float()
    hi = s.half()
    lo = (s - hi.float()).half()
    return hi.double(), lo.double()


def x3conv(xh, xl, wh, wl, **kw):
    f = torch.nn.functional.conv2d
    return f(xh, wh, **kw) + f(xh, wl, **kw) + f(xl, wh, **kw)


def rowscale(wt):            # per output row power of two so that the row max lands in [2^14, 2^15)
    m = wt.abs().flatten(1).max(1).values
    s = torch.floor(14 - torch.log2(m))
    return torch.exp2(s)


# direct f16x3
xh, xl = split(x, 64.0)
rs = rowscale(w)
wh, wl = split(w * rs[:, None, None, None], 1.0)
y_direct = x3conv(xh, xl, wh, wl, padding=1) / rs[None, :, None, None].double() / 64.0
# fp32 direct
y_f32 = torch.nn.functional.conv2d(x, w, padding=1).double()

# winograd along y: rows d[-1..H], pairs j: d0=row 2j-1, d1=2j, d2=2j+1, d3=2j+2
xp = torch.nn.functional.pad(x, (0, 0, 1, 1))                   # pad H by 1 each side
d0, d1, d2, d3 = xp[:, :, 0:H:2], xp[:, :, 1:H + 1:2], xp[:, :, 2:H + 2:2], xp[:, :, 3:H + 3:2]
V = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]                        # fp32
g0, g1, g2 = w[:, :, 0], w[:, :, 1], w[:, :, 2]                 # [Cout, Cin, 3(dx)]
U = [g0, ((g0.double() + g1.double() + g2.double()) / 2).float(), ((g0.double() - g1.double() + g2.double()) / 2).float(), g2]
Ucat = torch.stack(U, 0)                                         # [4, Cout, Cin, 3]
rsw = rowscale(Ucat.permute(1, 0, 2, 3))                         # per cout over all f
M = []
for f in range(4):
    vh, vl = split(V[f], 32.0)
    uh, ul = split(U[f] * rsw[:, None, None], 1.0)
    m = x3conv(vh, vl, uh[:, :, None, :], ul[:, :, None, :], padding=(0, 1))   # 1x3 conv over x
    M.append((m / rsw[None, :, None, None].double() / 32.0).float())         # fp32 accumulators scaled
Y0 = (M[0] + M[1]) + M[2]
Y1 = (M[1] - M[2]) - M[3]
y_w = torch.empty(N, Cout, H, W)
y_w[:, :, 0::2] = Y0
y_w[:, :, 1::2] = Y1
for name, y in (("fp32 direct", y_f32), ("f16x3 direct", y_direct), ("f16x3 winograd F(2,3)y", y_w.double())):
    e = (y - ref)
    print(f"{name:26s} max abs {e.abs().max():.3e}  rms {e.pow(2).mean().sqrt():.3e}   (|ref| max {ref.abs().max():.2f} rms {ref.pow(2).mean().sqrt():.3f})")

# ---- F(4,3) along y (points 0, +-1, +-2, inf)
import numpy as np
BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
Hp = H                                       # H multiple of 4
xp4 = torch.nn.functional.pad(x, (0, 0, 1, 1))                  # rows -1 .. H
rows = [xp4[:, :, k:k + H:4] for k in range(6)]                 # d0..d5 for tile j: rows 4j-1 .. 4j+4
# fp32 transform (as the writer would do): V_f = sum_k BT[f][k] d_k
V6 = [sum(float(BT[f, k]) * rows[k] for k in range(6) if BT[f, k] != 0) for f in range(6)]
gk = [w[:, :, k] for k in range(3)]
U6 = [sum(G[f, k] * gk[k].double() for k in range(3)).float() for f in range(6)]
rs6 = rowscale(torch.stack(U6, 0).permute(1, 0, 2, 3))
vmax = max(v.abs().max().item() for v in V6)
vs = 2.0 ** np.floor(np.log2(60000.0 / vmax))
M6 = []
for f in range(6):
    vh, vl = split(V6[f], vs)
    uh, ul = split(U6[f] * rs6[:, None, None], 1.0)
    m = x3conv(vh, vl, uh[:, :, None, :], ul[:, :, None, :], padding=(0, 1))
    M6.append((m / rs6[None, :, None, None].double() / vs).float())
y4 = torch.empty(N, Cout, H, W)
for r in range(4):
    acc = None
    for f in range(6):
        if AT[r, f] != 0:
            t = float(AT[r, f]) * M6[f]
            acc = t if acc is None else acc + t
    y4[:, :, r::4] = acc
e = (y4.double() - ref)
print(f"{'f16x3 winograd F(4,3)y':26s} max abs {e.abs().max():.3e}  rms {e.pow(2).mean().sqrt():.3e}   |V|max {vmax:.1f}")
