#!/usr/bin/env python3
"""HBM-side traffic of the Winograd-form conv3x3 launches of one ADM-128 forward, from tile counts.

bench.py's `roofline.traffic` is the fabric-side byte count per conv launch measured with rocprofv3 --pmc (FETCH_SIZE doubled per the
gfx950 correction, + WRITE_SIZE; Infinity-Cache hits are counted: MI355X_MICROARCH.md, HBM section) - 1.57 x the algorithmic bytes in
round 5.  This script enumerates the 70 ResBlock convolutions of the forward at a given batch (channel plan of init_model.ADM128_PARAMS,
form per shape by the library's own rule), and prices every launch three ways:

  algorithmic   V images once + output (+ residual) + packed weights once                         
  per-tile      what the persistent grid REQUESTS through its LDS-DMA: every (row tile, column tile) pair reads its V rows and its
                weight panel, i.e. V x ntn (column tiles of 128 channels) and weights x ntm (row tiles of 128 positions)
  l2-shared     the per-tile requests with the weight panel shared by the workgroups of an XCD (the tile list is chunked per XCD and
                ordered column-tile-major inside a chunk: 8 fabric-side reads of each panel per launch) and V read once per column tile

and compares the launch-weighted means per kernel symbol with a measured profiles/r*_pmc_traffic_conv.json.  No GPU needed:
    python scripts/conv_traffic_model.py [--batch 128] [--pmc profiles/r05z_pmc_traffic_conv.json] [--json out.json]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def adm128_convs():
    """(H, Cin, Cout, residual mode) of the 70 ResBlock convs of ADM-128 in execution order (unet_openai.py:481-617 with
    init_model.ADM128_PARAMS): conv1 of a block has no residual, conv2 adds x (mode 0) or x through the block's 2x resampling
    (+1 up: the source has a quarter of the pixels; -1 down: four times as many)."""
    from diffusion_uncertainty_amd.init_model import ADM128_PARAMS as P
    mc, mult, nrb = P["model_channels"], P["channel_mult"], P["num_res_blocks"]
    out, ch, H = [], mc, P["image_size"]
    chans = [ch]

    def res(cin, cout, H_in, updown):
        Hc = H_in * 2 if updown > 0 else (H_in // 2 if updown < 0 else H_in)
        out.append((Hc, cin, cout, None))
        out.append((Hc, cout, cout, updown))
        return Hc

    for level, m in enumerate(mult):
        for _ in range(nrb):
            res(ch, m * mc, H, 0)
            ch = m * mc
            chans.append(ch)
        if level != len(mult) - 1:
            H = res(ch, ch, H, -1)
            chans.append(ch)
    res(ch, ch, H, 0)
    res(ch, ch, H, 0)
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            res(ch + chans.pop(), m * mc, H, 0)
            ch = m * mc
            if level and i == nrb:
                H = res(ch, ch, H, +1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--pmc", default=None)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    from diffusion_uncertainty_amd import _lib
    L = _lib.lib()                                                      # host-side shape rules only: no GPU is touched
    N = a.batch
    rows, per = [], {}
    for H, cin, cout, rmode in adm128_convs():
        form = L.dua_conv3x3_f16x3_wino_pick(N, H, H, cin, cout)
        assert form in (2, 4), (H, cin, cout)
        px = N * H * H
        v = (6 if form == 4 else 8) * px * cin
        rows_per_tile = 4 if form == 4 else 2
        ntm = (N * (H // rows_per_tile) * H + 127) // 128
        ntn = (cout + 127) // 128
        w = 4 * ntn * 128 * 18 * cin if form == 4 else 4 * ntn * 256 * 6 * cin
        o = 4 * px * cout
        r = 0 if rmode is None else (o if rmode == 0 else (o // 4 if rmode > 0 else 4 * o))
        sym = ("k_wconv4<%d>" % (0 if rmode is None else (1 if rmode == 0 else (2 if rmode > 0 else 3)))) if form == 4 else "k_wconv"
        d = dict(symbol=sym, H=H, Cin=cin, Cout=cout, form=form, ntm=ntm, ntn=ntn, tiles=ntm * ntn, rounds=round(ntm * ntn / 256, 2),
                 V=v, W=w, out=o, residual=r, algorithmic=v + w + o + r, per_tile=v * ntn + (w // ntn) * ntm + o + r,
                 l2_shared=v * ntn + 8 * w + o + r, flops=2 * px * 9 * cin * cout)
        rows.append(d)
        s = per.setdefault(sym, dict(launches=0, algorithmic=0, per_tile=0, l2_shared=0, V=0, W=0, out=0, residual=0, flops=0, v_times_ntn=0))
        s["launches"] += 1
        for k in ("algorithmic", "per_tile", "l2_shared", "V", "W", "out", "residual", "flops"):
            s[k] += d[k]
        s["v_times_ntn"] += v * ntn
    tot = dict(launches=len(rows), **{k: sum(d[k] for d in rows) for k in ("algorithmic", "per_tile", "l2_shared", "V", "W", "out", "residual", "flops")})
    tot["v_times_ntn"] = sum(d["V"] * d["ntn"] for d in rows)
    per["all"] = tot
    meas = {}
    if a.pmc:
        k = json.load(open(a.pmc))["kernels"]
        for name, v in k.items():
            if name.startswith("symbol:"):
                meas[name[7:]] = dict(fetch=2048.0 * v["FETCH_SIZE_KB_per_launch"], write=1024.0 * v["WRITE_SIZE_KB_per_launch"], launches=v["launches"])
    print(f"ADM-128 forward at batch {N}: {len(rows)} Winograd-form conv3x3 launches; per-launch means in GB")
    print(f"{'symbol':14s} {'n':>3s} {'algorithmic':>11s} {'V':>6s} {'V x ntn':>8s} {'W':>6s} {'8 W':>6s} {'out':>6s} {'resid':>6s} {'l2-shared':>9s} | {'measured fetch':>14s} {'write':>6s} {'total':>6s}  fetch explained by")
    summary = {}
    for sym, s in sorted(per.items()):
        n = s["launches"]
        g = lambda x: x / n / 1e9
        line = f"{sym:14s} {n:3d} {g(s['algorithmic']):11.3f} {g(s['V']):6.3f} {g(s['v_times_ntn']):8.3f} {g(s['W']):6.3f} {g(8 * s['W']):6.3f} {g(s['out']):6.3f} {g(s['residual']):6.3f} {g(s['l2_shared']):9.3f}"
        rec = dict(launches_per_forward=n, algorithmic_GB=round(g(s["algorithmic"]), 4), V_GB=round(g(s["V"]), 4), V_times_ntn_GB=round(g(s["v_times_ntn"]), 4),
                   weights_GB=round(g(s["W"]), 4), out_GB=round(g(s["out"]), 4), residual_GB=round(g(s["residual"]), 4), l2_shared_model_GB=round(g(s["l2_shared"]), 4))
        m = meas.get(sym)
        if m:
            f, w_ = m["fetch"] / 1e9, m["write"] / 1e9
            act_reads = g(s["v_times_ntn"]) + g(s["residual"])
            rec.update(measured_fetch_GB=round(f, 4), measured_write_GB=round(w_, 4), measured_total_GB=round(f + w_, 4),
                       measured_over_algorithmic=round((f + w_) / g(s["algorithmic"]), 3), measured_over_l2_shared_model=round((f + w_) / g(s["l2_shared"]), 3),
                       fetch_attributed_to_activations_GB=round(min(f, act_reads), 4), fetch_attributed_to_weights_GB=round(max(0.0, f - act_reads), 4),
                       weight_panel_fabric_reads_per_launch=round(max(0.0, f - act_reads) / max(g(s["W"]), 1e-12), 1))
            line += f" | {f:14.3f} {w_:6.3f} {f + w_:6.3f}  V x ntn + residual {act_reads:.3f} + weights {max(0.0, f - act_reads):.3f} (= {max(0.0, f - act_reads) / max(g(s['W']), 1e-12):.1f} x W)"
        summary[sym] = rec
        print(line)
    if a.json:
        json.dump(dict(batch=N, model="ADM-128 ResBlock conv3x3 launches of one forward, Winograd forms by dua_conv3x3_f16x3_wino_pick", pmc=a.pmc,
                       note="GB per launch (means over the launches of a symbol in one forward); V x ntn: every column tile of 128 output channels reads its "
                            "rows of the V images through the fabric; fetch attributed to weights = measured fetch - (V x ntn + residual), expressed as "
                            "fabric-side reads of the packed weights per launch", per_symbol=summary, layers=rows), open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
