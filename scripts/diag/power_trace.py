#!/usr/bin/env python3
"""Socket power and clocks while the headline job runs (read-only: rocm-smi --showpower --showmaxpower --showclocks; no setting is
touched).  DESIGN.md calls the conv loop power-limited on the strength of in-kernel clocks of 1.46 - 1.66 GHz (PMC); this puts the
package power and the power cap beside it.

    python scripts/diag/power_trace.py [--seconds 40] > gpurun_out/r06_power_trace.txt

Starts `bench.py --steps 10 --warmup 1 --no-secondary --no-cpu-baseline --no-check --no-profile` as a child process, samples rocm-smi about
three times a second from model load to the end of the timed job, prints every sample (time, power W, sclk MHz, mclk MHz) and a summary."""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def smi(*flags):
    try:
        out = subprocess.run(["rocm-smi", *flags, "--json"], capture_output=True, text=True, timeout=10).stdout
        return json.loads(out[out.index("{"):])
    except Exception as e:
        return {"error": str(e)}


def first_number(d, *needles):
    for card in d.values():
        if not isinstance(card, dict):
            continue
        for k, v in card.items():
            if all(n in k.lower() for n in needles):
                m = re.search(r"[-+]?\d+(\.\d+)?", str(v))
                if m:
                    return float(m.group(0))
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    a = ap.parse_args()
    cap = smi("--showmaxpower")
    print("max power:", json.dumps(cap))
    idle = smi("--showpower", "--showclocks")
    print("idle:", json.dumps(idle)[:600])
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "1", "--no-secondary", "--no-cpu-baseline",
                              "--no-check", "--no-profile"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    t0 = time.time()
    rows = []
    while child.poll() is None and time.time() - t0 < a.seconds + 120:
        d = smi("--showpower", "--showclocks")
        p, sclk, mclk = first_number(d, "power"), first_number(d, "sclk"), first_number(d, "mclk")
        rows.append((time.time() - t0, p, sclk, mclk))
        print(f"t={rows[-1][0]:7.2f} s  power {p} W  sclk {sclk} MHz  mclk {mclk} MHz", flush=True)
        time.sleep(0.25)
    out = child.communicate()[0]
    line = [l for l in out.splitlines() if l.startswith("{")]
    print("bench:", line[-1][:400] if line else out[-400:])
    busy = [r for r in rows if r[1] is not None]
    if busy:
        top = sorted(r[1] for r in busy)
        hot = [r for r in busy if r[1] >= 0.8 * top[-1]]
        print(f"samples {len(busy)}; power max {top[-1]:.0f} W, median of the samples within 20 % of it {sorted(r[1] for r in hot)[len(hot) // 2]:.0f} W "
              f"({len(hot)} samples), their median sclk {sorted((r[2] or 0) for r in hot)[len(hot) // 2]:.0f} MHz; power cap {first_number(cap, 'max', 'power')} W")


if __name__ == "__main__":
    main()
