#!/usr/bin/env python3
"""the HBM-bound V writers (16.6 % of the headline job) beside the MFMA-bound convs (62 %), by running the two halves of
the batch on two HIP streams that own DISJOINT sets of compute units (hipExtStreamCreateWithCUMask) - de-phased halves put one half's writer
beside the other half's conv.  Same-box alternating A/B of the full ADM-128 forward at batch 128, f16x3:

    A  one stream, all 256 CUs (the product)
    B  two plain streams (UNetModel.stream_split = 2: both halves compete for every CU)
    C  two CU-masked streams, 4 + 4 XCDs (128 CUs each)
    D  two CU-masked streams, 6 + 2 XCDs by XCD count (uneven halves of the CHIP, equal halves of the batch)
    E  two CU-masked streams, each with all CUs of every XCD but interleaved halves of each XCD's CUs (16 + 16 CUs per XCD)

CU-mask bit i of a multi-XCD part addresses XCD (i mod 8), CU (i div 8) of that XCD (amdkfd's symmetric mapping for gfx9.4.3+).
    python scripts/diag/cumask_streams.py [--batch 128] [--reps 6]"""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def masked_stream(hip, bits):
    words = (ctypes.c_uint32 * 8)()
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask: error {rc}")
    return torch.cuda.ExternalStream(h.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    model, _, _ = bench.build_model(dev)
    model.precision = "f16x3"
    g = torch.Generator().manual_seed(49395)
    X = torch.randn(a.batch, 3, 128, 128, generator=g).to(dev)
    y = torch.randint(0, 1000, (a.batch,), generator=g).to(dev)
    t = torch.full((a.batch,), 500, device=dev, dtype=torch.long)
    allb = range(256)
    variants = {
        "A one stream, 256 CUs": None,
        "B two plain streams": "plain",
        "C masks 4 + 4 XCDs": ([i for i in allb if i % 8 < 4], [i for i in allb if i % 8 >= 4]),
        "D masks 6 + 2 XCDs": ([i for i in allb if i % 8 < 6], [i for i in allb if i % 8 >= 6]),
        "E masks 16 + 16 CUs of every XCD": ([i for i in allb if (i // 8) % 2 == 0], [i for i in allb if (i // 8) % 2 == 1]),
    }
    streams = {}
    for name, v in variants.items():
        if v is None:
            streams[name] = None
        elif v == "plain":
            streams[name] = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        else:
            streams[name] = [masked_stream(hip, v[0]), masked_stream(hip, v[1])]

    def run(name):
        st = streams[name]
        model.stream_split = 1 if st is None else 2
        if st is not None:
            model._streams = st
        return model(X, t, y=y, out_channels_keep=3)

    ref = run("A one stream, 256 CUs")
    for name in variants:                                   # warm-up + bits
        o = run(name)
        torch.cuda.synchronize()
        print(f"{name:36s} bit-identical to A: {torch.equal(o, ref)}", flush=True)
    times = {n: [] for n in variants}
    for r in range(a.reps):                                 # alternating: every variant once per round
        for name in variants:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                run(name)
            torch.cuda.synchronize()
            times[name].append((time.perf_counter() - t0) / 3)
    base = min(times["A one stream, 256 CUs"])
    for name, v in times.items():
        print(f"{name:36s} forward at batch {a.batch}: best {min(v) * 1e3:8.2f} ms  median {sorted(v)[len(v) // 2] * 1e3:8.2f} ms  ({base / min(v):.3f} x A)", flush=True)


if __name__ == "__main__":
    main()
