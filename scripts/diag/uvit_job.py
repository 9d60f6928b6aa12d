#!/usr/bin/env python3
"""Run bench.py's BASELINE configs[2] leg (U-ViT-H/2, batch 64, 50 steps, M=5, W10 + VAE decode) alone, for profiling."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get("DUA_DIAG_LIB"):                     # A/B: an alternative build of the library (scripts/diag/build/...)
    from diffusion_uncertainty_amd import _lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["DUA_DIAG_LIB"])
import bench  # noqa: E402

if os.environ.get("DUA_HG_CONC", "1") == "0":          # A/B: tile widths by whole rounds also while the chunks run on streams of their own
    import torch
    torch.zeros(1, device="cuda")                      # (torch's HIP runtime first: the library must attach to it, not bring its own)
    from diffusion_uncertainty_amd import _lib as _l2
    _l2.lib().dua_set_hgemm_concurrent = lambda on: 0

ns = argparse.Namespace(uvit_mc_group=int(sys.argv[1]) if len(sys.argv) > 1 else 1, uvit_split_k=(len(sys.argv) > 2 and sys.argv[2] == "1"))
print(json.dumps(bench.other_config_uvit("cuda", ns)))
