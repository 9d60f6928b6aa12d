#!/usr/bin/env python3
"""bench.py on an alternative build of the library (A/B of whole jobs):  python scripts/diag/bench_with_lib.py <lib.so> [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diffusion_uncertainty_amd import _lib as _l  # noqa: E402

_l.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
