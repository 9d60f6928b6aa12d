#!/usr/bin/env python3
"""BASELINE configs[2] (U-ViT-H/2 + VAE decode) alone, as bench.py runs it: driver for rocprofv3 --kernel-trace --stats."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
print(bench.other_config_uvit(torch.device("cuda:0"), argparse.Namespace()))
