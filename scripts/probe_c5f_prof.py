#!/usr/bin/env python3
"""GPU probe, run under rocprofv3 --kernel-trace --stats: 32 songs of [1500 x 768] frames through the batched per-song entry
(the D x D route: per-song covariance + batched float64 Newton-Schulz against the shared baseline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from fadtk_amd import hip
r = bench.extra_c5_frames(torch, hip, torch.device("cuda", 0))
print({k: v for k, v in r.items() if k not in ("cpu_baseline", "note")})
