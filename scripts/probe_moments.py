#!/usr/bin/env python3
"""GPU probe: tile-kernel time of fad_moments_update over (N, D) shapes -> TFLOP/s and GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fadtk_amd import hip
shapes = [(100_000, 512), (1_000_000, 128), (16_000_000, 128), (100_000, 768), (100_000, 1024), (1_000_000, 512), (10_000, 128), (2_000, 768)]
for n, d in shapes:
    x = torch.randn((n, d), device="cuda", dtype=torch.float16)
    m = hip.Moments(d)
    for _ in range(3):
        m.update(x)
    m.set_timing(True)
    for _ in range(10):
        m.update(x)
    k, r, v = m.last_timing()
    print(f"N={n:9d} D={d:5d} kernel={k*1e3:9.1f} us reduce={r*1e3:7.1f} us  {2*n*d*d/k/1e9:8.1f} TFLOP/s  {n*d*2/k/1e6:8.1f} GB/s  variant={v}")
    m.close(); del x
