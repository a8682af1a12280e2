#!/usr/bin/env python3
"""GPU probe: the two fp16 tile kernels (FAD_MOMENTS_VARIANT=4: workgroup tile, 8: wave tile) on the HBM-bound D=128 stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fadtk_amd import hip

for rows in (1_000_000, 4_194_304, 9_216_000, 16_777_216):
    x = torch.randn((rows, 128), device="cuda", dtype=torch.float16)
    m = hip.Moments(128)
    m.set_timing(True)
    for _ in range(3): m.update(x)
    ks = []
    for _ in range(5):
        m.update(x); k, r, v = m.last_timing(); ks.append(k)
    k = sorted(ks)[2]
    print(f"VARIANT={os.environ.get('FAD_MOMENTS_VARIANT','auto')} rows={rows:9d}: tile kernel {k*1e3:7.1f} us = {rows*256/k/1e9:6.2f} TB/s  (+ reduce {r*1e3:5.1f} us)  kernel variant {v}", flush=True)
    m.close(); del x
