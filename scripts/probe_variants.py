#!/usr/bin/env python3
"""GPU probe: the fp16 tile kernel on the HBM-bound D=128 stream at several lengths.  (Until round 2 the library also shipped a
one-tile-per-wave kernel, pinned with FAD_MOMENTS_VARIANT=8; the comparison that retired it -- 16.8M x 128: 760 vs 784 us, 9.2M: 441 vs
465, 1M: 48.6 vs 55.7 -- was made with this script at commit `moments: the one-tile-per-wave kernel leaves the library`.)"""
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
    print(f"rows={rows:9d}: tile kernel {k*1e3:7.1f} us = {rows*256/k/1e9:6.2f} TB/s  (+ reduce {r*1e3:5.1f} us)  kernel variant {v}", flush=True)
    m.close(); del x
