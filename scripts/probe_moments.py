#!/usr/bin/env python3
"""GPU probe: tile-kernel time of fad_moments_update / fad_moments_update_multi over (N, D) shapes -> TFLOP/s, GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fadtk_amd import hip

shapes = [(100_000, 512), (1_000_000, 128), (16_000_000, 128), (100_000, 768), (100_000, 1024), (1_000_000, 512), (10_000, 128), (2_000, 768)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = [(100_000, 512), (100_000, 768)]
for n, d in shapes:
    x = torch.randn((n, d), device="cuda", dtype=torch.float16)
    m = hip.Moments(d)
    for _ in range(3):
        m.update(x)
    m.set_timing(True)
    for _ in range(10):
        m.update(x)
    k, r, v = m.last_timing()
    print(f"single N={n:9d} D={d:5d} kernel={k*1e3:9.1f} us reduce={r*1e3:7.1f} us  {2*n*d*d/k/1e9:8.1f} TFLOP/s  {n*d*2/k/1e6:8.1f} GB/s  variant={v}", flush=True)
    m.close(); del x
# several sets per launch (the two datasets of a score; 4 / 8 = what score_inf-style batches would use)
for n, d in [(100_000, 512), (100_000, 768), (100_000, 1024), (100_000, 128)]:
    for sets in (2, 4, 8):
        xs = [torch.randn((n, d), device="cuda", dtype=torch.float16) for _ in range(sets)]
        ms = [hip.Moments(d) for _ in range(sets)]
        for _ in range(3):
            hip.Moments.update_multi(ms, xs)
        ms[0].set_timing(True)
        for _ in range(10):
            hip.Moments.update_multi(ms, xs)
        k, r, v = ms[0].last_timing()
        print(f"multi x{sets} N={n:9d} D={d:5d} kernel={k*1e3:9.1f} us ({k*1e3/sets:7.1f} per set) reduce={r*1e3:7.1f} us  "
              f"{sets*2*n*d*d/k/1e9:8.1f} TFLOP/s  {sets*n*d*2/k/1e6:8.1f} GB/s", flush=True)
        for m in ms:
            m.close()
        del xs
