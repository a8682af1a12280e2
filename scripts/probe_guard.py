#!/usr/bin/env python3
"""GPU probe: what the shift guard costs when it fires (columns with |mean| >> std, as transformer hidden states have)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fadtk_amd import hip

def timed(fn, n=5):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

for d, n in ((512, 100000), (768, 100000), (128, 1000000)):
    x = torch.randn((n, d), device="cuda")
    for outliers in (0, 4, 32):
        y = x.clone()
        if outliers: y[:, :outliers] = 30.0 + 0.1 * y[:, :outliers]
        y = y.to(torch.float16)
        with hip.Moments(d) as m:
            t = timed(lambda: m.update(y))
            m.set_timing(True); m.update(y); k, r, v = m.last_timing()
        print(f"D={d} N={n} outlier columns {outliers:2d}: update {t:8.1f} us  (kernel family {v}: 0 = fp16 MFMA, second pass over x - c when the guard fires)", flush=True)
