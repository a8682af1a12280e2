#!/usr/bin/env python3
"""GPU probe, run under rocprofv3 --kernel-trace --stats: the config-5 shape (10 000 two-frame songs, D = 768) through the batched
per-song entry point, so that the kernel statistics show where the call's time goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fadtk_amd import hip

rng = np.random.default_rng(0)
d, songs = 768, 10000
base = rng.standard_normal((4000, d)).astype(np.float32)
mu_b = base.mean(0).astype(np.float64); cov_b = np.cov(base.astype(np.float64), rowvar=False)
rows = torch.randn((2 * songs, d), device="cuda").to(torch.float16)
offs = np.arange(0, 2 * songs + 1, 2, dtype=np.int64)
for _ in range(3): hip.frechet_batched(mu_b, cov_b, rows, offs)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): scores, status = hip.frechet_batched(mu_b, cov_b, rows, offs)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
print(f"{songs} two-frame songs, D={d}: {t*1e3:.3f} ms per call = {songs/t/1e6:.2f} M songs/s; ok {(status == 0).sum()}")
