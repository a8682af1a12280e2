#!/usr/bin/env python3
"""GPU probe: per-song FAD throughput (config 5 shape: D=768 baseline, 10 000 two-frame songs; and multi-frame songs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
import recipes as R
from fadtk_amd import hip
for d, n_songs, rows in ((768, 10000, 2), (128, 2000, 10), (512, 200, 30)):
    mu_b, cov_b = R.baseline_stats(95, 3 * d, d)
    x = (np.random.default_rng(1).standard_normal((n_songs * rows, d)) * 0.9).astype(np.float16)
    off = np.arange(0, n_songs * rows + 1, rows)
    xd = torch.from_numpy(x).cuda()
    hip.frechet_batched(mu_b, cov_b, xd, off)          # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        s, st = hip.frechet_batched(mu_b, cov_b, xd, off)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"D={d} songs={n_songs} frames/song={rows}: {dt*1e3:9.2f} ms  -> {n_songs/dt:12.0f} songs/s  ok={int((st==0).sum())}")
