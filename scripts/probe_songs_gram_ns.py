#!/usr/bin/env python3
"""GPU probe: songs of 65..D frames (10-second clips of a 50 frames/s, D = 768 model: 499 frames) through the batched per-song
entry point.  Run twice: as is (Gram matrix + Newton-Schulz on n_pad x n_pad) and with FAD_SONG_GRAM=0 (the D x D product)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fadtk_amd import hip

rng = np.random.default_rng(0)
for d, n, songs in ((768, 499, 64), (512, 150, 256), (768, 100, 256)):
    base = rng.standard_normal((4000, d)).astype(np.float32) * (0.5 + rng.random(d))
    mu_b = base.mean(0).astype(np.float64); cov_b = np.cov(base.astype(np.float64), rowvar=False)
    rows = (torch.randn((n * songs, d), device="cuda") * 0.9 + 0.05).to(torch.float16)
    offs = np.arange(0, n * songs + 1, n, dtype=np.int64)
    mu_d, cov_d = torch.from_numpy(mu_b).cuda(), torch.from_numpy(cov_b).cuda()
    for _ in range(2): hip.frechet_batched(mu_d, cov_d, rows, offs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): scores, status = hip.frechet_batched(mu_d, cov_d, rows, offs)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
    from oracle import fad_oracle as O
    host = rows[:2 * n].cpu().numpy()
    want = O.individual_scores(mu_b, cov_b, [host[:n], host[n:]], run_sqrtm=False)
    print(f"   first two songs: {scores[:2]} oracle {np.array(want)} rel {np.abs(scores[:2] - want) / np.abs(want)}")
    print(f"FAD_SONG_GRAM={os.environ.get('FAD_SONG_GRAM', '1')}: {songs} songs of [{n} x {d}]: {t*1e3:.2f} ms per call = {songs/t:.0f} songs/s; "
          f"ok {(status == 0).sum()}, mean score {np.nanmean(scores):.9f}")
