#!/usr/bin/env python3
"""GPU probe, run under rocprofv3 --kernel-trace --stats: per-song FAD of many-frame songs (the D x D route) at the Encodec shape
(songs of [2250 x 128]) and a CLAP-like one ([600 x 512])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fadtk_amd import hip

rng = np.random.default_rng(0)
for d, frames, songs in ((128, 2250, 2000), (512, 600, 256)):
    base = rng.standard_normal((20 * d, d)) * (0.5 + rng.random(d))
    mu_b = base.mean(0); cov_b = np.cov(base, rowvar=False)
    rows = (torch.randn((songs * frames, d), device="cuda") * torch.from_numpy(0.5 + rng.random(d)).cuda().float()).to(torch.float16)
    offs = np.arange(0, songs * frames + 1, frames, dtype=np.int64)
    hip.frechet_batched(mu_b, cov_b, rows, offs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sc, st = hip.frechet_batched(mu_b, cov_b, rows, offs)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"{songs} songs of [{frames} x {d}]: {t*1e3:.2f} ms = {songs/t:.0f} songs/s; ok {(st == 0).sum()}", flush=True)
    del rows
