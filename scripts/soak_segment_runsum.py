#!/usr/bin/env python3
"""GPU soak: numpy's per-file / per-song float32 running column sums (fad_moments_update_segmented_ref) bit for bit over random file sets -- 1 .. 300
files of 0 .. 6000 rows (empty and one-row files included; files above and below the 256 rows at which the job-table walk takes over), D 8 .. 768,
float16 / float32 frames, device and host rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fadtk_amd import hip

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
bad = 0
for case in range(cases):
    d = int(rng.choice([8, 24, 96, 128, 136, 256, 384, 512, 768]))
    nf = int(rng.choice([1, 2, 7, 40, 300]))
    kind = rng.choice(["short", "long", "mixed"])
    hi = {"short": 200, "long": 6000, "mixed": 1500}[kind]
    sizes = rng.integers(0 if kind != "long" else 257, hi + 1, size=nf)
    if rng.random() < 0.3: sizes[rng.integers(0, nf)] = 0
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(off[-1])
    if n == 0: continue
    dt = np.float16 if rng.random() < 0.8 else np.float32
    x = (rng.standard_normal((n, d)) * (0.2 + rng.random()) + float(rng.choice([0.0, 0.5, 4.0]))).astype(dt)
    host = rng.random() < 0.3
    with hip.Moments(d) as m:
        src = x if host else torch.from_numpy(x).cuda()
        sums, runs = m.update_segmented(src, off, want_runsums=True)
    runs = runs.cpu().numpy() if hasattr(runs, "cpu") else runs
    ok = True
    for f in range(nf):
        xs = x[off[f]:off[f + 1]]
        if xs.shape[0] == 0: continue
        # numpy's own order: np.mean over axis 0 adds the rows one after the other in float32 (pinned by the golden fixtures)
        want_mean = np.mean(xs, axis=0)                      # dtype of the file
        got_mean = (runs[f].astype(np.float64) / xs.shape[0]).astype(np.float32).astype(dt)
        if not np.array_equal(got_mean, want_mean):
            ok = False; break
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: files={nf} kind={kind} d={d} dtype={np.dtype(dt).name} host={host} first bad file {f} rows {xs.shape[0]}")
print(f"seed {seed}: {cases} cases, {bad} mismatches")
