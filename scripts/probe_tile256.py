#!/usr/bin/env python3
"""256-column-slab moments kernel against the 128 x 128 kernel on the same frames (GPU box):
tile-kernel and reduce durations by HIP events (fad_moments_last_timing), inputs rotated so that every launch streams from HBM,
plus the difference of the two packed results.    python scripts/probe_tile256.py [quick]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fadtk_amd import hip  # noqa: E402

dev = torch.device("cuda", 0)
shapes = [(100_000, 512, 2), (100_000, 768, 2), (100_000, 1024, 2), (100_000, 512, 1), (400_000, 512, 2), (100_000, 512, 4), (50_000, 1280, 2)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = shapes[:3]
for n, d, sets in shapes:
    npairs = max(1, int(700e6 // (n * d * 2 * sets)) + 1)           # > 256 MiB Infinity Cache in rotation
    g = torch.Generator(device=dev); g.manual_seed(n + d)
    data = [[(torch.randn((n, d), generator=g, device=dev) * (1 + 0.1 * k) + 0.01 * k).to(torch.float16) for k in range(sets)] for _ in range(npairs)]
    res = {}
    for knob in ("0", "1"):
        os.environ["FAD_MOMENTS_TILE256"] = knob
        accs = [hip.Moments(d) for _ in range(sets)]
        for a in accs:
            a.reset()
        hip.Moments.update_multi(accs, data[0])
        torch.cuda.synchronize()
        accs[0].set_timing(1)
        reps = 12
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for r in range(reps):
            for a in accs:
                a.reset()
            hip.Moments.update_multi(accs, data[r % npairs])
        t1.record(); torch.cuda.synchronize()
        k_ms, r_ms, variant = accs[0].last_timing()
        accs[0].set_timing(0)
        for a in accs:
            a.reset()
        hip.Moments.update_multi(accs, data[0])
        packed = [a.export() for a in accs]
        res[knob] = (k_ms, r_ms, variant, t0.elapsed_time(t1) / reps, packed)
        for a in accs:
            a.close()
    flops = sets * 2.0 * n * d * d
    for knob in ("0", "1"):
        k_ms, r_ms, variant, tot, _ = res[knob]
        print(f"[{n} x {d}] x {sets}  variant {variant}: tile {k_ms * 1e3:7.1f} us ({flops / k_ms / 1e9 / 2500 * 100:5.1f} % of 2.5 PF algorithmic)  "
              f"guard+reduce {r_ms * 1e3:6.1f} us   update {tot * 1e3:7.1f} us", flush=True)
    dmax = max(np.abs(a - b).max() / np.abs(b[1 + d:]).max() for a, b in zip(res["1"][4], res["0"][4]))
    print(f"     max |new - old| / max|M| = {dmax:.2e}   speed-up of the update {res['0'][3] / res['1'][3]:.2f}x", flush=True)
    del data
    torch.cuda.empty_cache()
