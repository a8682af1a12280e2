#!/usr/bin/env python3
"""GPU probe: calc_embd_statistics on PAGEABLE numpy rows [100000 x 512] float16 (the reference's call, fad.py:42-48) -- ms per set and
scores/s of the three-call sequence, for the piece size given by FAD_H2D_PIECE_KB (read once per process: run once per setting;
0 = one copy, then one update)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fadtk_amd

rng = np.random.default_rng(0)
a = rng.standard_normal((100000, 512)).astype(np.float16)
b = (1.05 * rng.standard_normal((100000, 512)) + 0.01).astype(np.float16)
fadtk_amd.calc_embd_statistics(a[:4096])
ms, sets = [], []
for _ in range(8):
    t0 = time.perf_counter(); m1, c1 = fadtk_amd.calc_embd_statistics(a); t1 = time.perf_counter()
    m2, c2 = fadtk_amd.calc_embd_statistics(b); t2 = time.perf_counter()
    f = float(fadtk_amd.calc_frechet_distance(m1, c1, m2, c2)); t3 = time.perf_counter()
    ms.append((t3 - t0) * 1e3); sets += [(t1 - t0) * 1e3, (t2 - t1) * 1e3]
ms, sets = ms[2:], sets[4:]
from fadtk_amd import hip
import torch
with hip.Moments(512) as acc:                       # the pieces of one call: update (waited for), finalize
    acc.set_reference_mean(True)
    tu, tf = [], []
    for _ in range(6):
        acc.reset(); torch.cuda.synchronize()
        t0 = time.perf_counter(); acc.update(a); torch.cuda.synchronize(); t1 = time.perf_counter(); acc.finalize(); t2 = time.perf_counter()
        tu.append((t1 - t0) * 1e3); tf.append((t2 - t1) * 1e3)
print(f"   update + wait {np.median(tu[1:]):.3f} ms, finalize {np.median(tf[1:]):.3f} ms")
ok = np.array_equal(m1, np.mean(a, axis=0))
print(f"FAD_H2D_PIECE_KB={os.environ.get('FAD_H2D_PIECE_KB', '(default)')}: {np.median(sets):.3f} ms per set (min {min(sets):.3f}), "
      f"{np.median(ms):.3f} ms per score = {1e3 / np.median(ms):.0f} scores/s; fad {f:.6f}; mean == np.mean bit for bit: {ok}")
