#!/usr/bin/env python3
"""GPU probe: FAD scores/s with 1, 2, 3 scores in flight (one host thread + HIP stream + handle pair each)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fadtk_amd import hip, dist as fdist

N, D = 100_000, 512
g = torch.Generator(device="cuda"); g.manual_seed(10)
a = torch.randn((N, D), generator=g, device="cuda").to(torch.float16)
b = (1.02 * torch.randn((N, D), generator=g, device="cuda") + 0.01).to(torch.float16)

def worker(k, steps, out):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        sh = fdist.SharedStats(D, 2, 0)
        ma, mb = sh.moments
        for i in range(steps):
            ma.reset(); mb.reset()
            hip.Moments.update_multi([ma, mb], [a, b])
            fad, diag = hip.frechet_from_moments(ma, mb, mean_dtype=0)
        out[k] = fad
        s.synchronize()

for nthreads in (1, 2, 3, 4):
    out = {}
    for rep in range(2):                                    # first rep = warm-up (allocations, adaptive iteration count)
        steps = 20 if rep == 0 else 200
        ts = [threading.Thread(target=worker, args=(k, steps, out)) for k in range(nthreads)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"in flight {nthreads}: {nthreads * steps / dt:8.1f} scores/s  ({dt / (nthreads * steps) * 1e6:6.1f} us per score)  fad={out[0]:.9f}", flush=True)
