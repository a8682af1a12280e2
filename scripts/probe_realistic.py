#!/usr/bin/env python3
"""GPU probe, run under rocprofv3 --kernel-trace --stats: what bench.py's `value_realistic` loop launches -- PAIRS (16) realistic scores per batch
(k^-1 spectra, frames with an offset: bench.make_realistic_sets), two update_multi calls of 8 frame matrices each with numpy's running
sums on (detached walk), then ONE chain for the 8 pairs (fad_frechet_from_moments_multi_begin: scaled steps on the 128 x 128-tile
kernels, the exact correction, the verification products) -- a dozen batches, so that the kernel statistics show what a batch costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from fadtk_amd import hip, _capi as K

dev = torch.device("cuda", 0)
# EXTRA_STREAMS=N: N further torch streams, each used once and then left idle (bench.py's side blocks leave such streams behind: does their
# mere existence slow this loop?  r05n)
extra_streams = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("EXTRA_STREAMS", "0")))]
for st_ in extra_streams:
    with torch.cuda.stream(st_):
        torch.zeros(16, device=dev).add_(1)
torch.cuda.synchronize()
pairs = [bench.make_realistic_sets(torch, dev, k) for k in range(4)]
PAIRS = int(os.environ.get("PAIRS", "16"))          # pairs per batched chain (bench.py --batch: 16 since r05v; 8 before)
hs = [(hip.Moments(bench.DIM), hip.Moments(bench.DIM)) for _ in range(PAIRS)]
mode = sys.argv[1] if len(sys.argv) > 1 else "detached"
for a, b in hs:
    a.set_reference_mean(mode != "off", detached=(mode == "detached")); b.set_reference_mean(mode != "off", detached=(mode == "detached"))

NB = 3 if (len(sys.argv) > 2 and sys.argv[2] == "pipelined") else 1      # batches in flight (bench.py keeps three)
groups = [hs] + [[(hip.Moments(bench.DIM), hip.Moments(bench.DIM)) for _ in range(PAIRS)] for _ in range(NB - 1)]
for grp in groups[1:]:
    for a, b in grp:
        a.set_reference_mean(mode != "off", detached=(mode == "detached")); b.set_reference_mean(mode != "off", detached=(mode == "detached"))
jobs = [None] * NB

def feed(q):
    for g in range(PAIRS // 4):
        grp = groups[q][4 * g:4 * g + 4]
        for a, b in grp:
            a.reset(); b.reset()
        hip.Moments.update_multi([h for ab in grp for h in ab], [x for k in range(4) for x in pairs[k]])
    jobs[q] = hip.FrechetMultiJob(groups[q], mean_dtype=K.FAD_F16)

def run(n):
    res = None
    for i in range(n):
        q = i % NB
        if jobs[q] is not None:
            res = jobs[q].result(); jobs[q] = None
        feed(q)
    for q in range(NB):
        if jobs[q] is not None:
            res = jobs[q].result(); jobs[q] = None
    return res

run(6)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 12
res = run(n)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"extra streams {len(extra_streams)}: mode {mode}, {NB} batch(es) in flight: {dt * 1e3:.3f} ms per batch of {PAIRS} scores = {PAIRS / dt:.0f} scores/s; last batch: routes {[d['route'] for _, d in res]} iterations {[d['iters'] for _, d in res]}")
