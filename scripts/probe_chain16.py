#!/usr/bin/env python3
"""The batched square-root chain alone at the bench's batch: B config-3-shaped pairs (D = 512), moments once, then
fad_frechet_from_moments_multi_begin / _multi_end repeatedly on one stream.  `--lib path` loads a variant of the library
(scripts/build_variant.sh).  Run under `rocprofv3 --kernel-trace --stats` for per-kernel durations."""
import argparse, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ap = argparse.ArgumentParser(); ap.add_argument("--lib", default=None); ap.add_argument("--pairs", type=int, default=16); ap.add_argument("--reps", type=int, default=30)
args = ap.parse_args()
import torch
from fadtk_amd import _capi as K
if args.lib:
    K._lib = K.load_library(args.lib)
from fadtk_amd import hip
dev = torch.device("cuda", 0)
d, n, B = 512, 100000, args.pairs
g = torch.Generator(device=dev); g.manual_seed(3)
pairs = []
a = torch.randn((n, d), generator=g, device=dev).to(torch.float16)
for k in range(B):
    b = ((1.0 + 0.01 * (k + 2)) * torch.randn((n, d), generator=g, device=dev) + 0.01).to(torch.float16)
    ma, mb = hip.Moments(d), hip.Moments(d)
    hip.Moments.update_multi([ma, mb], [a, b])
    pairs.append((ma, mb))
    del b
for _ in range(4):
    res = hip.FrechetMultiJob(pairs, mean_dtype=K.FAD_F16).result()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(args.reps):
    vals, diags = hip.FrechetMultiJob(pairs, mean_dtype=K.FAD_F16).result_arrays()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.reps
print(f"{args.lib or 'default'}: chain of {B} pairs: {dt * 1e6:.1f} us per call ({dt / B * 1e6:.1f} us per score); fad[0] {vals[0]:.9f} fad[-1] {vals[-1]:.9f} iters {diags[0].as_dict()['iters']} route {diags[0].as_dict()['route']}")
