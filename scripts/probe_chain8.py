#!/usr/bin/env python3
"""The batched square-root chain alone: eight config-3-shaped pairs (D = 512), moments once, then fad_frechet_from_moments_multi_begin /
_multi_end twenty times on one stream.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations of the chain
(scripts/rocpd_summary.py stats <db>), or plain for the time per call."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fadtk_amd import hip, _capi as K
dev = torch.device("cuda", 0)
d, n, B = 512, 30000, 8
g = torch.Generator(device=dev); g.manual_seed(3)
pairs = []
for k in range(B):
    a = torch.randn((n, d), generator=g, device=dev).to(torch.float16)
    b = (1.1 * torch.randn((n, d), generator=g, device=dev) + 0.01).to(torch.float16)
    ma, mb = hip.Moments(d), hip.Moments(d)
    hip.Moments.update_multi([ma, mb], [a, b])
    pairs.append((ma, mb))
for _ in range(3):
    res = hip.FrechetMultiJob(pairs, mean_dtype=K.FAD_F16).result()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    res = hip.FrechetMultiJob(pairs, mean_dtype=K.FAD_F16).result()
torch.cuda.synchronize()
print(f"chain of {B} pairs: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per call ({(time.perf_counter() - t0) / 20 / B * 1e6:.1f} us per score); fad[0] {res[0][0]:.6f} iters {res[0][1]['iters']}")
