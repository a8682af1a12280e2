#!/usr/bin/env python3
"""GPU probe: what numpy's float32 running column sums (fad_moments_set_reference_mean; csrc/moments_kernels.h:
moments_running_colsum_h16) cost an update of S sets of [100000 x 512] float16 frames, with the walk on the device's side stream
(default) and in line on the caller's stream (FAD_MOMENTS_RUNSUM_SIDE=0, read once per process: run the probe twice), and whether the
mean it leads to is numpy's bit for bit on frames with an offset."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fadtk_amd import hip

d, n = 512, 100000
g = torch.Generator(device="cuda"); g.manual_seed(5)
mats = [(torch.randn((n, d), generator=g, device="cuda") * 0.7 + 0.5).to(torch.float16) for _ in range(8)]
side = os.environ.get("FAD_MOMENTS_RUNSUM_SIDE", "1")
for sets in (2, 8):
    for ref in (0, 1):
        hs = [hip.Moments(d) for _ in range(sets)]
        for h in hs:
            h.set_reference_mean(bool(ref))
        for _ in range(3):
            for h in hs: h.reset()
            hip.Moments.update_multi(hs, mats[:sets])
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            for h in hs: h.reset()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            hip.Moments.update_multi(hs, mats[:sets])
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"side={side} sets={sets} reference mean {'on ' if ref else 'off'}: update {np.median(ts):.3f} ms (min {min(ts):.3f})", flush=True)
        if ref:
            mu = hs[0].finalize()[0]
            want = mats[0].cpu().numpy().mean(axis=0)                  # numpy: float32 running sum, float16 result
            got16 = mu.astype(np.float32).astype(np.float16)
            print(f"   mean vs numpy's: {int((got16 != want).sum())} of {d} float16 values differ; float32 quotient max |diff| vs np.mean(dtype=float32-order) "
                  f"{np.abs(mu - (mats[0].cpu().numpy().astype(np.float32).cumsum(axis=0, dtype=np.float32)[-1] / np.float32(n)).astype(np.float64)).max():.3e}", flush=True)
        for h in hs: h.close()
# rows not a multiple of the tile (384), an odd pitch view, a second update carried on
x = mats[0][:100003 - 100000 + 77777]
with hip.Moments(d) as h:
    h.set_reference_mean(True)
    h.update(x[:50001]); h.update(x[50001:])
    mu = h.finalize()[0]
    want = np.cumsum(x.cpu().numpy().astype(np.float32), axis=0, dtype=np.float32)[-1] / np.float32(x.shape[0])
    print(f"two updates of 50001 + {x.shape[0] - 50001} rows: float32 mean max |diff| vs the sequential sum {np.abs(mu - want.astype(np.float64)).max():.3e}", flush=True)
