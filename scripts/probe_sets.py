#!/usr/bin/env python3
"""Sets per moments launch (fad_moments_update_multi) at the config-3 set size: tile / reduce durations per launch and per pair."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fadtk_amd import hip  # noqa: E402
dev = torch.device("cuda", 0)
n, d = 100_000, 512
g = torch.Generator(device=dev); g.manual_seed(1)
pool = [(torch.randn((n, d), generator=g, device=dev) * (1 + 0.1 * (k % 2)) + 0.01 * k).to(torch.float16) for k in range(32)]
for sets in (2, 4, 8, 12, 16):
    accs = [hip.Moments(d) for _ in range(sets)]
    groups = [pool[i:i + sets] for i in range(0, 32 - sets + 1, sets)]
    for a in accs: a.reset()
    hip.Moments.update_multi(accs, groups[0]); torch.cuda.synchronize()
    accs[0].set_timing(1)
    reps = 12
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for r in range(reps):
        for a in accs: a.reset()
        hip.Moments.update_multi(accs, groups[r % len(groups)])
    t1.record(); torch.cuda.synchronize()
    k_ms, r_ms, variant = accs[0].last_timing()
    tot = t0.elapsed_time(t1) / reps
    fl = sets * 2.0 * n * d * d
    print(f"{sets} sets per launch (variant {variant}): tile {k_ms*1e3:7.1f} us ({fl / k_ms / 1e9 / 2500 * 100:5.1f} % of 2.5 PF)  guard+reduce {r_ms*1e3:6.1f} us  update {tot*1e3:7.1f} us = {tot*1e3/(sets/2):6.1f} us per pair", flush=True)
    for a in accs: a.close()
