"""Why are the first tile-kernel launches of a process slow?  (260-297 us against 214-221 us warm.)
Per-launch durations of moments_tile256 (the dispatch's own begin / end stamps) for the first 24 updates of a fresh process, under four
conditions: as they come / after 60 ms of unrelated GPU work (clocks up, the library's buffers untouched) / after one update on OTHER
handles of the same shape (buffers of THESE handles untouched, code object loaded) / handles whose buffers were touched by a 1-row...
Usage: python scripts/probe_cold.py <mode>   mode in {plain, spin, other, sleep}"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from fadtk_amd import hip  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
sets = [torch.randn((100000, 512), device=dev, generator=g).to(torch.float16) for _ in range(8)]
torch.cuda.synchronize()
hs = [hip.Moments(512) for _ in range(8)]
hs[0].set_timing(2)
if mode == "spin":
    x = torch.randn((4096, 4096), device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.06:
        x = (x @ x) * 1e-4
    torch.cuda.synchronize()
elif mode == "other":
    os_ = [hip.Moments(512) for _ in range(8)]
    for _ in range(3):
        hip.Moments.update_multi(os_, sets)
    torch.cuda.synchronize()
elif mode == "sleep":
    hip.Moments.update_multi(hs, sets); torch.cuda.synchronize(); hs[0].last_timing()
    time.sleep(0.5)                                  # the GPU idles: do the clocks fall back?
out = []
for i in range(24):
    for h in hs:
        h.reset()
    hip.Moments.update_multi(hs, sets)
    torch.cuda.synchronize()
    out.append(hs[0].last_timing()[0] * 1e3)
print(mode, " ".join(f"{v:.0f}" for v in out))
