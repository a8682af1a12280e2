#!/usr/bin/env python3
"""moments_reduce256 split lanes (FAD_MOMENTS_R256_SL) for launches of 2 and 8 frame matrices at the config-3 set size: guard + reduce time."""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fadtk_amd import hip
dev = torch.device("cuda", 0)
n, d = 100_000, 512
g = torch.Generator(device=dev); g.manual_seed(1)
pool = [(torch.randn((n, d), generator=g, device=dev) * (1 + 0.1 * (k % 2)) + 0.01 * k).to(torch.float16) for k in range(16)]
for sets in (2, 8):
    for sl in ("0", "1", "2", "4", "8", "16"):
        os.environ["FAD_MOMENTS_R256_SL"] = sl
        accs = [hip.Moments(d) for _ in range(sets)]
        groups = [pool[i:i + sets] for i in range(0, 16 - sets + 1, sets)]
        for a in accs: a.reset()
        hip.Moments.update_multi(accs, groups[0]); torch.cuda.synchronize()
        accs[0].set_timing(1)
        for r in range(12):
            for a in accs: a.reset()
            hip.Moments.update_multi(accs, groups[r % len(groups)])
        k_ms, r_ms, _ = accs[0].last_timing()
        chk = float(accs[-1].export()[5])
        print(f"{sets} sets, split lanes {sl:>2}: tile {k_ms*1e3:7.1f} us  guard+reduce {r_ms*1e3:6.1f} us   [acc[5] = {chk:.6f}]", flush=True)
        for a in accs: a.close()
