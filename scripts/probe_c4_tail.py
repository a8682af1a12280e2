#!/usr/bin/env python3
"""Config-4 moments pass (bench.py extra_c4) for a rocprofv3 --kernel-trace timeline: six updates of 4096 files of [2250 x 128] float16."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fadtk_amd.utils import OnlineStats
dev = torch.device("cuda", 0)
files, rows, d = 4096, 2250, 128
x = torch.randn((files * rows, d), device=dev, dtype=torch.float16)
sizes = np.full(files, rows, dtype=np.int64)
st = OnlineStats(d, 0, compat=True)
st.add_group(x, sizes); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(4): st.add_group(x, sizes)
    torch.cuda.synchronize()
    print(f"pass {rep}: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
st.finish(); st.close()
