#!/usr/bin/env python3
"""GPU probe, run under rocprofv3 --kernel-trace: a few config-4 groups of 4096 files through OnlineStats (what bench.py's
extra.c4_moments times) so that the kernel statistics show where a group's time goes (scripts/probes/hbm_read.hip gives
the plain-read ceiling of the chip for the same bytes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fadtk_amd.utils import OnlineStats

files = 4096
x = torch.randn((files * 2250, 128), device="cuda", dtype=torch.float16)
sizes = np.full(files, 2250, dtype=np.int64)
st = OnlineStats(128, 0, compat=True)
for _ in range(3): st.add_group(x, sizes)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): st.add_group(x, sizes)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
print(f"group of {files} files: {t*1e6:.1f} us = {x.numel()*2/t/1e12:.2f} TB/s")
st.close()
