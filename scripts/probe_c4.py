#!/usr/bin/env python3
"""GPU probe: where the time of a config-4 group (files of [2250 x 128] fp16 frames, resident in HBM) goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fadtk_amd import hip
from fadtk_amd.utils import OnlineStats

def timed(fn, n=8):
    fn(); fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

for files in (256, 1024, 4096):
    x = torch.randn((files * 2250, 128), device="cuda", dtype=torch.float16)
    sizes = np.full(files, 2250, dtype=np.int64); offs = np.concatenate([[0], np.cumsum(sizes)])
    nbytes = x.numel() * 2
    m = hip.Moments(128)
    t_upd = timed(lambda: m.update(x))
    t_seg = timed(lambda: m.update_segmented(x, offs, want_sums=True, sums_on_device=True))
    st = OnlineStats(128, 0, compat=True)
    t_all = timed(lambda: st.add_group(x, sizes))
    m.set_timing(True); m.update_segmented(x, offs, want_sums=True, sums_on_device=True); k, r, _ = m.last_timing()
    print(f"files={files:5d} {nbytes/1e6:7.1f} MB  update {t_upd:7.1f} us ({nbytes/t_upd/1e6:5.2f} TB/s)  segmented {t_seg:7.1f} us ({nbytes/t_seg/1e6:5.2f} TB/s) "
          f"[tile {k*1e3:6.1f} reduce {r*1e3:6.1f}]  + file means {t_all:7.1f} us ({nbytes/t_all/1e6:5.2f} TB/s = {nbytes/t_all/1e6/8*100:4.1f}% of 8 TB/s)", flush=True)
    m.close(); st.close(); del x
