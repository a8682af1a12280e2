#!/usr/bin/env python3
"""GPU probe: the host-buffer route (numpy in, PCIe inside the library) at config 3.
FAD_H2D_MODE / FAD_H2D_THREADS / FAD_H2D_CHUNK_KB select the staging route (csrc/host_stage.cpp); run once per setting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import recipes as R
import fadtk_amd
from fadtk_amd import hip
a, b = R.c3_pair()
fadtk_amd.calc_embd_statistics(a[:1000])
tag = "mode=%s threads=%s chunk_kb=%s" % (os.environ.get("FAD_H2D_MODE", "threads"), os.environ.get("FAD_H2D_THREADS", "8"),
                                          os.environ.get("FAD_H2D_CHUNK_KB", "4096"))
# the copy + moments alone (no finalize, no D2H)
with hip.Moments(a.shape[1]) as m:
    m.update(a)
    import torch
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); m.reset(); m.update(a); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"[{tag}] update(host [100000 x 512] f16): best {1e3*best:.2f} ms = {a.nbytes/best/1e9:.1f} GB/s")
for rep in range(3):
    t0 = time.perf_counter(); m1, c1 = fadtk_amd.calc_embd_statistics(a); t1 = time.perf_counter()
    m2, c2 = fadtk_amd.calc_embd_statistics(b); t2 = time.perf_counter()
    f = fadtk_amd.calc_frechet_distance(m1, c1, m2, c2); t3 = time.perf_counter()
    print(f"[{tag}] stats {1e3*(t1-t0):.2f} + {1e3*(t2-t1):.2f} ms, frechet {1e3*(t3-t2):.2f} ms, total {1e3*(t3-t0):.2f} ms -> {1/(t3-t0):.1f} scores/s  fad={f:.9f}")
