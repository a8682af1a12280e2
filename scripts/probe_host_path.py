#!/usr/bin/env python3
"""GPU probe: the host-buffer route (numpy in, PCIe inside the library) at config 3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import recipes as R
import fadtk_amd
a, b = R.c3_pair()
fadtk_amd.calc_embd_statistics(a[:1000])
for rep in range(3):
    t0 = time.perf_counter(); m1, c1 = fadtk_amd.calc_embd_statistics(a); t1 = time.perf_counter()
    m2, c2 = fadtk_amd.calc_embd_statistics(b); t2 = time.perf_counter()
    f = fadtk_amd.calc_frechet_distance(m1, c1, m2, c2); t3 = time.perf_counter()
    print(f"stats {1e3*(t1-t0):.2f} + {1e3*(t2-t1):.2f} ms, frechet {1e3*(t3-t2):.2f} ms, total {1e3*(t3-t0):.2f} ms -> {1/(t3-t0):.1f} scores/s  fad={f:.6f}")
