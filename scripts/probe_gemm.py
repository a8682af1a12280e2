#!/usr/bin/env python3
"""GPU probe: time the fp64 GEMM kernel variants (FAD_GEMM_DBG) under rocprofv3 --kernel-trace."""
import sys
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fadtk_amd import hip
rng = np.random.default_rng(0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = rng.standard_normal((4 * d, d)); y = 1.1 * rng.standard_normal((4 * d, d))
c1, c2 = np.cov(x, rowvar=False), np.cov(y, rowvar=False)
for _ in range(12):
    try:
        hip.frechet(np.zeros(d), c1, np.zeros(d), c2)
    except Exception as e:      # noqa: BLE001  probes produce garbage numerics
        pass
