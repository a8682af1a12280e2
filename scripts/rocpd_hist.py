#!/usr/bin/env python3
"""Per-kernel duration distribution by grid size from a rocpd DB (median / p10 / p90, us)."""
import sqlite3, sys, re
import numpy as np
cur = sqlite3.connect(sys.argv[1]).cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else "gemm"
rows = cur.execute("select name, grid_x*grid_y*grid_z, duration from kernels").fetchall()
groups = {}
for n, g, dur in rows:
    if pat in n:
        groups.setdefault((re.sub(r"\(.*", "", n)[-60:], g), []).append(dur / 1e3)
for (n, g), v in sorted(groups.items()):
    v = np.array(v)
    print(f"{n:60s} grid={g:8d} n={len(v):4d} p10={np.percentile(v,10):7.2f} med={np.median(v):7.2f} p90={np.percentile(v,90):7.2f}")
