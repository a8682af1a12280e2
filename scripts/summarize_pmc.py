#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs: per-kernel mean counter value per dispatch."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])
for f in sorted(root.rglob("*counter_collection.csv")):
    acc = defaultdict(lambda: [0.0, 0])
    with open(f) as fh:
        for row in csv.DictReader(fh):
            key = (row.get("Kernel_Name", "?")[:70], row.get("Counter_Name", "?"))
            acc[key][0] += float(row.get("Counter_Value", 0) or 0)
            acc[key][1] += 1
    print(f"# {f}")
    for (k, c), (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"{c:12s} mean/dispatch={s / max(n, 1):14.1f} dispatches={n:6d}  {k}")
