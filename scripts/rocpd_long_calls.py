#!/usr/bin/env python3
"""The longest entries of every timed table of a rocprofv3 rocpd SQLite database (HIP API regions with --hip-trace, kernels with
--kernel-trace, memory copies ...): what took more than <min_ms> (default 5).    python scripts/rocpd_long_calls.py <results.db> [min_ms]"""
import sqlite3
import sys

db = sys.argv[1]
min_ns = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
con = sqlite3.connect(db)
cur = con.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table', 'view')")]
for t in names:
    try:
        cols = [r[1] for r in cur.execute(f"pragma table_info('{t}')")]
    except sqlite3.Error:
        continue
    if "start" not in cols or "end" not in cols:
        continue
    try:
        rows = cur.execute(f"select * from '{t}' where (\"end\" - start) >= ? order by (\"end\" - start) desc limit 12", (min_ns,)).fetchall()
    except sqlite3.Error as e:
        print(f"# {t}: {e}")
        continue
    if not rows:
        continue
    print(f"== {t} ({len(rows)} entries >= {min_ns / 1e6:g} ms)")
    for r in rows:
        d = dict(zip(cols, r))
        dur = (d["end"] - d["start"]) / 1e6
        txt = {k: v for k, v in d.items() if isinstance(v, str) and len(v) < 120}
        ids = {k: v for k, v in d.items() if k in ("tid", "pid", "queue_id", "stream_id", "name_id", "category", "size")}
        print(f"  {dur:9.3f} ms  start {d['start'] / 1e6:.1f} ms  {txt} {ids}")
