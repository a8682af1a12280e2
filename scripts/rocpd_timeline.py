#!/usr/bin/env python3
"""Timeline of the last N dispatches of a rocprofv3 run (rocpd SQLite): start offset, duration, stream-ish queue id, name --
shows which kernels of different streams actually overlap.    python scripts/rocpd_timeline.py <results.db> [N]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:]+(?:<[^()]{0,40}>)?)", name)
    return (m.group(1) if m else name)[:48]


db = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = cur.execute(f"select name, start, end, {q} from kernels order by start desc limit ?", (n,)).fetchall()[::-1]
t0 = rows[0][1]
busy_end = None
print("start_us,dur_us,queue,overlaps_prev_end_us,kernel")
for name, st, en, qid in rows:
    ov = (busy_end - st) / 1e3 if busy_end is not None and busy_end > st else 0.0
    print(f"{(st - t0) / 1e3:9.1f},{(en - st) / 1e3:7.1f},{qid},{ov:6.1f},{short(name)}")
    busy_end = en if busy_end is None or en > busy_end else busy_end
span = (max(r[2] for r in rows) - t0) / 1e3
tot = sum(r[2] - r[1] for r in rows) / 1e3
print(f"# span {span:.1f} us, sum of kernel durations {tot:.1f} us, ratio {tot / span:.2f}")
