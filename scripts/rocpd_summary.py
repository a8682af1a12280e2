#!/usr/bin/env python3
"""Turn rocprofv3's rocpd SQLite output into small text summaries (kernel stats / PMC means).

    python scripts/rocpd_summary.py stats  <results.db>  > profiles/rNN_kernel_stats.csv
    python scripts/rocpd_summary.py pmc    <results.db>  > profiles/rNN_pmc_<counter>.csv
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:]+(?:<[^()]{0,60}>)?)", name)
    s = m.group(1) if m else name
    return s[:90]


def main():
    mode, db = sys.argv[1], sys.argv[2]
    cur = sqlite3.connect(db).cursor()
    if mode == "stats":
        print("kernel,calls,total_us,avg_us,min_us,max_us,pct,vgpr,agpr,lds_bytes,grid,workgroup")
        rows = cur.execute(
            "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, "
            "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(grid_x*grid_y*grid_z), max(workgroup_x) "
            "from kernels group by name order by sum(duration) desc").fetchall()
        total = sum(r[2] for r in rows) or 1.0
        for r in rows[:40]:
            print(f"\"{short(r[0])}\",{r[1]},{r[2]:.1f},{r[3]:.2f},{r[4]:.2f},{r[5]:.2f},{100 * r[2] / total:.2f},"
                  f"{r[6]},{r[7]},{r[8]},{r[9]},{r[10]}")
    elif mode == "seq":
        # the last N dispatches in launch order: name, duration, gap to the previous kernel's end (us)
        n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
        rows = cur.execute("select name, start, end, grid_x*grid_y*grid_z, workgroup_x from kernels order by start desc limit ?", (n,)).fetchall()[::-1]
        print("kernel,dur_us,gap_us,grid,workgroup")
        prev_end = None
        for name, st, en, grid, wg in rows:
            gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
            print(f"\"{short(name)}\",{(en - st) / 1e3:.2f},{gap:.2f},{grid},{wg}")
            prev_end = en
    elif mode == "schema":
        for (name,) in cur.execute("select name from sqlite_master where type in ('table', 'view') and name like '%counter%'").fetchall():
            print(name, [r[1] for r in cur.execute(f"pragma table_info('{name}')").fetchall()])
    elif mode == "pmcgrid":
        # counter means per (kernel, grid size): launches of one kernel that carry different numbers of frame matrices apart
        cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')").fetchall()]
        grid = "grid_size" if "grid_size" in cols else ("grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None))
        print("kernel,counter,grid,mean_per_dispatch,dispatches")
        if grid is None:
            print("# no grid column in counters_collection:", cols)
            return
        for r in cur.execute(f"select kernel_name, counter_name, {grid}, avg(value), count(*) from counters_collection "
                             f"group by kernel_name, counter_name, {grid} order by kernel_name, counter_name, {grid}"):
            print(f"\"{short(r[0])}\",{r[1]},{r[2]},{r[3]:.1f},{r[4]}")
    else:
        print("kernel,counter,mean_per_dispatch,dispatches")
        for r in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                             "group by kernel_name, counter_name order by avg(value) desc limit 40"):
            print(f"\"{short(r[0])}\",{r[1]},{r[2]:.1f},{r[3]}")


if __name__ == "__main__":
    main()
