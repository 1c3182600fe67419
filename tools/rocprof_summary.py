#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db, ROCm 7.2 default output) into the
text table committed under profiles/: per kernel calls, total ms, avg/min/max us, % of GPU kernel time,
plus VGPR/LDS of the dispatch.   usage: rocprof_summary.py <results.db> [title]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else db
    c = sqlite3.connect(db)
    rows = list(c.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), "
        "max(accum_vgpr_count), max(lds_size), max(grid_x/workgroup_x*grid_y) from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows)
    print(f"# {title}")
    print(f"# rocprofv3 --kernel-trace --stats ; total GPU kernel time {total / 1e6:.3f} ms")
    print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s} {'vgpr':>5s} {'agpr':>5s} {'lds':>7s} {'max_wgs':>8s}")
    for r in rows:
        print(f"{r[0][:72]:72s} {r[1]:6d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.1f} {r[4] / 1e3:10.1f} {r[5] / 1e3:10.1f} "
              f"{100.0 * r[2] / total:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:7d} {r[9]:8d}")


if __name__ == "__main__":
    main()
