#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db, ROCm 7.2 default output) into the
text table committed under profiles/: per kernel calls, total ms, avg/min/max us, % of GPU kernel time,
plus VGPR/LDS of the dispatch.   usage: rocprof_summary.py <results.db> [title] [--drop-first FRACTION]
--drop-first f: per kernel name, the first round(f * calls) dispatches (in start order) are left out - with `bench.py --steps K --warmup 1`
and f = 1/(K+1) that is the warm-up step, so the table describes the timed steps only."""
import sqlite3
import sys


def main():
    argv = list(sys.argv[1:])
    drop = 0.0
    if "--drop-first" in argv:
        i = argv.index("--drop-first")
        drop = float(argv[i + 1])
        del argv[i:i + 2]
    db = argv[0]
    title = argv[1] if len(argv) > 1 else db
    c = sqlite3.connect(db)
    if drop > 0.0:
        per = {}
        for name, start, end, vg, ag, lds, wgs in c.execute(
                "select name, start, end, vgpr_count, accum_vgpr_count, lds_size, grid_x/workgroup_x*grid_y from kernels order by start"):
            per.setdefault(name, []).append((end - start, vg, ag, lds, wgs))
        rows = []
        for name, ds in per.items():
            ds = ds[int(round(drop * len(ds))):]
            if not ds:
                continue
            t = [d[0] for d in ds]
            rows.append((name, len(ds), sum(t), sum(t) / len(t), min(t), max(t), max(d[1] for d in ds), max(d[2] for d in ds),
                         max(d[3] for d in ds), max(d[4] for d in ds)))
        rows.sort(key=lambda r: -r[2])
    else:
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
