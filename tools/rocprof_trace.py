#!/usr/bin/env python
"""List the individual dispatches of a rocprofv3 --kernel-trace run (rocpd sqlite .db) whose kernel name contains a pattern, in launch order:
grid in workgroups, duration.   usage: rocprof_trace.py <results.db> <pattern> [last N]"""
import sqlite3
import sys


def main():
    db, pat = sys.argv[1], sys.argv[2]
    last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end, grid_x/workgroup_x, grid_y/workgroup_y, grid_z/workgroup_z, lds_size, vgpr_count from kernels "
                          "where name like ? order by start", (f"%{pat}%",)))
    if last:
        rows = rows[-last:]
    t0 = rows[0][1] if rows else 0
    for r in rows:
        print(f"{(r[1] - t0) / 1e3:10.1f} us  {(r[2] - r[1]) / 1e3:8.1f} us  grid {r[3]:6d} x {r[4]:3d} x {r[5]:2d}  lds {r[6]:6d} vgpr {r[7]:3d}  {r[0][:80]}")


if __name__ == "__main__":
    main()
