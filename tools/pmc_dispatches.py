#!/usr/bin/env python
"""Per-dispatch counter table from one or more rocprofv3 --pmc passes (csv output): the LAST n dispatches whose kernel name contains a pattern,
in dispatch order, one column per counter.   usage: pmc_dispatches.py <pattern> <n> <pass_dir> [<pass_dir> ...]"""
import csv
import glob
import sys


def main():
    pat, n = sys.argv[1], int(sys.argv[2])
    cols, table = [], {}
    for d in sys.argv[3:]:
        for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            rows = {}
            for r in csv.DictReader(open(path)):
                if pat in r["Kernel_Name"]:
                    rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
                    rows[int(r["Dispatch_Id"])]["_grid"] = r.get("Grid_Size", "")
            ids = sorted(rows)[-n:]
            for k, i in enumerate(ids):
                for c, v in rows[i].items():
                    table.setdefault(k, {})[c] = v
                    if c not in cols:
                        cols.append(c)
    cols = [c for c in cols if c != "_grid"]
    print("idx grid " + " ".join(cols))
    for k in sorted(table):
        print(k, table[k].get("_grid", ""), " ".join(f"{table[k].get(c, float('nan')):.4g}" for c in cols))


if __name__ == "__main__":
    main()
