#!/usr/bin/env python
"""HBM traffic of ONE FX chain (BASELINE configs[3]) from two rocprofv3 --pmc passes of `tools/bench_fx.py --chain-only N`
(FETCH_SIZE, WRITE_SIZE in KiB; FETCH_SIZE doubled as the microarch guide prescribes for wide coalesced reads on gfx950 - the narrow
accesses of some FX kernels make that an upper bound, stated per kernel as raw and corrected).
usage: pmc_fx_traffic.py <fetch_dir> <write_dir> <chains incl. the warm-up> <out.json>"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import defaultdict


def table(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].split("(")[0]
            tot[k] += float(r["Counter_Value"])
            n[k] += 1
    return tot, n


def main():
    fetch_dir, write_dir, chains, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, nf = table(f"{fetch_dir}/pmc_counter_collection.csv", "FETCH_SIZE")
    w, _ = table(f"{write_dir}/pmc_counter_collection.csv", "WRITE_SIZE")
    n_items, L = 64, 131072
    rows = {}
    for k in sorted(set(f) | set(w)):
        if not (k.startswith("fx_") or "fx_" in k):
            continue
        rows[k] = {"launches_per_chain": nf.get(k, 0) / chains, "fetch_KiB_raw_per_chain": f.get(k, 0.0) / chains,
                   "write_KiB_per_chain": w.get(k, 0.0) / chains}
    rd = sum(v["fetch_KiB_raw_per_chain"] for v in rows.values()) * 1024.0
    wr = sum(v["write_KiB_per_chain"] for v in rows.values()) * 1024.0
    fused = 16.0 * L * n_items
    res = {"workload": "configs[3] chain on 64 x [131072, 2] float32 (AugmentationChain, fused rms-normalise)", "chains_counted": chains,
           "read_bytes_raw": rd, "read_bytes_x2": 2.0 * rd, "write_bytes": wr, "traffic_bytes": 2.0 * rd + wr,
           "one_read_one_write_bytes": fused, "traffic_over_one_read_one_write": (2.0 * rd + wr) / fused,
           "unfused_bytes_survey_8d": 144.0 * L * n_items, "per_kernel": rows,
           "note": "traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes); the x2 is the guide's gfx950 correction for 16-byte-per-lane streams"}
    import bench
    res["csrc_sha256"] = bench.csrc_sha256("fx")      # bench.py prints this figure only while the FX kernels' sources are these
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_kernel"}))


if __name__ == "__main__":
    main()
