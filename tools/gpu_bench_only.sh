#!/bin/bash
# the driver's bench command alone (line + details file)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/benchonly; mkdir -p $O
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "rc=$?"
wc -c $O/bench_driver_cmd.json; cat $O/bench_driver_cmd.json; cp gpurun_out/bench_details.json $O/
