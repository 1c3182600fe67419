#!/bin/bash
# refresh after a change of the split-bf16 kernels only: the driver's bench command (all legs), the x3 bench on its own and its rocprofv3 kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); T=${TAG:-r04_final}; O=$R/gpurun_out/round; mkdir -p $O
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_driver_cmd.json 2> $O/${T}_bench_driver_cmd.err; echo "rc=$?" >> $O/${T}_bench_driver_cmd.err
cp gpurun_out/bench_details.json $O/${T}_bench_details.json
timeout 600 python bench.py --precision bf16x3 --workload configs1 --steps 4 --warmup 2 --no-cpu-baseline > $O/${T}_bench_bf16x3.json 2> $O/${T}_bench_x3.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_x3 -o bench -- python $R/bench.py --steps 2 --warmup 1 --precision bf16x3 --workload configs1 --no-cpu-baseline > $O/prof_x3.log 2>&1
cd $R
python tools/rocprof_summary.py "$(find $O/prof_x3 -name '*.db' | head -1)" "bench.py --workload configs1 --precision bf16x3 (2 steps + 1 warm-up)" > $O/${T}_bench_bf16x3_kernel_stats.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/prof_x3
cat $O/${T}_bench_driver_cmd.json; head -8 $O/${T}_bench_bf16x3_kernel_stats.txt | cut -c1-150
