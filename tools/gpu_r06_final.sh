#!/bin/bash
# round 6, the full visit on the CURRENT sources: GPU test suite, the driver's bench command, rocprofv3 kernel stats of the driver command and of
# the timed steps alone, FETCH_SIZE / WRITE_SIZE passes for the dominant kernel and the FX chain (fingerprinted), the FX bench + timeline.
# Everything lands in gpurun_out/r06 (copied to profiles/ by hand).   usage: bash tools/gpu_r06_final.sh [skip-tests]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out/r06; mkdir -p $O
(rocminfo | grep -m3 -E "Marketing Name|gfx950|Compute Unit"; rocm-smi --showpower --showclocks 2>/dev/null | head -30) > $O/r06_box.txt 2>&1
if [ "$1" != "skip-tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x -s > $O/r06_pytest_gpu.log 2>&1; tail -4 $O/r06_pytest_gpu.log
fi
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_driver_cmd.json 2> $O/r06_bench_driver_cmd.err; cp gpurun_out/bench_details.json $O/r06_bench_driver_cmd_details.json
cut -c1-900 $O/r06_bench_driver_cmd.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_driver -o bench -- python $R/bench.py --gpus 1 --steps 3 --warmup 1 > $O/prof_driver.json 2> $O/prof_driver.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o bench -- python $R/bench.py --steps 10 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $O/prof_bf16.json 2> $O/prof_bf16.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $O/pmc_write.log 2>&1
cd $R
python tools/rocprof_summary.py "$(find $O/prof_driver -name '*.db' | head -1)" "python bench.py --gpus 1 --steps 3 --warmup 1 (every leg of the driver's command, under rocprofv3 --kernel-trace --stats)" > $O/r06_bench_driver_cmd_kernel_stats.txt 2>&1
python tools/rocprof_summary.py "$(find $O/prof_bf16 -name '*.db' | head -1)" "bench.py --workload configs1 --precision bf16 --steps 10 --warmup 1: the 10 TIMED steps only (first 1/11 of every kernel's dispatches dropped)" --drop-first 0.0909 > $O/r06_bench_bf16_kernel_stats_timed_steps.txt 2>&1
FD=$(dirname $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1)); WD=$(dirname $(find $O/pmc_write -name "*counter_collection.csv" | head -1))
for d in $FD $WD; do f=$(ls $d/*counter_collection.csv | head -1); [ "$f" != "$d/pmc_counter_collection.csv" ] && cp $f $d/pmc_counter_collection.csv; done
python tools/pmc_traffic.py $FD $WD "tcn_block_bf16_kernel<4, false, 8, 2," $O/r06_tcn_block_bf16_traffic.json > $O/pmc_traffic.log 2>&1
N=4 bash tools/gpu_fx_pmc.sh > $O/fx_pmc.log 2>&1; cp gpurun_out/fx_chain_traffic.json $O/r06_fx_chain_traffic.json
timeout 200 python tools/bench_fx.py > $O/r06_bench_fx.json 2> $O/bench_fx.err
cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $O/trace_fx -- python $R/tools/bench_fx.py --chain-only 3 > /dev/null 2>&1; cd $R
python tools/rocprof_trace.py $(find $O/trace_fx -name "*.db" | head -1) "fx_" 22 > $O/r06_fx_chain_timeline.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/prof_driver $O/prof_bf16 $O/pmc_fetch $O/pmc_write $O/trace_fx gpurun_out/pmc_fx_fetch gpurun_out/pmc_fx_write
tail -2 $O/pmc_traffic.log; tail -3 $O/fx_pmc.log; head -12 $O/r06_bench_bf16_kernel_stats_timed_steps.txt; python -c "
import json; d=json.load(open('$O/r06_bench_fx.json')); print('fx chain ms', d['ms_per_chain'], d['per_processor_ms'])"
