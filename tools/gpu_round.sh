#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (the driver's exact command, all legs), rocprofv3 kernel stats of the bf16 / split-bf16 steps and the
# FX chain; WITH_PMC=1 adds the FETCH_SIZE / WRITE_SIZE passes (TCN block kernel, FX chain) and the SQ counters.  Everything lands in
# gpurun_out/round/ under the names it is committed with in profiles/ (prefix ${TAG:-r04_final}_).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); T=${TAG:-r05_final}; O=$R/gpurun_out/round; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(rocm-smi --showproductname; lscpu | head -20; nproc) > $O/${T}_box.txt 2>&1
if [ -z "$SKIP_TESTS" ]; then
timeout 2400 python -m pytest tests -m gpu -q -rA --durations=15 2>&1 | tail -120 > $O/${T}_pytest_gpu.log
fi
timeout 300 python __graft_entry__.py --smoke > $O/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $O/${T}_smoke.log
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_driver_cmd.json 2> $O/${T}_bench_driver_cmd.err; echo "rc=$?" >> $O/${T}_bench_driver_cmd.err
cp gpurun_out/bench_details.json $O/${T}_bench_details.json
timeout 600 python bench.py --precision bf16x3 --workload configs1 --steps 4 --warmup 2 --no-cpu-baseline > $O/${T}_bench_bf16x3.json 2> $O/${T}_bench_x3.err
# N > 1 code path on the single-GPU box: 2 ranks sharing cuda:0, gloo for the collective
MST_BENCH_SHARE_GPU=1 MST_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 > $O/${T}_bench_2rank_one_gpu_gloo.json 2> $O/${T}_bench_2rank.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o bench -- python $R/bench.py --steps 3 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $O/prof_bf16.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_x3 -o bench -- python $R/bench.py --steps 2 --warmup 1 --precision bf16x3 --workload configs1 --no-cpu-baseline > $O/prof_x3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_fx -o bench -- python $R/tools/bench_fx.py > $O/prof_fx.log 2>&1
if [ -n "$WITH_PMC" ]; then
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $O/pmc_write.log 2>&1
fi
cd $R
python tools/rocprof_summary.py "$(find $O/prof_bf16 -name '*.db' | head -1)" "bench.py --workload configs1 --precision bf16 (3 steps + 1 warm-up)" > $O/${T}_bench_bf16_kernel_stats.txt 2>&1
python tools/rocprof_summary.py "$(find $O/prof_x3 -name '*.db' | head -1)" "bench.py --workload configs1 --precision bf16x3 (2 steps + 1 warm-up)" > $O/${T}_bench_bf16x3_kernel_stats.txt 2>&1
python tools/rocprof_summary.py "$(find $O/prof_fx -name '*.db' | head -1)" "tools/bench_fx.py" > $O/${T}_bench_fx_kernel_stats.txt 2>&1
if [ -n "$WITH_PMC" ]; then
FD=$(dirname $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1)); WD=$(dirname $(find $O/pmc_write -name "*counter_collection.csv" | head -1))
for d in $FD $WD; do f=$(ls $d/*counter_collection.csv | head -1); [ "$f" != "$d/pmc_counter_collection.csv" ] && cp $f $d/pmc_counter_collection.csv; done
python tools/pmc_traffic.py $FD $WD tcn_block_bf16_duo_kernel $O/r05_tcn_block_bf16_traffic.json > $O/pmc_traffic.log 2>&1
N=4 bash tools/gpu_fx_pmc.sh > $O/fx_pmc.log 2>&1; cp gpurun_out/fx_chain_traffic.json $O/r05_fx_chain_traffic.json
fi
find $O -name "*.db" -delete; rm -rf $O/pmc_fetch $O/pmc_write $O/prof_bf16 $O/prof_x3 $O/prof_fx; ls $O
tail -3 $O/${T}_pytest_gpu.log; cat $O/${T}_smoke.log | tail -4; cat $O/${T}_bench_driver_cmd.json
