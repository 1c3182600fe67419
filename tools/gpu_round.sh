#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (all legs), 60-min track, 2-rank path on one GPU, rocprofv3 kernel stats (+ PMC with WITH_PMC=1).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
(rocm-smi --showproductname; lscpu | head -20; nproc) > gpurun_out/box.txt 2>&1
if [ -z "$SKIP_TESTS" ]; then
timeout 2400 python -m pytest tests -m gpu -q -rA -s --durations=25 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
fi
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "rc=$?" >> gpurun_out/bench_default.err
timeout 600 python bench.py --precision bf16x3 --workload configs1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_x3.json 2> gpurun_out/bench_x3.err
timeout 600 python bench.py --precision bf16x3 --workload configs1 --steps 3 --warmup 1 --no-cpu-baseline --x3-large-tiles > gpurun_out/bench_x3_large.json 2> gpurun_out/bench_x3_large.err
timeout 900 python bench.py --workload track60 --steps 3 --warmup 1 > gpurun_out/bench_track60.json 2> gpurun_out/bench_track60.err; echo "rc=$?" >> gpurun_out/bench_track60.err
# N > 1 code path on the single-GPU box: 2 ranks sharing cuda:0, gloo for the collective
MST_BENCH_SHARE_GPU=1 MST_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank_gloo.err; echo "rc=$?" >> gpurun_out/bench_2rank_gloo.err
# RCCL initialises with one rank per process (world 1 through torchrun)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 2 --warmup 1 --workload configs1 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err; echo "rc=$?" >> gpurun_out/bench_torchrun1.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_bf16" -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > "$R/gpurun_out/prof_bf16.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_x3" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --precision bf16x3 --workload configs1 --no-cpu-baseline > "$R/gpurun_out/prof_x3.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_fx" -o bench -- python "$R/tools/bench_fx.py" > "$R/gpurun_out/prof_fx.log" 2>&1
if [ -n "$WITH_FP32_PROF" ]; then
timeout 900 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_fp32" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --precision fp32 --workload configs1 --no-cpu-baseline > "$R/gpurun_out/prof_fp32.log" 2>&1
fi
if [ -n "$WITH_PMC" ]; then
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/gpurun_out/pmc_sq -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $R/gpurun_out/pmc_sq.log 2>&1
fi
cd "$R"; DB=$(find gpurun_out/prof_bf16 -name "*.db" | head -1); python tools/rocprof_summary.py "$DB" "bench.py --workload configs1 --precision bf16 (3 steps + 1 warm-up)" > gpurun_out/prof_bf16_kernel_stats.txt 2>&1
for k in x3 fx; do DB=$(find gpurun_out/prof_$k -name "*.db" | head -1); python tools/rocprof_summary.py "$DB" "prof_$k" > gpurun_out/prof_${k}_kernel_stats.txt 2>&1; done
find gpurun_out -name "*.db" -size +20M -delete; ls gpurun_out | head -80
