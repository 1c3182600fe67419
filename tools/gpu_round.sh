#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (bf16 + fp32), FX bench, rocprofv3 kernel stats + PMC traffic.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
(rocm-smi --showproductname; lscpu | head -20; nproc) > gpurun_out/box.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rA 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 --precision bf16 > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err; echo "rc=$?" >> gpurun_out/bench_bf16.err
timeout 900 python bench.py --steps 3 --warmup 1 --precision fp32 --no-cpu-baseline > gpurun_out/bench_fp32.json 2> gpurun_out/bench_fp32.err; echo "rc=$?" >> gpurun_out/bench_fp32.err
timeout 600 python tools/bench_fx.py > gpurun_out/bench_fx.json 2> gpurun_out/bench_fx.err; echo "rc=$?" >> gpurun_out/bench_fx.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_bf16" -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --precision bf16 --no-cpu-baseline > "$R/gpurun_out/prof_bf16.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_fp32" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --precision fp32 --no-cpu-baseline > "$R/gpurun_out/prof_fp32.log" 2>&1
if [ -n "$WITH_PMC" ]; then
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/gpurun_out/pmc_sq -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline > $R/gpurun_out/pmc_sq.log 2>&1
fi
cd "$R"; find gpurun_out -name "*.db" -size +20M -delete; ls gpurun_out | head -50
