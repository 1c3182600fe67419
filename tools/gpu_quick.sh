#!/bin/bash
# short visit: FX chain profile + bf16x3 / fp32 step profiles (no test suite)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests -m gpu -q -x -k "fx or chain or encoder or normalizer or standalone" 2>&1 | tail -15 > gpurun_out/pytest_quick.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_fx" -o bench -- python "$R/tools/bench_fx.py" > "$R/gpurun_out/prof_fx.log" 2>&1
timeout 600 python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/bench_default.json" 2> "$R/gpurun_out/bench_default.err"
timeout 900 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_x3" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --precision bf16x3 --workload configs1 --no-cpu-baseline > "$R/gpurun_out/prof_x3.log" 2>&1
cd "$R"
for k in x3 fx; do DB=$(find gpurun_out/prof_$k -name "*.db" | head -1); python tools/rocprof_summary.py "$DB" "prof_$k" > gpurun_out/prof_${k}_kernel_stats.txt 2>&1; done
find gpurun_out -name "*.db" -size +20M -delete; ls gpurun_out | head -40
