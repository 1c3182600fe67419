#!/bin/bash
# shortest visit: the FX / normaliser GPU tests and the FX chain profile
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd)
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "fx or chain or normalizer or reverb" 2>&1 | tail -15 > gpurun_out/pytest_fx.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_fx" -o bench -- python "$R/tools/bench_fx.py" > "$R/gpurun_out/prof_fx.log" 2>&1
cd "$R"
DB=$(find gpurun_out/prof_fx -name "*.db" | head -1); python tools/rocprof_summary.py "$DB" "prof_fx" > gpurun_out/prof_fx_kernel_stats.txt 2>&1
find gpurun_out -name "*.db" -size +20M -delete; ls gpurun_out | head -40
