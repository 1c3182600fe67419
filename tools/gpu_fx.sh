#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_fx" -o fx -- python "$R/tools/bench_fx.py" > "$R/gpurun_out/bench_fx.json" 2> "$R/gpurun_out/bench_fx.err"
