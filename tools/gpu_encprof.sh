#!/bin/bash
# per-launch trace of the encoder kernels of one bf16 step (launch order, grid, duration)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_enc" -o enc -- python "$R/bench.py" --steps 2 --warmup 1 --precision ${PREC:-bf16} --workload configs1 --no-cpu-baseline $EXTRA > "$R/gpurun_out/prof_enc.json" 2> "$R/gpurun_out/prof_enc.err"
cd "$R"; DB=$(find gpurun_out/prof_enc -name "*.db" | head -1)
python tools/rocprof_trace.py "$DB" enc_ 38 > gpurun_out/enc_trace.txt 2>&1
find gpurun_out -name "*.db" -size +20M -delete
cat gpurun_out/enc_trace.txt
