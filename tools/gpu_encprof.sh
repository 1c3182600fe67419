#!/bin/bash
# kernel stats of a short bf16 bench run (encoder kernels in focus) + encoder parity tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -k "encoder or golden" 2>&1 | tail -3 > gpurun_out/enc_pytest.log
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_enc" -o enc -- python "$R/bench.py" --steps 3 --warmup 1 --precision bf16 --no-cpu-baseline > "$R/gpurun_out/prof_enc.json" 2> "$R/gpurun_out/prof_enc.err"
