#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out
MST_TCN_PROF_BLOCK=5 MST_TCN_PROF_FILE=$R/gpurun_out/phase6_v0.bin timeout 600 python bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline > /dev/null 2> gpurun_out/phase_v0.err
timeout 600 python bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline > gpurun_out/var_0.json 2> gpurun_out/var_0.err
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/var_pytest.log
