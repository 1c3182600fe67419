#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out
MST_TCN_PERSIST=256 MST_TCN_PROF_BLOCK=5 MST_TCN_PROF_FILE=$R/gpurun_out/persist_phase.bin timeout 600 python bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline > /dev/null 2> gpurun_out/persist_phase.err
MST_TCN_PERSIST=256 timeout 600 python bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline > gpurun_out/var_256.json 2> gpurun_out/var_256.err
MST_TCN_PERSIST=256 timeout 900 python -m pytest tests -m gpu -q -k "bf16 or golden or independent" 2>&1 | tail -5 > gpurun_out/var_pytest.log
