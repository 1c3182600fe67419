#!/bin/bash
# A/B the bf16 TCN main-loop variants (env MST_TCN_BF16_VARIANT) + phase stamps of block 5; quick parity check.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out
for v in ${VARIANTS:-0 1 2}; do
  MST_TCN_BF16_VARIANT=$v MST_TCN_PROF_BLOCK=5 MST_TCN_PROF_FILE=$R/gpurun_out/phase_v$v.bin timeout 600 python bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline > /dev/null 2> gpurun_out/phase_v$v.err
  MST_TCN_BF16_VARIANT=$v timeout 600 python bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err
done
timeout 900 python -m pytest tests -m gpu -q -k "bf16 or golden or independent" 2>&1 | tail -5 > gpurun_out/var_pytest.log
