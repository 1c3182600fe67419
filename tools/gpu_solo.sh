#!/bin/bash
# experimental one-set persistent TCN kernel on 512-time tiles (MST_TCN_SOLO): bench + phase stamps + GPU parity subset
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out; rm -f gpurun_out/duo_*
for v in ${VARIANTS:-0 256}; do
  MST_TCN_SOLO=$v MST_TCN_PROF_BLOCK=5 MST_TCN_PROF_FILE=$R/gpurun_out/duo_phase_${v}_0.bin timeout 600 python bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline > /dev/null 2> gpurun_out/duo_${v}_0.err
  MST_TCN_SOLO=$v timeout 600 python bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline > gpurun_out/duo_${v}_0.json 2>> gpurun_out/duo_${v}_0.err
done
if [ -n "$WITH_TESTS" ]; then MST_TCN_SOLO=256 timeout 900 python -m pytest tests -m gpu -q -k "bf16 or golden or independent" 2>&1 | tail -5 > gpurun_out/duo_pytest.log; fi
