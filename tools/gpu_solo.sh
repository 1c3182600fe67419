#!/bin/bash
# experimental persistent TCN kernels: MST_TCN_SOLO (one set, 512-time tiles) / MST_TCN_DUO (two sets): bench + phase stamps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out; rm -f gpurun_out/duo_*
for v in ${VARIANTS:-SOLO:0 SOLO:256 DUO:256}; do
  kind=${v%%:*}; n=${v##*:}; tag=${n}_${kind}
  env MST_TCN_$kind=$n MST_TCN_PROF_BLOCK=5 MST_TCN_PROF_FILE=$R/gpurun_out/duo_phase_$tag.bin timeout 600 python bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline > /dev/null 2> gpurun_out/duo_$tag.err
  env MST_TCN_$kind=$n timeout 600 python bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline > gpurun_out/duo_$tag.json 2>> gpurun_out/duo_$tag.err
done
if [ -n "$WITH_TESTS" ]; then MST_TCN_SOLO=256 timeout 900 python -m pytest tests -m gpu -q -k "bf16 or golden or independent" 2>&1 | tail -5 > gpurun_out/duo_pytest.log; fi
