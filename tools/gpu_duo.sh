#!/bin/bash
# experimental two-set persistent TCN kernel: bench + phase stamps (+ GPU parity subset); VARIANTS = "WGS:KNOB" pairs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out; rm -f gpurun_out/duo_*
for v in ${VARIANTS:-0:0 256:0 256:2}; do
  wg=${v%%:*}; kn=${v##*:}; tag=${wg}_${kn}
  MST_TCN_DUO=$wg MST_TCN_STAGGER2=$kn MST_TCN_PROF_BLOCK=5 MST_TCN_PROF_FILE=$R/gpurun_out/duo_phase_$tag.bin timeout 600 python bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline > /dev/null 2> gpurun_out/duo_$tag.err
  MST_TCN_DUO=$wg MST_TCN_STAGGER2=$kn timeout 600 python bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline > gpurun_out/duo_$tag.json 2>> gpurun_out/duo_$tag.err
done
if [ -n "$WITH_TESTS" ]; then MST_TCN_DUO=256 timeout 900 python -m pytest tests -m gpu -q -k "bf16 or golden or independent" 2>&1 | tail -5 > gpurun_out/duo_pytest.log; fi
