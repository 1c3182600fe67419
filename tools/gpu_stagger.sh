#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out
for v in 0 1; do
  MST_TCN_PRIO=$v MST_TCN_PROF_BLOCK=5 MST_TCN_PROF_FILE=$R/gpurun_out/phase6_v$v.bin timeout 600 python bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline > /dev/null 2> gpurun_out/phase_v$v.err
  MST_TCN_PRIO=$v timeout 600 python bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err
done
