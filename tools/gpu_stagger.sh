#!/bin/bash
# first-generation start-delay experiment on the bf16 TCN block kernel: "S:S2" pairs (clocks across CUs : extra for 2nd WG)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out
for v in ${VARIANTS:-0:0 80000:0 80000:40000 0:40000}; do
  s=${v%%:*}; s2=${v##*:}; tag=${s}_${s2}
  MST_TCN_STAGGER=$s MST_TCN_STAGGER2=$s2 MST_TCN_PROF_BLOCK=5 MST_TCN_PROF_FILE=$R/gpurun_out/stag_$tag.bin timeout 600 python bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline > /dev/null 2> gpurun_out/stag_$tag.err
  MST_TCN_STAGGER=$s MST_TCN_STAGGER2=$s2 timeout 600 python bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline > gpurun_out/stag_$tag.json 2>> gpurun_out/stag_$tag.err
done
