#!/bin/bash
# round 5, visit 9: the fused encoder kernels (stereo block: mst_enc_set_schedule bit 3 selects the two direct launches; block 1: bit 4 the two conv launches):
# whole-step A/B (alternating), per-launch encoder timeline of the default, GPU encoder tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out/v9; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for sch in ${SCHEDS:-17 1 17 1}; do
  timeout 300 python bench.py --precision bf16 --workload configs1 --steps 10 --warmup 3 --no-cpu-baseline --enc-schedule $sch > $O/bench_sch$sch.json 2>> $O/bench.err
  python -c "
import json; d=json.load(open('$O/bench_sch$sch.json')); print('enc schedule $sch: segments/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'feature_extraction', d.get('feature_extraction'))" | tee -a $O/enc_ab.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_enc -o enc -- python $R/bench.py --steps 2 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $O/prof_enc.log 2>&1
cd $R
python tools/rocprof_trace.py "$(find $O/prof_enc -name '*.db' | head -1)" "enc_" 52 > $O/r05_enc_timeline_bf16_fused_stereo.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/prof_enc
tail -40 $O/r05_enc_timeline_bf16_fused_stereo.txt | cut -c1-150
if [ -z "$SKIP_TESTS" ]; then timeout 900 python -m pytest tests -m gpu -q -k "encoder or headline or standalone or feature" > $O/pytest_enc.log 2>&1; tail -4 $O/pytest_enc.log; fi
