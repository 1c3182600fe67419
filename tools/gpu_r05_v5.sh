#!/bin/bash
# round 5, visit 5: full GPU suite after the pruning (stream kernel, split-bf16 duo kernel gone), the in-kernel split-K finalize of the
# encoder (schedule 9 vs 1, alternating), the standard anchors and the long-FFT tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out/v5; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=12 > $O/pytest_gpu_full.log 2>&1; tail -30 $O/pytest_gpu_full.log > $O/pytest_gpu.log
grep -n "2^19\|fir_causal L\|class-major" $O/pytest_gpu_full.log | head -12
for sch in 1 9 1 9; do
  timeout 300 python bench.py --precision bf16 --workload configs1 --steps 10 --warmup 3 --no-cpu-baseline --enc-schedule $sch > $O/bench_sch$sch.json 2>> $O/bench.err
  python -c "
import json; d=json.load(open('$O/bench_sch$sch.json')); print('enc schedule $sch: segments/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'avg launch', d['roofline']['avg_launch_ms'])" | tee -a $O/enc_ab.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_enc -o enc -- python $R/bench.py --steps 2 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $O/prof_enc.log 2>&1
cd $R
python tools/rocprof_trace.py "$(find $O/prof_enc -name '*.db' | head -1)" "enc_" 40 > $O/r05_enc_timeline_bf16_inkernel_finalize.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/prof_enc
tail -4 $O/pytest_gpu.log; tail -42 $O/r05_enc_timeline_bf16_inkernel_finalize.txt
