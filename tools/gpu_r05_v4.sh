#!/bin/bash
# round 5, visit 4: stereo slab kernels of the equaliser (ends / apply through LDS) - FX GPU tests, configs[3] chain, kernel timeline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out/v4; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -k "fx or compressor or equaliser or chain or config4 or time_parallel or normaliz or stem_sized" > $O/pytest_fx_full.log 2>&1; tail -5 $O/pytest_fx_full.log
for t in 1 9 1 9; do   # 1 = the default (three slices), 9 = four
  timeout 300 python tools/bench_fx.py --fx-tuning $t > $O/bench_fx_t$t.json 2>> $O/bench_fx.err
  python -c "
import json; d=json.load(open('$O/bench_fx_t$t.json')); print('fx tuning $t: chain ms', round(d['ms_per_chain'],4), 'dev', d['max_abs_dev_vs_oracle'], {k: round(v,3) for k,v in d['per_processor_ms'].items()})" | tee -a $O/fx_ab.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_fx -o fx -- python $R/tools/bench_fx.py --fx-tuning 1 --chain-only 3 > $O/prof_fx.log 2>&1
cd $R
python tools/rocprof_trace.py "$(find $O/prof_fx -name '*.db' | head -1)" "fx_" 20 > $O/r05_fx_chain_timeline.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/prof_fx
cat $O/r05_fx_chain_timeline.txt
