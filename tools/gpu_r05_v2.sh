#!/bin/bash
# round 5, visit 2: the FX changes (compressor time slices on a side stream, dot-product EQ state pass) - GPU tests of the FX rows, A/B of
# mst_fx_set_tuning 0 | 1 on configs[3], a kernel timeline of two chains (do the slices overlap?), and the form-21 test with its printout
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out/v2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -rA -k "fx or compressor or equaliser or chain or config4 or forms or normaliz or reverb or cli or time_parallel or haas" > $O/pytest_fx_full.log 2>&1; tail -40 $O/pytest_fx_full.log > $O/pytest_fx.log
grep -n "class-major" $O/pytest_fx_full.log | head -5
for t in 0 1 0 1; do
  timeout 300 python tools/bench_fx.py --fx-tuning $t > $O/bench_fx_t$t.json 2>> $O/bench_fx.err
  python -c "
import json; d=json.load(open('$O/bench_fx_t$t.json')); print('fx tuning $t: chain ms', round(d['ms_per_chain'],4), 'dev', d['max_abs_dev_vs_oracle'], d['per_processor_ms'])" | tee -a $O/fx_ab.txt
done
cd /tmp && export TMPDIR=/tmp
for t in 0 1; do
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_fx_t$t -o fx -- python $R/tools/bench_fx.py --fx-tuning $t --chain-only 3 > $O/prof_fx_t$t.log 2>&1
cd $R
python tools/rocprof_trace.py "$(find $O/prof_fx_t$t -name '*.db' | head -1)" "fx_" 34 > $O/r05_fx_chain_timeline_tuning$t.txt 2>&1
python tools/rocprof_summary.py "$(find $O/prof_fx_t$t -name '*.db' | head -1)" "tools/bench_fx.py --chain-only 3 --fx-tuning $t (1 warm-up chain + 3)" > $O/r05_fx_kernel_stats_tuning$t.txt 2>&1
cd /tmp
done
cd $R
find $O -name "*.db" -delete; rm -rf $O/prof_fx_t0 $O/prof_fx_t1
tail -5 $O/pytest_fx.log; cat $O/r05_fx_chain_timeline_tuning1.txt | tail -20
