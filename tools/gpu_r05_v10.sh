#!/bin/bash
# round 5, visit 10: the stereo equaliser's apply pass on slabs (default) against one lane per chunk (mst_fx_set_tuning bit 4): chain time, alternating;
# per-kernel timeline of the default; GPU tests of the FX rows
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out/v10; mkdir -p $O
for t in ${TUNINGS:-17 1 17 1}; do
  timeout 200 python tools/bench_fx.py --fx-tuning $t > $O/bench_fx_t$t.json 2>> $O/bench.err
  python -c "
import json; d=json.load(open('$O/bench_fx_t$t.json')); print('fx tuning $t: chain ms', d['ms_per_chain'], 'max dev vs oracle', d.get('max_abs_dev_vs_oracle'), {k: round(v, 4) for k, v in d.get('per_processor_ms', {}).items()})" | tee -a $O/eq_ab.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_fx -o fx -- python $R/tools/bench_fx.py --chain-only 4 > $O/prof_fx.log 2>&1
cd $R
python tools/rocprof_summary.py "$(find $O/prof_fx -name '*.db' | head -1)" "tools/bench_fx.py --chain-only 4" > $O/r05_fx_kernel_stats_slab_apply.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/prof_fx
grep -i "biquad" $O/r05_fx_kernel_stats_slab_apply.txt | cut -c1-150
if [ -z "$SKIP_TESTS" ]; then timeout 900 python -m pytest tests -m gpu -q -k "normaliz or reverb or fx or cli or fir or chain or equal or config4 or anchor" > $O/pytest_fx.log 2>&1; tail -3 $O/pytest_fx.log; fi
