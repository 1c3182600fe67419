#!/bin/bash
# round 5, visit 3: FX chain A/B over the number of compressor time slices (mst_fx_set_tuning), the EQ state pass with four lanes per chunk,
# and a per-launch timeline of the FXencoder (bf16) inside the bench step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out/v3; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests -m gpu -q -k "fx or compressor or equaliser or chain or config4 or time_parallel" > $O/pytest_fx_full.log 2>&1; tail -5 $O/pytest_fx_full.log
for t in 0 1 5 9 13 0 1 5 9 13; do
  timeout 300 python tools/bench_fx.py --fx-tuning $t > $O/bench_fx_t$t.json 2>> $O/bench_fx.err
  python -c "
import json; d=json.load(open('$O/bench_fx_t$t.json')); print('fx tuning $t: chain ms', round(d['ms_per_chain'],4), 'dev', d['max_abs_dev_vs_oracle'], {k: round(v,3) for k,v in d['per_processor_ms'].items()})" | tee -a $O/fx_ab.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_fx -o fx -- python $R/tools/bench_fx.py --fx-tuning 5 --chain-only 3 > $O/prof_fx.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_enc -o enc -- python $R/bench.py --steps 2 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $O/prof_enc.log 2>&1
cd $R
python tools/rocprof_trace.py "$(find $O/prof_fx -name '*.db' | head -1)" "fx_" 30 > $O/r05_fx_chain_timeline_2slices.txt 2>&1
python tools/rocprof_trace.py "$(find $O/prof_enc -name '*.db' | head -1)" "enc_" 53 > $O/r05_enc_timeline_bf16.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/prof_fx $O/prof_enc
cat $O/r05_fx_chain_timeline_2slices.txt | tail -24; cat $O/r05_enc_timeline_bf16.txt
