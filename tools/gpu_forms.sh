#!/bin/bash
# per-block kernel times of the bf16 headline under several mst_tcn_set_tuning values (same box, alternating)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/forms; mkdir -p $O
one() { python bench.py --precision bf16 --workload configs1 --steps 10 --warmup 2 --no-cpu-baseline --tcn-tuning $1 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); x=json.load(open('gpurun_out/bench_details.json'))['details']['headline']['roofline']
print(round(d['value'],1), round(d['ms_per_step'],3), d['roofline'].get('calib_ms'), [round(v,3) for v in x['per_block_ms']])"; }
for r in 1 2; do for f in ${FORMS:-5 7 1}; do echo "tuning $f: $(one $f)" >> $O/forms.txt; done; done
cat $O/forms.txt
