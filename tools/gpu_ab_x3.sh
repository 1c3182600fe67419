#!/bin/bash
# same-box A/B of two builds (tools/_ab/old.so, tools/_ab/new.so) on the split-bf16 headline with per-block kernel times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/ab; mkdir -p $O
one() { python bench.py --precision bf16x3 --workload configs1 --steps 4 --warmup 2 --no-cpu-baseline 2>$O/err_x3.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); x=json.load(open('gpurun_out/bench_details.json'))['details']['headline']['roofline']
print(round(d['value'],1), round(d['ms_per_step'],3), [round(v,3) for v in x['per_block_ms']])"; }
for r in 1 2; do for v in old new; do cp tools/_ab/$v.so music_mixing_style_transfer_amd/csrc/libmst_hip.so; echo "$v bf16x3: $(one)" >> $O/ab_x3.txt; done; done
cat $O/ab_x3.txt
cp tools/_ab/new.so music_mixing_style_transfer_amd/csrc/libmst_hip.so
