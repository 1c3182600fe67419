#!/bin/bash
# same-box A/B of two builds (tools/_ab/old.so, new.so) on the encoder alone (bench.py's feature_extraction leg) and the whole bf16 step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
one() { python bench.py --precision bf16 --workload configs1 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('step', round(d['value'],1), 'seg/s', round(d['ms_per_step'],3), 'ms; encoder', d['feature_extraction']['ms'], 'ms')"; }
for r in 1 2; do for v in old new; do cp tools/_ab/$v.so music_mixing_style_transfer_amd/csrc/libmst_hip.so; echo "$v: $(one)"; done; done
