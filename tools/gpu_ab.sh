#!/bin/bash
# same-box A/B of two builds of libmst_hip.so: put them at tools/_ab/old.so and tools/_ab/new.so (untracked), then
# gpurun -- 'bash tools/gpu_ab.sh' alternates them under bench.py (split-bf16 and exact-fp32 legs; MODES="bf16:10 bf16x3:4" overrides)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
one() { python bench.py --precision $1 --workload configs1 --steps $2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
for r in 1 2; do for v in old new; do cp tools/_ab/$v.so music_mixing_style_transfer_amd/csrc/libmst_hip.so
  for m in ${MODES:-bf16x3:4 fp32:2}; do echo "$v ${m%%:*}: $(one ${m%%:*} ${m##*:})"; done; done; done
