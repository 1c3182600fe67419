#!/bin/bash
# N = 1, 2, 4, 8 back to back on ONE node (python bench.py --gpus N starts its own N ranks over RCCL): the weak-scaling value (configs[1] per GPU) and the
# strong-scaling efficiency of the 60-minute stem (track60.efficiency_t1_over_n_tn) per N.  No curve has been measured yet: the pool's boxes have one GPU.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; O=gpurun_out/scale; mkdir -p $O; export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in ${NS:-1 2 4 8}; do
  timeout 1800 python bench.py --gpus $n --steps ${STEPS:-20} --warmup ${WARMUP:-5} > $O/bench_n$n.json 2> $O/bench_n$n.err || echo "N=$n: rc=$?"
done
python - <<'PY'
import glob, json, re
rows = sorted((json.loads(open(f).read().strip().splitlines()[-1]) for f in glob.glob("gpurun_out/scale/bench_n*.json") if open(f).read().strip()), key=lambda d: d["n_gpus"])
v1 = next((d["value"] for d in rows if d["n_gpus"] == 1), None)
print("N  ranks_seen  segments/s  per-GPU  weak-eff  track60 seg/s  strong-eff(t1/(N tN))")
for d in rows:
    t = d.get("track60", {})
    print(d["n_gpus"], d.get("ranks_seen"), round(d["value"], 1), round(d["value"] / d["n_gpus"], 1), round(d["value"] / d["n_gpus"] / v1, 3) if v1 else None,
          t.get("value"), t.get("efficiency_t1_over_n_tn"))
PY
