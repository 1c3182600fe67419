#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out
for v in 0 1; do
  MST_TCN_T128=$v timeout 600 python bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err
done
MST_TCN_T128=1 timeout 900 python -m pytest tests -m gpu -q -k "bf16 or golden or independent" 2>&1 | tail -3 > gpurun_out/var_pytest.log
