#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out
for v in 0 1 0 1; do
  MST_TCN_XCD=$v timeout 600 python bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline >> gpurun_out/xcd_$v.json 2> gpurun_out/xcd_$v.err
done
MST_TCN_XCD=1 timeout 900 python -m pytest tests -m gpu -q -k "bf16 or golden or independent" 2>&1 | tail -3 > gpurun_out/xcd_pytest.log
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
MST_TCN_XCD=$v timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_xcd$v -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline > $R/gpurun_out/pmc_fetch_xcd$v.log 2>&1
done
