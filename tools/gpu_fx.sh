#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python tools/bench_fx.py > gpurun_out/bench_fx.json 2> gpurun_out/bench_fx.err; echo "rc=$?" >> gpurun_out/bench_fx.err
timeout 900 python -m pytest tests -m gpu -q -k "fx" 2>&1 | tail -4 > gpurun_out/var_pytest.log
