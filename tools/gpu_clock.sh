#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS -d $R/gpurun_out/pmc_clk -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --precision bf16 --no-cpu-baseline > $R/gpurun_out/pmc_clk.log 2>&1
