#!/bin/bash
# shader clock during the FX kernels: GRBM_GUI_ACTIVE (cycles, summed over 8 XCDs) / kernel duration
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_clk_fx -o pmc --output-format csv -- python $R/tools/bench_fx.py > $R/gpurun_out/pmc_clk_fx.log 2>&1
