#!/bin/bash
# rocprofv3 kernel stats of tools/bench_fx.py (FX chain + convolution reverb)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out/fxprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o fx -- python $R/tools/bench_fx.py > $O/prof.log 2>&1
cd $R
python tools/rocprof_summary.py "$(find $O/prof -name '*.db' | head -1)" "tools/bench_fx.py" > $O/kernel_stats.txt 2>&1
find $O -name "*.db" -delete; head -22 $O/kernel_stats.txt
