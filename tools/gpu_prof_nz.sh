#!/bin/bash
# rocprofv3 kernel trace of the device-resident input normaliser on two 3-minute stems (tools/prof_normalizer_kernels.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out/nz; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_nz -o nz -- python $R/tools/prof_normalizer_kernels.py 3 > $O/prof_nz.log 2>&1
cd $R
python tools/rocprof_summary.py "$(find $O/prof_nz -name '*.db' | head -1)" "normaliser: 2 stems x (1 cold + 3 warm)" > $O/prof_nz_kernel_stats.txt 2>&1
find $O -name "*.db" -delete; grep "ms per warm" $O/prof_nz.log; head -14 $O/prof_nz_kernel_stats.txt
