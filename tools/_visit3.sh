cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out/r04_v3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_nz -o nz -- python $R/tools/prof_normalizer_kernels.py 3 > $O/prof_nz.log 2>&1
cd $R
DB=$(find $O/prof_nz -name "*.db" | head -1); python tools/rocprof_summary.py "$DB" "normaliser: 2 stems x (1 cold + 3 warm)" > $O/prof_nz_kernel_stats.txt 2>&1
find $O -name "*.db" -size +20M -delete
cat $O/prof_nz.log | tail -5; head -45 $O/prof_nz_kernel_stats.txt
