#!/bin/bash
# round 6: FETCH_SIZE / WRITE_SIZE passes of the dominant kernel alone (fingerprinted) -> gpurun_out/r06/r06_tcn_block_bf16_traffic.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $O/pmc_write.log 2>&1
cd $R
FD=$(dirname $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1)); WD=$(dirname $(find $O/pmc_write -name "*counter_collection.csv" | head -1))
for d in $FD $WD; do f=$(ls $d/*counter_collection.csv | head -1); [ "$f" != "$d/pmc_counter_collection.csv" ] && cp $f $d/pmc_counter_collection.csv; done
python tools/pmc_traffic.py $FD $WD "tcn_block_bf16_kernel<4, false, 8, 2," $O/r06_tcn_block_bf16_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
