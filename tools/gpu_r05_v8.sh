#!/bin/bash
# round 5, visit 8: HBM traffic of the dominant kernel and of the FX chain on the CURRENT kernel sources (separate FETCH_SIZE / WRITE_SIZE passes,
# kernel-trace only) -> gpurun_out/v8/r05_tcn_block_bf16_traffic.json, r05_fx_chain_traffic.json (committed under profiles/, fingerprinted)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out/v8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $O/pmc_write.log 2>&1
cd $R
FD=$(dirname $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1)); WD=$(dirname $(find $O/pmc_write -name "*counter_collection.csv" | head -1))
for d in $FD $WD; do f=$(ls $d/*counter_collection.csv | head -1); [ "$f" != "$d/pmc_counter_collection.csv" ] && cp $f $d/pmc_counter_collection.csv; done
python tools/pmc_traffic.py $FD $WD tcn_block_bf16_duo_kernel $O/r05_tcn_block_bf16_traffic.json > $O/pmc_traffic.log 2>&1
N=4 bash tools/gpu_fx_pmc.sh > $O/fx_pmc.log 2>&1; cp gpurun_out/fx_chain_traffic.json $O/r05_fx_chain_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write gpurun_out/pmc_fx_fetch gpurun_out/pmc_fx_write
cat $O/pmc_traffic.log | tail -2; tail -3 $O/fx_pmc.log
