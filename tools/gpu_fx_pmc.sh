#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the FX chain's kernels (separate --pmc passes, kernel-trace only) -> gpurun_out/fx_chain_traffic.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd)
mkdir -p gpurun_out
N=${N:-4}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fx_fetch -o pmc --output-format csv -- python $R/tools/bench_fx.py --chain-only $N > $R/gpurun_out/pmc_fx_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_fx_write -o pmc --output-format csv -- python $R/tools/bench_fx.py --chain-only $N > $R/gpurun_out/pmc_fx_write.log 2>&1
cd "$R"
FD=$(dirname $(find gpurun_out/pmc_fx_fetch -name "*counter_collection.csv" | head -1)); WD=$(dirname $(find gpurun_out/pmc_fx_write -name "*counter_collection.csv" | head -1))
for d in $FD $WD; do f=$(ls $d/*counter_collection.csv | head -1); [ "$f" != "$d/pmc_counter_collection.csv" ] && cp $f $d/pmc_counter_collection.csv; done
python tools/pmc_fx_traffic.py $FD $WD $((N + 1)) gpurun_out/fx_chain_traffic.json > gpurun_out/fx_chain_traffic.log 2>&1
tail -3 gpurun_out/fx_chain_traffic.log
