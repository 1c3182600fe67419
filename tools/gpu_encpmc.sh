#!/bin/bash
# PMC counters of the encoder's conv kernels, one small group per pass (never together with a trace domain)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out; rm -rf gpurun_out/encpmc_*
cd /tmp; export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d "$R/gpurun_out/encpmc_$i" -o pmc --output-format csv -- python "$R/bench.py" --steps 1 --warmup 1 --precision ${PREC:-bf16} --workload configs1 --no-cpu-baseline $EXTRA > "$R/gpurun_out/encpmc_$i.log" 2>&1
done <<< "${GROUPS_PMC:-SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS
SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_LDS}"
cd "$R"
python tools/pmc_dispatches.py "${PAT:-enc_conv_nlc_kernel}" ${N:-16} gpurun_out/encpmc_* > gpurun_out/encpmc_table.txt 2>&1
cat gpurun_out/encpmc_table.txt
