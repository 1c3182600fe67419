#!/bin/bash
# round 6: SQ counters of the bf16 block kernels in their round-6 default form (128-time class-major tiles, three workgroups per CU), two
# 8-counter groups in their own passes.   -> gpurun_out/r06/r06_pmc_sq_tcn_block_bf16.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"
i=0
for grp in "$G1" "$G2"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_sq_$i -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $O/pmc_sq_$i.log 2>&1
done
cd $R
python tools/pmc_dispatches.py tcn_block_bf16 13 $O/pmc_sq_1 $O/pmc_sq_2 > $O/r06_pmc_sq_tcn_block_bf16.txt 2>&1
rm -rf $O/pmc_sq_1 $O/pmc_sq_2
cat $O/r06_pmc_sq_tcn_block_bf16.txt
