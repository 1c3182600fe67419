#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); mkdir -p gpurun_out
for v in 0 2; do
MST_TCN_BF16_VARIANT=$v MST_TCN_PROF_BLOCK=5 MST_TCN_PROF_FILE=$R/gpurun_out/phase_v$v.bin timeout 600 python bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline > gpurun_out/phase_v$v.json 2> gpurun_out/phase_v$v.err
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/gpurun_out/pmc1 -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --precision bf16 --no-cpu-baseline > $R/gpurun_out/pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES -d $R/gpurun_out/pmc2 -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --precision bf16 --no-cpu-baseline > $R/gpurun_out/pmc2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc3 -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --precision bf16 --no-cpu-baseline > $R/gpurun_out/pmc3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc4 -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --precision bf16 --no-cpu-baseline > $R/gpurun_out/pmc4.log 2>&1
cd $R; ls -la gpurun_out/pmc1 gpurun_out/pmc2 | head -20
