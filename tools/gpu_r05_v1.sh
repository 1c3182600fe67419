#!/bin/bash
# round 5, visit 1: GPU suite with the new 2^19 / form 53 / bit 6 tests, A/B of forms 21 | 53 (bf16) and 53 | 117 (bf16x3, bit 6),
# SQ counters of the current block kernels (two 8-counter groups per mode, own passes), and a kernel trace of 10 timed steps.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out/v1; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(rocm-smi --showproductname; lscpu | head -20; nproc) > $O/box.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=15 2>&1 > $O/pytest_gpu_full.log; tail -150 $O/pytest_gpu_full.log > $O/pytest_gpu.log
timeout 300 python tools/bench_tcn_forms.py --forms 21,53 --steps 10 --rounds 3 --out $O/tcn_forms_21_53.json > $O/tcn_forms_21_53.log 2>&1
for f in 53 117 53 117; do
  timeout 300 python bench.py --precision bf16x3 --workload configs1 --steps 4 --warmup 2 --no-cpu-baseline --tcn-tuning $f >> $O/x3_ab_53_117.jsonl 2>> $O/x3_ab.err
done
cd /tmp && export TMPDIR=/tmp
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"
for prec in bf16 bf16x3; do
  i=0
  for grp in "$G1" "$G2"; do
    i=$((i+1))
    timeout 400 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_${prec}_$i -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --precision $prec --workload configs1 --no-cpu-baseline > $O/pmc_${prec}_$i.log 2>&1
  done
done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o bench -- python $R/bench.py --steps 10 --warmup 1 --precision bf16 --workload configs1 --no-cpu-baseline > $O/prof_bf16.json 2> $O/prof_bf16.err
cd $R
python tools/pmc_dispatches.py tcn_block_bf16 13 $O/pmc_bf16_1 $O/pmc_bf16_2 > $O/r05_pmc_sq_tcn_block_bf16.txt 2>&1
python tools/pmc_dispatches.py tcn_block_bf16x3 13 $O/pmc_bf16x3_1 $O/pmc_bf16x3_2 > $O/r05_pmc_sq_tcn_block_bf16x3.txt 2>&1
python tools/rocprof_summary.py "$(find $O/prof_bf16 -name '*.db' | head -1)" "bench.py --workload configs1 --precision bf16 --steps 10 --warmup 1: the 10 TIMED steps only (first 1/11 of every kernel's dispatches dropped)" --drop-first 0.0909 > $O/r05_bench_bf16_kernel_stats_timed_steps.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/prof_bf16 $O/pmc_bf16_1 $O/pmc_bf16_2 $O/pmc_bf16x3_1 $O/pmc_bf16x3_2
tail -5 $O/pytest_gpu.log; tail -12 $O/tcn_forms_21_53.log; cat $O/x3_ab_53_117.jsonl | cut -c1-400; cat $O/prof_bf16.json | cut -c1-600
