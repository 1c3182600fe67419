#!/bin/bash
# exercises bench.py's N>1 code path on the single-GPU box: 2 ranks sharing cuda:0, gloo for the collective
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
MST_BENCH_SHARE_GPU=1 MST_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank_gloo.err; echo "rc=$?" >> gpurun_out/bench_2rank_gloo.err
# and whether RCCL initialises at all with one rank per process on this box (world 1 through torchrun)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err; echo "rc=$?" >> gpurun_out/bench_torchrun1.err
timeout 900 python -m pytest tests -m gpu -q -k "feature_extraction or cli" 2>&1 | tail -4 > gpurun_out/var_pytest.log
