#!/usr/bin/env python
"""bench.py - headline benchmark: stereo 44.1 kHz segments/sec, FXencoder + MixFXcloner forward.

Workload (BASELINE.json configs[1]): batch = 32 segments of 2 x 131072 samples per GPU, default configs.yaml
architectures, deterministic synthetic weights, synthetic audio; one step = FXencoder on the 32 reference
segments -> (all-gather of segment embeddings when N > 1) -> mean -> FiLM factors -> TCN on the 32 input
segments.  Inputs are resident in HBM before the timed region.  Weak scaling: every rank owns 32 segments.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision bf16|fp32] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel = the dilated 128x128x15 TCN block
conv, timed with HIP events on its stream inside the timed region) and "cpu_baseline" (the oracle's torch-CPU
restatement on a bounded sample, rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

SEG_LEN = 131072
BATCH = 32
TCN_FLOP_PER_SAMPLE_BLOCK = 2 * 128 * 128 * 15          # one dense TCN block, per output time step
PEAK = {"bf16": 2500.0, "fp32": 157.3}                  # dense MFMA TFLOP/s, MI355X_MICROARCH.md


def load_cfg():
    import yaml
    with open(os.path.join(REPO, "music_mixing_style_transfer_amd", "networks", "configs.yaml")) as f:
        c = yaml.full_load(f)
    return c["Effects_Encoder"]["default"], c["TCN"]["default"]


def cpu_baseline(enc_cfg, enc_sd, tcn_sd, seconds_budget=25.0):
    """Oracle (torch-CPU fp32 restatement of the reference) on a bounded sample of the same workload."""
    from oracle import networks_ref as R
    from music_mixing_style_transfer_amd.utils import synth
    cfg = dict(enc_cfg)
    x = synth.synth_audio((1, 2, SEG_LEN), seed=1)
    n, t0 = 0, time.time()
    while True:
        emb = R.fxencoder_forward(enc_sd, cfg, x)
        R.tcn_forward(tcn_sd, x, emb.mean(0, keepdim=True))
        n += 1
        if time.time() - t0 > seconds_budget * 0.5 or n >= 4:
            break
    dt = time.time() - t0
    return {"value": n / dt, "unit": "segments/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} segment(s) of 2x{SEG_LEN}, FXencoder+TCN fp32, oracle/networks_ref.py (torch-CPU), batch 1"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    import torch.distributed as dist
    if os.environ.get("MST_BENCH_SHARE_GPU"):      # test hook: several ranks on one GPU (with MST_DIST_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("MST_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from music_mixing_style_transfer_amd import _lib
    from music_mixing_style_transfer_amd.inference import StyleTransferEngine, build_models
    from music_mixing_style_transfer_amd.utils import synth

    enc_cfg, tcn_cfg = load_cfg()
    enc_sd = synth.fxencoder_state_dict(enc_cfg, seed=0)
    tcn_sd = synth.tcn_state_dict(seed=0)
    enc, tcn = build_models({k: (list(v) if isinstance(v, list) else v) for k, v in enc_cfg.items()}, tcn_cfg, dev,
                            precision=args.precision)
    enc.load_state_dict(enc_sd)
    tcn.load_state_dict(tcn_sd)
    engine = StyleTransferEngine(enc, tcn)

    B = args.batch
    ref = synth.synth_audio((B, 2, SEG_LEN), seed=100 + rank).to(dev)      # resident in HBM before timing
    inp = synth.synth_audio((B, 2, SEG_LEN), seed=200 + rank).to(dev)

    lib = _lib.lib()
    enc._get_runner()._ensure(lib)      # weight folding / packing is setup, not part of a step
    tcn._ensure(lib)
    for _ in range(args.warmup):
        engine.step(ref, inp)
    torch.cuda.synchronize()

    lib.check(lib.mst_tcn_timing_begin(tcn._handle, args.steps), "timing_begin")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y, _ = engine.step(ref, inp)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    nb = tcn.hparams.nblocks
    ms = (C.c_float * (nb + 1))()
    nf = C.c_int(0)
    lib.check(lib.mst_tcn_timing_end(tcn._handle, ms, C.byref(nf)), "timing_end")

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        traffic = None      # HBM bytes per launch of the dominant kernel: offline rocprofv3 --pmc passes (tools/pmc_traffic.py)
        tpath = os.path.join(REPO, "profiles", "r01_tcn_block_bf16_traffic.json")
        if args.precision == "bf16" and B == BATCH and os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get("traffic_bytes")
        block_ms = [float(v) for v in ms]
        dense = block_ms[1:nb]                                   # the 13 dilated 128->128 blocks
        avg_ms = sum(dense) / len(dense)
        flop_per_launch = TCN_FLOP_PER_SAMPLE_BLOCK * B * SEG_LEN
        achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12
        out = {
            "metric": "stereo 44.1 kHz segments/sec (FXencoder+MixFXcloner fwd)",
            "value": world * B * args.steps / dt,
            "unit": "segments/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision if args.precision == "bf16" else "f32",
            "data": "synthetic",
            "config": {"workload": f"configs[1]: batch={B} segments of 2x{SEG_LEN} per GPU, FXencoder+MixFXcloner forward, "
                                   f"default configs.yaml nets, synthetic weights; TCN dense blocks {args.precision} MFMA "
                                   f"(fp32 accumulate), FXencoder convs {args.precision} MFMA",
                       "segments_per_gpu": B, "segment_length": SEG_LEN,
                       "parallelism": f"segment-sharded x{world}, all-gather of embeddings"},
            "roofline": {"kernel": "tcn_block_%s_kernel (dilated 128x128x15 conv + fused BN/LeakyReLU/FiLM/residual)" %
                                   ("bf16" if args.precision == "bf16" else "f32"),
                         "bound": "mfma", "achieved": achieved, "peak": PEAK[args.precision], "unit": "TFLOP/s",
                         "frac": achieved / PEAK[args.precision], "traffic": traffic,
                         "avg_launch_ms": avg_ms, "launches_per_step": nb - 1, "timed_forwards": int(nf.value),
                         "flop_per_launch": flop_per_launch, "per_block_ms": block_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(enc_cfg, enc_sd, tcn_sd)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
