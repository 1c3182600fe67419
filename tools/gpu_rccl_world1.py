#!/usr/bin/env python
"""The collectives of the sharded engine (inference/engine.py) and of bench.py on the nccl (= RCCL) backend with ONE rank: whether RCCL initialises on the box
with a bound device and runs them at all - the only part of the N > 1 path a single-GPU box can exercise on that backend (two ranks need two GPUs; over gloo
they share one: tests/test_gpu_parity.py, tools/gpu_dist.sh).   python tools/gpu_rccl_world1.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    seen = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(seen)
    torch.cuda.synchronize()
    print(f"init + first all_reduce: {time.perf_counter() - t0:.2f} s, ranks seen {int(seen.item())}, backend {dist.get_backend()}")
    # the engine's embedding exchange: equal-sized zero-padded shards [n_pad, 2048] -> [world * n_pad, 2048]
    padded = torch.randn(152, 2048, device=dev)
    gathered = torch.empty(152, 2048, device=dev)
    dist.all_gather_into_tensor(gathered, padded)
    assert torch.equal(gathered, padded)
    ranges = [None]
    dist.all_gather_object(ranges, (0, 1212))
    assert ranges == [(0, 1212)]
    out = [torch.empty(2, 4096, device=dev)]
    dist.gather(torch.ones(2, 4096, device=dev), out, dst=0)
    assert float(out[0].sum()) == 2 * 4096
    dist.barrier()
    from music_mixing_style_transfer_amd.inference import engine
    print("engine collectives on nccl, world 1: all_gather_into_tensor, all_gather_object, gather, barrier OK;", engine.__name__)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
