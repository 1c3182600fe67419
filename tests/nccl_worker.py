"""Started by tests/test_multi_gpu.py as `python -m torch.distributed.run --nproc-per-node N tests/nccl_worker.py` on a box with N >= 2 GPUs:
one process per GPU, RCCL ("nccl") for the engine's one collective.  A synthetic stem pair is converted with its segments sharded over the
N ranks (StyleTransferEngine.transfer_stem: per-rank shard, all-gather of segment embeddings, canonical-order mean), the ranks' time ranges are
gathered on rank 0, and rank 0 converts the same pair alone: the two results must be BIT-equal (the mean does not depend on N, segments are
independent).  Rank 0 prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    import yaml
    from music_mixing_style_transfer_amd.inference import StyleTransferEngine, build_models
    from music_mixing_style_transfer_amd.inference import segmentation as seg
    from music_mixing_style_transfer_amd.utils import synth
    with open(os.path.join(REPO, "music_mixing_style_transfer_amd", "networks", "configs.yaml")) as f:
        cfgs = yaml.full_load(f)
    enc_cfg, tcn_cfg = cfgs["Effects_Encoder"]["default"], cfgs["TCN"]["default"]
    precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    enc, tcn = build_models({k: (list(v) if isinstance(v, list) else v) for k, v in enc_cfg.items()}, tcn_cfg, dev, precision=precision)
    enc.load_state_dict(synth.fxencoder_state_dict(enc_cfg, seed=0))
    tcn.load_state_dict(synth.tcn_state_dict(seed=0))
    SEG, L_IN, L_REF = 131072, 21 * 131072 - 777, 23 * 131072 + 5          # 21 input / 24 reference segments: uneven shards at N = 2, 4, 8
    g = torch.Generator(device=dev).manual_seed(99)                       # the same stems on every rank
    x_in = torch.rand(2, L_IN, generator=g, device=dev) - 0.5
    x_ref = torch.rand(2, L_REF, generator=g, device=dev) - 0.5
    eng = StyleTransferEngine(enc, tcn)
    assert eng.world == world and eng.rank == rank
    res = eng.transfer_stem(x_in, x_ref, SEG, SEG)
    y, rng = res if world > 1 else (res, (0, L_IN))          # one rank (a smoke run of this script): the whole stem
    ranges = [None] * world
    dist.all_gather_object(ranges, tuple(int(v) for v in rng))
    full = eng.gather_stem(y, rng, L_IN)
    out = None
    if rank == 0:
        solo = StyleTransferEngine(enc, tcn)
        solo.dist, solo.world, solo.rank = None, 1, 0
        alone = solo.transfer_stem(x_in, x_ref, SEG, SEG)
        torch.cuda.synchronize()
        n_in = seg.segment_count(L_IN, SEG)
        want = [(min(L_IN, seg.shard_range(n_in, r, world)[0] * SEG), min(L_IN, seg.shard_range(n_in, r, world)[1] * SEG)) for r in range(world)]
        out = {"world": world, "precision": precision, "bit_equal": bool(torch.equal(full, alone)), "ranges": ranges, "ranges_expected": want,
               "shape": list(full.shape), "finite": bool(torch.isfinite(full).all()),
               "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()), "devices": torch.cuda.device_count()}
    dist.barrier()
    dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
