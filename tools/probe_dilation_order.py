#!/usr/bin/env python
"""Is the 2-3 % that the d = 4 ... 16 blocks run slower a property of the DILATION or of the DATA at that depth?  Per-block kernel times
of the default MixFXcloner at 32 x 2x131072 (bf16) with the blocks' dilations in their own order (2, 4, ... 8192) and REVERSED (8192 ...
2; same weights, same input): if the slow launches move with the dilation it is the tile geometry, if they stay at depth 2-4 it is what
the operands look like there (the chip is power-limited: the rate depends on the operand bits).

    python tools/probe_dilation_order.py [--steps 10]
"""
import argparse
import ctypes as C
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
SEG = 131072


def build(tcn_cfg, sd, dev, order):
    from music_mixing_style_transfer_amd.networks import TCNModel
    tcn = TCNModel(nparams=tcn_cfg["condition_dimension"], ninputs=2, noutputs=2, nblocks=tcn_cfg["nblocks"],
                   dilation_growth=tcn_cfg["dilation_growth"], kernel_size=tcn_cfg["kernel_size"], channel_width=tcn_cfg["channel_width"],
                   stack_size=tcn_cfg["stack_size"], cond_dim=tcn_cfg["condition_dimension"], causal=tcn_cfg["causal"]).to(dev)
    tcn.load_state_dict(sd)
    tcn.precision = "bf16"
    nb = tcn_cfg["nblocks"]
    if order == "reversed":
        for n in range(1, nb):
            tcn.blocks[n].dilation = 2 ** (nb - n)          # 8192 ... 2 (block 0 keeps d = 1: its own kernel)
    return tcn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import yaml
    from music_mixing_style_transfer_amd import _lib
    from music_mixing_style_transfer_amd.utils import synth
    dev = torch.device("cuda", 0)
    with open(os.path.join(REPO, "music_mixing_style_transfer_amd", "networks", "configs.yaml")) as f:
        tcn_cfg = yaml.full_load(f)["TCN"]["default"]
    sd = synth.tcn_state_dict(seed=0)
    lib = _lib.lib()
    x = synth.synth_audio((32, 2, SEG), seed=200).to(dev)
    cond = synth.synth_audio((1, tcn_cfg["condition_dimension"]), seed=3).to(dev)
    nets = {o: build(tcn_cfg, sd, dev, o) for o in ("normal", "reversed")}
    for t in nets.values():
        t._ensure(lib)
    nb = tcn_cfg["nblocks"]
    for r in range(2):
        for o, tcn in nets.items():
            for _ in range(2):
                tcn(x, cond)
            torch.cuda.synchronize()
            lib.check(lib.mst_tcn_timing_begin(tcn._handle, args.steps), "timing_begin")
            for _ in range(args.steps):
                tcn(x, cond)
            torch.cuda.synchronize()
            ms = (C.c_float * (nb + 1))()
            nf = C.c_int(0)
            lib.check(lib.mst_tcn_timing_end(tcn._handle, ms, C.byref(nf)), "timing_end")
            d = [int(tcn.blocks[n].dilation) for n in range(nb)]
            print(f"round {r} {o:8s}: " + " ".join(f"d{d[n]}:{ms[n]:.3f}" for n in range(nb)), flush=True)
        # mean |activation| behind blocks 1..nb-1 of the normal order (what the next block's matrix cores multiply)
    tcn = nets["normal"]
    for n in (1, 2, 3, 4, 6, 9, 12):
        a = tcn.forward_blocks(x[:2], cond, n)
        print(f"normal order, behind block {n}: mean|a| {float(a.abs().mean()):.3f} max|a| {float(a.abs().max()):.2f} "
              f"fraction |a| < 2^-6: {float((a.abs() < 2 ** -6).float().mean()):.3f}", flush=True)


if __name__ == "__main__":
    main()
