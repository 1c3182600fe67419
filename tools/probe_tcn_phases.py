#!/usr/bin/env python
"""Phase clocks of the dominant kernel (tcn_block_bf16_duo_kernel<4>, the d = 64 block of the default TCN at 32 x 131072; --kernel x3: the
split-bf16 tcn_block_bf16x3_kernel<2, 4>): a PROBE BUILD of the library (tools/_ab/probe.so, built by tools/build_probe.py: csrc/ with s_memtime
stamps patched into one wave of one workgroup; tools/_ab is untracked) accumulates the shader clocks that wave spends per tile in: [0] loop bookkeeping, [1] accumulator init + ring preload +
classes 0 .. 2, [2] the last class, [3] issuing the residual reads, [4] barrier 1, [5] epilogue arithmetic + LDS writes, [6] barrier 2; [7] tiles.
    python tools/probe_tcn_phases.py [--forwards 5]"""
import argparse
import ctypes as C
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--forwards", type=int, default=5)
    ap.add_argument("--kernel", default="duo", choices=["duo", "x3", "enc"])
    ap.add_argument("--lib", default=os.path.join(REPO, "tools", "_ab", "probe.so"))
    args = ap.parse_args()
    import yaml
    from music_mixing_style_transfer_amd import _lib
    b = _lib.bind(args.lib)
    _lib.set_default_binding(b)
    from music_mixing_style_transfer_amd.networks import TCNModel
    from music_mixing_style_transfer_amd.utils import synth
    dev = torch.device("cuda", 0)
    with open(os.path.join(REPO, "music_mixing_style_transfer_amd", "networks", "configs.yaml")) as f:
        cfg = yaml.full_load(f)["TCN"]["default"]
    tcn = TCNModel(nparams=cfg["condition_dimension"], ninputs=2, noutputs=2, nblocks=cfg["nblocks"], dilation_growth=cfg["dilation_growth"],
                   kernel_size=cfg["kernel_size"], channel_width=cfg["channel_width"], stack_size=cfg["stack_size"],
                   cond_dim=cfg["condition_dimension"], causal=cfg["causal"]).to(dev)
    tcn.load_state_dict(synth.tcn_state_dict(seed=0))
    tcn.precision = "bf16x3" if args.kernel == "x3" else "bf16"
    if args.kernel == "enc":          # the FXencoder's channel-minor conv kernel on the 2048 -> 2048 layers (32 x 131072, bf16)
        from music_mixing_style_transfer_amd.networks import FXencoder
        with open(os.path.join(REPO, "music_mixing_style_transfer_amd", "networks", "configs.yaml")) as f:
            ecfg = yaml.full_load(f)["Effects_Encoder"]["default"]
        enc = FXencoder({k: (list(v) if isinstance(v, list) else v) for k, v in ecfg.items()}).to(dev)
        enc.load_state_dict(synth.fxencoder_state_dict(ecfg, seed=0))
        enc.precision = "bf16"
        run = lambda: enc(x)
    else:
        run = lambda: tcn(x, cond)
    x = synth.synth_audio((32, 2, 131072), seed=200).to(dev)
    cond = synth.synth_audio((1, cfg["condition_dimension"]), seed=3).to(dev)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    out = (C.c_longlong * 32)()
    rd = b.cdll.mst_probe_read
    rd.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    assert rd(out, 1) == 0
    for _ in range(args.forwards):
        run()
    torch.cuda.synchronize()
    assert rd(out, 0) == 0
    v = [int(t) for t in out]
    names = ["loop bookkeeping", "acc init + ring preload + classes 0..2", "last class", "residual reads issued", "barrier 1", "epilogue arithmetic + LDS writes", "barrier 2"]
    if args.kernel == "x3":          # tcn_block_bf16x3_kernel<2, 4>: one 128-time tile per workgroup, two workgroups per CU
        v = v[8:]
        names = ["staging (loads, hi / lo split, LDS writes)", "barrier", "main loop (three MFMAs per product)", "barrier", "LeakyReLU / FiLM + transposed LDS writes", "barrier",
                 "rows: residual from global + store"]
    if args.kernel == "enc":
        v = [int(t) for t in out][16:24]
        names = ["prologue (descriptors, first fetch)", "barrier (previous tile consumed)", "wait for the chunk's loads + LDS writes", "barrier (tile complete)",
                 "next chunk's loads issued", "16 MFMAs + 16 ds_read_b128"]
        v = v[:6] + [0, v[7]]
    tiles = max(1, v[7])
    tot = sum(v[:7])
    print(f"tiles seen by the probed wave: {tiles} ({args.forwards} forwards); clocks per tile: {tot / tiles:.0f}")
    for n, t in zip(names, v[:7]):
        print(f"  {n:42s} {t / tiles:9.0f} clocks per tile  {100.0 * t / tot:5.1f} %")


if __name__ == "__main__":
    main()
