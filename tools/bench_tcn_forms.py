#!/usr/bin/env python
"""A/B of the bf16 TCN block kernel forms on the MI355X (mst_tcn_set_tuning flags 1 / 5 / 21 / 53): the one-tile-per-workgroup kernel
against the persistent LDS-DMA-fed "duo" kernel (5 tap-major, 21 class-major, 53 = 21 + block 0 inside block 1's launch: the default's bf16 part).  Per-block kernel times from HIP events on the launch stream (mst_tcn_timing_*), the forms
alternating so that both see the same clock / thermal state; parity of the two forms against each other at full size and against
the oracle on a short segment.

    python tools/bench_tcn_forms.py [--batch 32] [--steps 10] [--rounds 3] [--out gpurun_out/tcn_forms.json]
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
SEG = 131072
FLOP = 2 * 128 * 128 * 15


def timed(lib, tcn, x, cond, steps):
    nb = tcn.hparams.nblocks
    for _ in range(2):
        tcn(x, cond)
    torch.cuda.synchronize()
    lib.check(lib.mst_tcn_timing_begin(tcn._handle, steps), "timing_begin")
    for _ in range(steps):
        tcn(x, cond)
    torch.cuda.synchronize()
    ms = (C.c_float * (nb + 1))()
    nf = C.c_int(0)
    lib.check(lib.mst_tcn_timing_end(tcn._handle, ms, C.byref(nf)), "timing_end")
    return [float(v) for v in ms]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--forms", default="1,5,21,53")
    ap.add_argument("--segment", type=int, default=131072, help="segment length (the reference's default is 2^19: --segment 524288 --batch 8)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    global SEG
    SEG = args.segment
    import yaml
    from music_mixing_style_transfer_amd import _lib
    from music_mixing_style_transfer_amd.networks import TCNModel
    from music_mixing_style_transfer_amd.utils import synth
    from oracle import networks_ref as R

    dev = torch.device("cuda", 0)
    with open(os.path.join(REPO, "music_mixing_style_transfer_amd", "networks", "configs.yaml")) as f:
        tcn_cfg = yaml.full_load(f)["TCN"]["default"]
    sd = synth.tcn_state_dict(seed=0)
    tcn = TCNModel(nparams=tcn_cfg["condition_dimension"], ninputs=2, noutputs=2, nblocks=tcn_cfg["nblocks"],
                   dilation_growth=tcn_cfg["dilation_growth"], kernel_size=tcn_cfg["kernel_size"], channel_width=tcn_cfg["channel_width"],
                   stack_size=tcn_cfg["stack_size"], cond_dim=tcn_cfg["condition_dimension"], causal=tcn_cfg["causal"]).to(dev)
    tcn.load_state_dict(sd)
    tcn.precision = "bf16"
    lib = _lib.lib()
    tcn._ensure(lib)
    nb = tcn.hparams.nblocks
    forms = [int(v) for v in args.forms.split(",")]
    out = {"batch": args.batch, "segment": SEG, "steps": args.steps, "forms": {}}

    # parity on a short segment against the oracle, every form
    xs, cs = synth.synth_audio((2, 2, 16384), seed=6), synth.synth_audio((1, tcn_cfg["condition_dimension"]), seed=7)
    y_ref = R.tcn_forward(sd, xs, cs)
    for f in forms:
        lib.check(lib.mst_tcn_set_tuning(tcn._handle, f), "tuning")
        y = tcn(xs.to(dev), cs.to(dev)).cpu()
        out["forms"][str(f)] = {"max_abs_vs_oracle_16384": float((y - y_ref).abs().max())}
        print(f"form {f}: max|y - oracle| at 2x16384 = {out['forms'][str(f)]['max_abs_vs_oracle_16384']:.3e}", flush=True)

    x = synth.synth_audio((args.batch, 2, SEG), seed=200).to(dev)
    cond = synth.synth_audio((1, tcn_cfg["condition_dimension"]), seed=3).to(dev)
    ys = {}
    for f in forms:
        lib.check(lib.mst_tcn_set_tuning(tcn._handle, f), "tuning")
        ys[f] = tcn(x, cond).clone()
        torch.cuda.synchronize()
    if len(forms) > 1:
        d = float((ys[forms[0]] - ys[forms[1]]).abs().max())
        out["max_abs_between_forms_full_size"] = d
        print(f"forms {forms[0]} vs {forms[1]} at {args.batch}x2x{SEG}: max abs difference {d:.3e}", flush=True)
        # the last activation of the two forms
        a = {}
        for f in forms:
            lib.check(lib.mst_tcn_set_tuning(tcn._handle, f), "tuning")
            a[f] = tcn.forward_blocks(x[:2], cond, nb - 1)
        da = float((a[forms[0]] - a[forms[1]]).abs().max())
        out["max_abs_between_forms_block%d" % (nb - 1)] = da
        print(f"activation behind block {nb - 1}: max abs difference {da:.3e} (max |a| {float(a[forms[0]].abs().max()):.2f})", flush=True)

    flop = FLOP * args.batch * SEG
    for r in range(args.rounds):
        for f in forms:
            lib.check(lib.mst_tcn_set_tuning(tcn._handle, f), "tuning")
            ms = timed(lib, tcn, x, cond, args.steps)
            dense = ms[1:nb]
            avg = sum(dense) / len(dense)
            rec = out["forms"][str(f)].setdefault("rounds", [])
            rec.append({"avg_dense_ms": avg, "tflops": flop / (avg * 1e-3) / 1e12, "per_block_ms": ms})
            print(f"round {r} form {f}: dense block avg {avg:.4f} ms = {flop / (avg * 1e-3) / 1e12:.0f} TFLOP/s = "
                  f"{flop / (avg * 1e-3) / 1e12 / 2500:.3f} of peak; blocks " + " ".join(f"{v:.3f}" for v in ms), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
