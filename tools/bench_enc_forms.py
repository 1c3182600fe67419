#!/usr/bin/env python
"""A/B of FXencoder schedules on the MI355X (mst_enc_set_schedule flags): the encoder pass over 32 x 2 x 131072 in bf16, forms alternating in one
process, bit identity of the embeddings between the forms checked first.

    python tools/bench_enc_forms.py [--forms 1,65] [--steps 20] [--rounds 3]
"""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--forms", default="1,65")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    import yaml
    from music_mixing_style_transfer_amd import _lib
    from music_mixing_style_transfer_amd.networks import FXencoder
    from music_mixing_style_transfer_amd.utils import synth
    dev = torch.device("cuda", 0)
    with open(os.path.join(REPO, "music_mixing_style_transfer_amd", "networks", "configs.yaml")) as f:
        cfg = yaml.full_load(f)["Effects_Encoder"]["default"]
    sd = synth.fxencoder_state_dict(cfg, seed=0)          # (the module's constructor mutates its config like the reference's: a copy)
    enc = FXencoder({k: (list(v) if isinstance(v, list) else v) for k, v in cfg.items()}).to(dev)
    enc.load_state_dict(sd)
    enc.precision = "bf16"
    lib = _lib.lib()
    x = synth.synth_audio((args.batch, 2, 131072), seed=100).to(dev)
    run = enc._get_runner()
    run._ensure(lib)
    forms = [int(v) for v in args.forms.split(",")]
    ref = None
    for f in forms:
        lib.check(lib.mst_enc_set_schedule(run.handle, f), "schedule")
        for rep in range(3):
            e = enc(x).clone()
            torch.cuda.synchronize()
            if ref is None:
                ref = e
            print(f"schedule {f} run {rep}: bit-identical to schedule {forms[0]}: {bool(torch.equal(e, ref))}  max abs diff {float((e - ref).abs().max()):.3e}", flush=True)
    for r in range(args.rounds):
        for f in forms:
            lib.check(lib.mst_enc_set_schedule(run.handle, f), "schedule")
            for _ in range(3):
                enc(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                enc(x)
            torch.cuda.synchronize()
            print(f"round {r} schedule {f}: {(time.perf_counter() - t0) / args.steps * 1e3:.4f} ms per encoder pass", flush=True)


if __name__ == "__main__":
    main()
