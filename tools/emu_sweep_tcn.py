#!/usr/bin/env python
"""Randomised shape sweep of the TCN block kernels on the SIMT emulator (CPU only): nblocks, dilation growth, batch, ragged lengths, FiLM row
forms, bf16 / split-bf16 / fp32 - the module API through tests/emu against the oracle.  Written after the main loops went class-major
(round 4): the fixed emulator tests cover the tile kinds, this covers the shapes in between (first / last tiles full of zero rows, lengths
around the phase-count thresholds, one-step tiles).

    python tools/emu_sweep_tcn.py [--cases 40] [--seed 0]
"""
import argparse
import os
import random
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
TOL = {"fp32": 2e-5, "bf16": 4e-2, "bf16x3": 3e-5}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--tuning", type=int, default=None, help="mst_tcn_set_tuning flags for every model (e.g. 85 = default + bit 6)")
    ap.add_argument("--only", default=None, help="restrict the sweep to one precision")
    ap.add_argument("--fuse0", action="store_true", help="bf16 cases only: also run with mst_tcn_set_tuning bit 5 (block 0 computed by block 1's loader "
                                                          "waves) and require the same bits")
    args = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    from emu_binding import bind_emulator
    from music_mixing_style_transfer_amd import _lib
    from music_mixing_style_transfer_amd.networks import TCNModel
    from music_mixing_style_transfer_amd.utils import synth
    from oracle import networks_ref as R
    emu = bind_emulator()
    _lib.set_default_binding(emu)
    rng = random.Random(args.seed)
    worst = {}
    t0 = time.time()
    for case in range(args.cases):
        growth = rng.choice([2, 2, 2, 3, 4])
        nb = rng.randint(2, 7 if growth == 2 else 4)
        B = rng.randint(1, 3)
        dmax = growth ** (nb - 1)
        L = rng.choice([rng.randint(30, 400), rng.randint(400, 2500), 64 * dmax + rng.randint(-3, 3), 32 * dmax + rng.randint(-2, 2),
                        16 * dmax + rng.randint(-2, 2), 256 * rng.randint(1, 6) + rng.randint(-1, 1)])
        L = max(L, 20)
        prec = "bf16" if args.fuse0 else (args.only or rng.choice(["bf16", "bf16", "bf16x3", "bf16x3", "fp32"]))
        per_item = rng.random() < 0.4
        sd = synth.tcn_state_dict(nblocks=nb, cond_dim=64, seed=case)
        m = TCNModel(nparams=64, ninputs=2, noutputs=2, nblocks=nb, dilation_growth=growth, kernel_size=15, channel_width=128, stack_size=15,
                     cond_dim=64, causal=False)
        m.load_state_dict(sd)
        m.precision = prec
        x = synth.synth_audio((B, 2, L), seed=1000 + case)
        cond = synth.synth_audio((B if per_item else 1, 64), seed=2000 + case)
        col = []
        y_ref = R.tcn_forward(sd, x, cond, nblocks=nb, dilation_growth=growth, collect=col)
        if args.tuning is not None:
            m._ensure(emu)
            emu.check(emu.mst_tcn_set_tuning(m._handle, args.tuning), "tuning")
        y = m(x, cond)
        err = float((y - y_ref).abs().max())
        n_probe = rng.randint(1, nb)
        a = m.forward_blocks(x, cond, n_probe)
        erra = float((a - col[n_probe - 1]).abs().max()) / max(1e-9, float(col[n_probe - 1].abs().max()))
        ok = err <= TOL[prec] and erra <= TOL[prec]
        if args.fuse0:
            emu.check(emu.mst_tcn_set_tuning(m._handle, _lib.TCN_TUNING_DEFAULT | 32), "tuning")
            ok = ok and torch.equal(m(x, cond), y) and torch.equal(m.forward_blocks(x, cond, n_probe), a)
            emu.check(emu.mst_tcn_set_tuning(m._handle, _lib.TCN_TUNING_DEFAULT), "tuning")
        worst[prec] = max(worst.get(prec, 0.0), err)
        print(f"case {case:3d} nb={nb} g={growth} B={B} L={L:5d} {prec:6s} film_rows={'B' if per_item else '1'}: waveform {err:.2e}, block {n_probe} rel {erra:.2e} "
              f"{'ok' if ok else 'FAIL'}", flush=True)
        if not ok:
            raise SystemExit(1)
    print(f"{args.cases} cases in {time.time() - t0:.0f} s; worst waveform error per mode: " + ", ".join(f"{k} {v:.2e}" for k, v in sorted(worst.items())))


if __name__ == "__main__":
    main()
