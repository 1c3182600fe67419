#!/usr/bin/env python
"""Row F at a real size: Audio_Effects_Normalizer.normalize_audio (the reference CLI's default `--normalize_input True`,
data_normalization.py:76-155) on one synthetic stereo stem of --seconds (default 180 s = 7 938 000 samples) per stem type on one
MI355X, effect by effect; the oracle chain (numpy / scipy / oracle/fx_ref.c) timed on --cpu-seconds of the same stem beside it.
Prints one JSON line."""
import argparse
import copy
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def features():
    k = np.arange(32769)
    eq = lambda a, b: (a / (1.0 + (k / b) ** 1.3) + 0.02).astype(np.float64)
    return {"eq": {"drums": eq(40.0, 900.0), "bass": eq(60.0, 150.0), "other": eq(30.0, 600.0), "vocals": eq(35.0, 700.0)},
            "compression": {"drums": [-14.0, 2.0], "bass": [-12.0, 2.5], "other": [-15.0, 2.0], "vocals": [-13.0, 2.0]},
            "imager": {"drums": 0.8, "bass": 0.95, "other": 0.7, "vocals": 0.85},
            "loudness": {"drums": -20.0, "bass": -22.0, "other": -24.0, "vocals": -21.0}}


def stem(L, k):
    """band-limited music-like bed + decaying noise bursts every ~0.4 s (something for the onset detector and the compressor)"""
    from music_mixing_style_transfer_amd.utils import synth
    base = synth.synth_music(2, L, seed=40 + k).numpy().T
    rng = np.random.default_rng(7 + k)
    noise = rng.standard_normal(L).astype(np.float32)
    env = np.zeros(L, np.float32)
    n0 = 2000
    while n0 < L - 4000:
        n1 = min(L, n0 + 12000)
        env[n0:n1] += (0.3 + 0.6 * rng.random()) * np.exp(-np.arange(n1 - n0) / 1800.0).astype(np.float32)
        n0 += int(44100 * (0.3 + 0.25 * rng.random()))
    d = env * (0.6 * noise * 0.3 + 0.4 * np.sin(2 * np.pi * 180.0 * np.arange(L) / 44100.0).astype(np.float32))
    return (0.25 * base + np.stack([d, (0.5 + 0.1 * k) * np.roll(d, 40 * k)], 1)).astype(np.float32)


def run(seconds=180.0, cpu_seconds=6.0, which="drums,bass,other,vocals"):
    args = argparse.Namespace(seconds=seconds, cpu_seconds=cpu_seconds, stems=which)
    from music_mixing_style_transfer_amd.mixing_manipulator.data_normalization import Audio_Effects_Normalizer
    order = ["loudness", "eq", "compression", "imager", "loudness"]
    stems = ["drums", "bass", "other", "vocals"]
    tmp = tempfile.mkdtemp()
    np.save(os.path.join(tmp, "features.npy"), features())
    norm = Audio_Effects_Normalizer(os.path.join(tmp, "features.npy"), STEMS=stems, EFFECTS=order)
    L = int(args.seconds * 44100)
    per_stem, per_effect = {}, {}
    norm.normalize_audio(stem(44100 * 2, 0), "drums")            # plans, scratch, first launches
    torch.cuda.synchronize()
    for k, s in enumerate(stems):
        if s not in args.stems.split(","):
            continue
        x = stem(L, k)
        t0 = time.perf_counter()
        y = x
        for eff in order:
            t1 = time.perf_counter()
            y = norm.normalize_audio_per_effect(y, src=s, effect=eff)
            torch.cuda.synchronize()
            per_effect.setdefault(eff, 0.0)
            per_effect[eff] += time.perf_counter() - t1
        per_stem[s] = time.perf_counter() - t0
        assert y.shape == x.shape and np.isfinite(y).all()
    # the oracle on a short excerpt of the first stem
    import ctypes as C
    import subprocess
    from oracle import normalizer_ref as N
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle")], check=True, capture_output=True)
    lib = C.CDLL(os.path.join(REPO, "oracle", "libfx_ref.so"))
    fp = C.POINTER(C.c_float)

    def cc(xx, threshold, attack_time, release_time, ratio, sample_rate):
        xx = np.ascontiguousarray(xx, dtype=np.float32)
        yy = np.empty_like(xx)
        lib.ref_compressor(xx.ctypes.data_as(fp), yy.ctypes.data_as(fp), C.c_long(xx.shape[0]), xx.shape[1], C.c_double(threshold),
                           C.c_double(attack_time), C.c_double(release_time), C.c_double(ratio), C.c_double(0.0), C.c_double(sample_rate))
        return yy
    feats = N.smooth_features(copy.deepcopy(features()), stems, order)
    Lc = int(args.cpu_seconds * 44100)
    k0 = stems.index(args.stems.split(",")[0])
    xc = stem(L, k0)[:Lc]
    t2 = time.perf_counter()
    rc = N.normalize_audio(xc, stems[k0], feats, order, compress_fn=cc)
    cpu_dt = time.perf_counter() - t2
    yc = norm.normalize_audio(xc, stems[k0])
    dev = float(np.abs(yc - rc).max() / np.abs(rc).max())
    total = sum(per_stem.values())
    audio_s = args.seconds * len(per_stem)
    return ({"metric": "input normaliser: seconds of stereo audio normalised per second (loudness, eq, compression, imager, loudness)",
                      "value": audio_s / total, "unit": "audio-s/s", "stem_seconds": args.seconds, "stems": list(per_stem),
                      "s_per_stem": per_stem, "s_per_effect": per_effect,
                      "rel_dev_vs_oracle_on_excerpt": dev,
                      "cpu_baseline": {"value": args.cpu_seconds / cpu_dt, "unit": "audio-s/s", "cores": 1, "kind": "port",
                                       "sample": f"{args.cpu_seconds:g} s excerpt of the first stem, oracle/normalizer_ref.py (numpy/scipy + oracle/fx_ref.c compressor)"}})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=180.0)
    ap.add_argument("--cpu-seconds", type=float, default=6.0)
    ap.add_argument("--stems", default="drums,bass,other,vocals")
    args = ap.parse_args()
    print(json.dumps(run(args.seconds, args.cpu_seconds, args.stems)))


if __name__ == "__main__":
    main()
