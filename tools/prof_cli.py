#!/usr/bin/env python
"""Where a warm file-to-file song goes (the `file_to_file` leg of bench.py): cProfile of ONE warm `runner.inference()` pass of
tools/bench_cli.py's setup with --workers 0 (no prefetch thread, so the host profile is one thread's), plus wall times of the
stages with device synchronisation between them.   python tools/prof_cli.py [--seconds 180] > gpurun_out/prof_cli.txt"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

import numpy as np
import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=180.0)
    ap.add_argument("--workers", type=int, default=0)
    a = ap.parse_args()
    import bench_cli
    import bench_normalizer as BN
    from music_mixing_style_transfer_amd.inference import style_transfer as st
    from music_mixing_style_transfer_amd.utils import synth
    tmp = tempfile.mkdtemp()
    stems = ["drums", "bass", "other", "vocals"]
    L = int(a.seconds * 44100)
    for n in range(2):
        song = os.path.join(tmp, "data", f"song{n}", "separated")
        for kind in ("input", "reference"):
            os.makedirs(os.path.join(song, kind))
            for k, s in enumerate(stems):
                bench_cli.write_wav(os.path.join(song, kind, s + ".wav"), 0.8 * BN.stem(L, k + (4 if kind == "reference" else 0) + 8 * n))
    np.save(os.path.join(tmp, "features.npy"), BN.features())
    with open(os.path.join(REPO, "music_mixing_style_transfer_amd", "networks", "configs.yaml")) as f:
        cfgs = yaml.full_load(f)
    enc_cfg = cfgs["Effects_Encoder"]["default"]
    synth.save_reference_format_checkpoint(os.path.join(tmp, "enc.pt"), synth.fxencoder_state_dict(enc_cfg, seed=0))
    synth.save_reference_format_checkpoint(os.path.join(tmp, "tcn.pt"), synth.tcn_state_dict(seed=0))
    args = st.build_parser().parse_args([
        "--target_dir", os.path.join(tmp, "data") + "/", "--output_dir", os.path.join(tmp, "out") + "/",
        "--ckpt_path_enc", os.path.join(tmp, "enc.pt"), "--ckpt_path_conv", os.path.join(tmp, "tcn.pt"), "--do_not_separate", "True",
        "--precomputed_normalization_feature", os.path.join(tmp, "features.npy"), "--precision", "bf16", "--workers", str(a.workers)])
    args.cfg_encoder, args.cfg_converter = cfgs["Effects_Encoder"]["default"], cfgs["TCN"]["default"]
    runner = st.Mixing_Style_Transfer_Inference(args)
    runner.inference()                      # cold pass: plans, run-time compiled transform kernels, packed weights
    torch.cuda.synchronize()
    # stage times of the dataset's work for one song, synchronised
    ds = runner.data_loader
    t0 = time.perf_counter()
    item = ds[0]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"dataset item (decode 8 stems + normalise 4): {t1 - t0:.3f} s")
    nz = ds.normalization_chain
    from music_mixing_style_transfer_amd.data_loader.loader_utils import load_wav_device
    path = os.path.join(tmp, "data", "song0", "separated", "input", "drums.wav")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wav = load_wav_device(path, runner.device, sample_rate=44100)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"load_wav_device: {t1 - t0:.3f} s")
    x = wav.t().contiguous()
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = nz.normalize_audio(x, src="drums")
        torch.cuda.synchronize()
        print(f"normalize_audio(drums) pass {rep}: {time.perf_counter() - t0:.3f} s")
    cur = x
    for eff in nz.EFFECTS:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cur = nz.normalize_audio_per_effect(cur, "drums", eff)
        torch.cuda.synchronize()
        print(f"  effect {eff}: {time.perf_counter() - t0:.3f} s")
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr.enable()
    runner.inference()
    torch.cuda.synchronize()
    pr.disable()
    dt = time.perf_counter() - t0
    print(f"warm pass over 2 songs with --workers {a.workers}: {dt:.3f} s = {dt / 2:.3f} s per song")
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70)
    print(s.getvalue()[:14000])
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
    print(s.getvalue()[:7000])
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
