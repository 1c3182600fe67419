#!/usr/bin/env python
"""File to file: the runner (`inference/style_transfer.py`, the reference's CLI with its default flags `--normalize_input True`,
segment_length 2**19) on one synthetic song - 4 stems x 3 minutes of input and of reference on disk, `mixture_output.wav` out - on
one MI355X.  Prints one JSON line: seconds per song and where they go (wav reading + input normaliser in the dataset, the networks,
writing)."""
import argparse
import json
import os
import sys
import tempfile
import time
import wave

import numpy as np
import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))


def write_wav(path, x):
    pcm = np.clip(np.rint(x * 32767), -32768, 32767).astype("<i2")          # x: [L, 2]
    with wave.open(str(path), "w") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(44100)
        w.writeframes(pcm.tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=180.0)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16x3"])
    a = ap.parse_args()
    print(json.dumps(run(a.seconds, a.precision)))


def run(seconds=180.0, precision="bf16", songs=1):
    a = argparse.Namespace(seconds=seconds, precision=precision)
    import bench_normalizer as BN
    from music_mixing_style_transfer_amd.inference import style_transfer as st
    from music_mixing_style_transfer_amd.utils import synth
    tmp = tempfile.mkdtemp()
    stems = ["drums", "bass", "other", "vocals"]
    L = int(a.seconds * 44100)
    for n in range(songs):
        song = os.path.join(tmp, "data", f"song{n}", "separated")
        for kind in ("input", "reference"):
            os.makedirs(os.path.join(song, kind))
            for k, s in enumerate(stems):
                write_wav(os.path.join(song, kind, s + ".wav"), 0.8 * BN.stem(L, k + (4 if kind == "reference" else 0) + 8 * n))
    np.save(os.path.join(tmp, "features.npy"), BN.features())
    with open(os.path.join(REPO, "music_mixing_style_transfer_amd", "networks", "configs.yaml")) as f:
        cfgs = yaml.full_load(f)
    enc_cfg = cfgs["Effects_Encoder"]["default"]
    synth.save_reference_format_checkpoint(os.path.join(tmp, "enc.pt"), synth.fxencoder_state_dict(enc_cfg, seed=0))
    synth.save_reference_format_checkpoint(os.path.join(tmp, "tcn.pt"), synth.tcn_state_dict(seed=0))
    args = st.build_parser().parse_args([
        "--target_dir", os.path.join(tmp, "data") + "/", "--output_dir", os.path.join(tmp, "out") + "/",
        "--ckpt_path_enc", os.path.join(tmp, "enc.pt"), "--ckpt_path_conv", os.path.join(tmp, "tcn.pt"), "--do_not_separate", "True",
        "--precomputed_normalization_feature", os.path.join(tmp, "features.npy"), "--precision", a.precision])
    args.cfg_encoder, args.cfg_converter = cfgs["Effects_Encoder"]["default"], cfgs["TCN"]["default"]
    assert args.normalize_input is True
    t0 = time.perf_counter()
    runner = st.Mixing_Style_Transfer_Inference(args)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    runner.inference()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # a second song-pass with everything warm (plans, transform kernels, packed weights)
    runner.inference()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    out = os.path.join(tmp, "out", "song0", "mixture_output.wav")
    with wave.open(out) as w:
        assert w.getnframes() == L and w.getnchannels() == 2
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return {"metric": "file-to-file style transfer of 4-stem songs (default flags: --normalize_input True, segment_length 2**19): wav files in, "
                      "input normaliser, FXencoder + MixFXcloner, 16-bit wav files out",
            "unit": "s per song", "song_seconds": a.seconds, "songs": songs, "precision": a.precision,
            "setup_s": t1 - t0, "first_pass_s_per_song": (t2 - t1) / songs, "value": (t3 - t2) / songs,
            "audio_seconds_per_second_warm": a.seconds * songs / (t3 - t2)}


if __name__ == "__main__":
    main()
