"""REAL music through the whole path, against the reference's OWN end-to-end run (tests/golden/real_audio.npz, written by
`make_golden.py real_audio`: the real Mixing_Style_Transfer_Inference.inference() of /root/reference/inference/style_transfer.py:112-177 with the
real FXencoder / TCNModel on the stems the reference ships - 661 538-sample inputs, 882 433-sample references, 2^19 segments - plus a derived
song with a segment of exact digital silence, +-full-scale saturated stems and an all-zero reference).

CPU: the fixture's codec; the oracle against the reference's run (embeddings of all four references, two converter segments).
GPU: the product's command line on the same bytes - fp32 and bf16x3 within 1e-4 (+ one 16-bit step) on every stem and on the mixture, bf16
within 1e-2 with the measured deviation printed; embeddings, fp64 checksums and clamp counts of the float outputs."""
import os
import sys

import numpy as np
import pytest
import torch
import yaml

from music_mixing_style_transfer_amd.utils import synth

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import real_audio as RA  # noqa: E402

L_IN, L_REF = 661_538, 882_433


@pytest.fixture(scope="module")
def gold():
    g = np.load(os.path.join(HERE, "golden", "real_audio.npz"))
    pcm = {k[4:]: RA.unpack(g[k]) for k in g.files if k.startswith("pcm/")}
    return g, {"real": pcm, "xtreme": RA.extremes_from(pcm)}


def _cfgs():
    with open(os.path.join(REPO, "music_mixing_style_transfer_amd", "networks", "configs.yaml")) as f:
        cfgs = yaml.full_load(f)
    return cfgs["Effects_Encoder"]["default"], cfgs["TCN"]["default"]


def _decode(pcm):
    """the wave reader's arithmetic (loader_utils.py:47-70) + the dataset's float / clamp (data_loader.py:589): int16 [L, 2] -> float32 [2, L]"""
    return np.clip((pcm.astype(np.float64) / 2.0 ** 15).T, -1, 1).astype(np.float32)


def test_fixture_holds_the_reference_files(gold):
    g, songs = gold
    assert list(g["stems"]) == list(RA.STEMS) and int(g["seg"]) == RA.SEG
    for s in RA.STEMS:
        assert songs["real"][f"input/{s}"].shape == (L_IN, 2) and songs["real"][f"reference/{s}"].shape == (L_REF, 2)
    x = songs["xtreme"]
    assert not x["input/drums"][:RA.SEG].any() and not x["reference/drums"].any()                   # exact digital silence
    assert x["input/drums"].min() == -32768 and x["input/drums"].max() == 32767                      # both ends of the 16-bit range
    assert (np.abs(x["input/bass"].astype(np.int32)) >= 32767).mean() > 0.15                         # a fifth of the samples sit ON the rails
    rng = np.random.default_rng(0)
    for shape in ((1, 1), (1000, 2), (4097, 3)):
        v = rng.integers(-32768, 32768, size=shape).astype(np.int16)
        assert np.array_equal(RA.unpack(RA.pack(v)), v)
    mix = RA.unpack(g["real/mix/pcm16"])
    assert mix.shape == (L_IN, 2)
    assert np.array_equal(mix[g["probe_idx"]], np.rint(g["real/mix/probe"].astype(np.float64) * 32767.0).astype(np.int16))


def test_oracle_against_the_reference_run_on_real_audio(gold):
    """Pins oracle/networks_ref.py + segmentation_ref.py on real-audio statistics: the four reference embeddings (one un-segmented encoder call
    on 882 433 samples each - incl. the all-zero stem) and two converter segments - the real drums' zero-padded tail and the saturated bass' tail."""
    from oracle import networks_ref as R
    from oracle import segmentation_ref as O
    g, songs = gold
    enc_cfg, _ = _cfgs()
    enc_sd, tcn_sd = synth.fxencoder_state_dict(enc_cfg, seed=0), synth.tcn_state_dict(seed=0)
    idx = g["probe_idx"]
    for song in ("real", "xtreme"):
        for stem in RA.STEMS:
            rb = O.reference_batches(_decode(songs[song][f"reference/{stem}"]), RA.SEG, RA.SEG, 1)
            assert len(rb) == 1 and rb[0].shape == (1, 2, L_REF)                                     # below 2 * segment_length: not cut
            emb = R.fxencoder_forward(enc_sd, enc_cfg, torch.from_numpy(rb[0]))[0].numpy()
            assert np.abs(emb - g[f"{song}/{stem}/emb"]).max() <= 1e-5 * np.abs(g[f"{song}/{stem}/emb"]).max(), (song, stem)
    for song, stem in (("real", "drums"), ("xtreme", "bass")):
        ib = O.input_batches(_decode(songs[song][f"input/{stem}"]), RA.SEG, 1)
        assert len(ib) == 2
        y = R.tcn_forward(tcn_sd, torch.from_numpy(ib[1]), torch.from_numpy(g[f"{song}/{stem}/emb"])[None])[0].numpy()
        sel = idx[idx >= RA.SEG]
        assert np.abs(y[:, sel - RA.SEG].T - g[f"{song}/{stem}/probe"][idx >= RA.SEG]).max() <= 1e-5, (song, stem)


# ------------------------------------------------------------------------------------------------------------------------ GPU
def _run_cli(tmp_path, songs, precision):
    from music_mixing_style_transfer_amd.inference import style_transfer as st
    enc_cfg, tcn_cfg = _cfgs()
    synth.save_reference_format_checkpoint(str(tmp_path / "enc.pt"), synth.fxencoder_state_dict(enc_cfg, seed=0))
    synth.save_reference_format_checkpoint(str(tmp_path / "tcn.pt"), synth.tcn_state_dict(seed=0))
    RA.stage(tmp_path / "data", songs)
    args = st.build_parser().parse_args([
        "--target_dir", str(tmp_path / "data") + "/", "--output_dir", str(tmp_path / "out") + "/", "--ckpt_path_enc", str(tmp_path / "enc.pt"),
        "--ckpt_path_conv", str(tmp_path / "tcn.pt"), "--do_not_separate", "True", "--normalize_input", "False", "--save_each_inst", "True",
        "--precision", precision])                     # default segment lengths (2**19) and batch size, like the reference's run
    args.instruments = list(RA.STEMS)
    args.cfg_encoder, args.cfg_converter = enc_cfg, tcn_cfg
    runner = st.Mixing_Style_Transfer_Inference(args)
    runner.inference()
    return runner


def _read_pcm(path):
    import wave
    with wave.open(str(path)) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate()) == (2, 2, 44100)
        return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, 2).astype(np.int32)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16x3", 1e-4), ("bf16", 1e-2)])
def test_cli_on_real_audio_against_the_reference_run(tmp_path, gold, precision, tol):
    """The product's command line on the bytes the reference ran on.  Files are compared as 16-bit integers: |file - lrint(32767 * reference)| <=
    tol * 32767 + 1 (the tolerance on the waveform + one rounding step), for every stem of both songs at the stored probes and for the WHOLE
    mixture of the real song; the all-zero segment and the +-full-scale stems are part of it.  The measured deviations are printed."""
    g, songs = gold
    runner = _run_cli(tmp_path, songs, precision)
    idx = g["probe_idx"]
    steps = int(np.floor(tol * 32767)) + 1
    worst = {}
    for song in ("real", "xtreme"):
        out = tmp_path / "out" / song
        for stem in RA.STEMS:
            got = _read_pcm(out / f"{stem}_output_notnormed.wav")
            assert got.shape == (L_IN, 2)
            want = np.rint(g[f"{song}/{stem}/probe"].astype(np.float64) * 32767.0).astype(np.int32)
            worst[f"{song}/{stem}"] = int(np.abs(got[idx] - want).max())
        got = _read_pcm(out / "mixture_output_notnormed.wav")
        want = np.clip(np.rint(g[f"{song}/mix/probe"].astype(np.float64) * 32767.0), -32768, 32767).astype(np.int32)   # the writer saturates beyond +-1
        worst[f"{song}/mix"] = int(np.abs(got[idx] - want).max())
    mix = _read_pcm(tmp_path / "out" / "real" / "mixture_output_notnormed.wav")
    worst["real/mix (all 661538 x 2 samples)"] = int(np.abs(mix - RA.unpack(g["real/mix/pcm16"]).astype(np.int32)).max())
    print(f"real audio, {precision}: max |16-bit file - reference run| in 16-bit steps (allowed {steps} per stem): {worst}")
    for k, v in worst.items():
        assert v <= (steps if "mix" not in k else 2 * steps - 1), (k, v)              # the mixture adds two stems' deviations
    # the digital-silence segment: the converter's answer to exact zeros is part of the comparison above; it must also be finite and quiet
    sil = _read_pcm(tmp_path / "out" / "xtreme" / "drums_output_notnormed.wav")[:RA.SEG]
    assert np.abs(sil).max() < 32767
    # float side: embeddings, checksums and clamp counts of the un-rounded outputs, straight from the engine
    eng = runner._engine()
    rel = 1e-4 if precision != "bf16" else 2e-2
    dev_emb, dev_y = 0.0, 0.0
    for song in ("real", "xtreme"):
        for stem in RA.STEMS:
            x_in = torch.from_numpy(_decode(songs[song][f"input/{stem}"])).cuda()
            x_ref = torch.from_numpy(_decode(songs[song][f"reference/{stem}"])).cuda()
            emb = eng.mean_embedding(x_ref, 1, None).cpu().numpy()
            e = np.abs(emb - g[f"{song}/{stem}/emb"]).max() / max(1.0, np.abs(g[f"{song}/{stem}/emb"]).max())
            dev_emb = max(dev_emb, float(e))
            assert e <= rel, (song, stem, "embedding", e)
            y = eng.transfer_stem(x_in, x_ref, RA.SEG, RA.SEG).cpu().numpy()
            assert y.shape == (2, L_IN)
            d = float(np.abs(y.T[idx] - g[f"{song}/{stem}/probe"]).max())
            dev_y = max(dev_y, d)
            assert d <= tol, (song, stem, "waveform", d)
            assert int((np.abs(y) >= 1.0).sum()) == int(g[f"{song}/{stem}/clamped"]) == 0
            ab = np.abs(y.astype(np.float64)).sum(1)
            assert np.abs(ab - g[f"{song}/{stem}/abs"]).max() <= (1e-5 if precision != "bf16" else 1e-2) * g[f"{song}/{stem}/abs"].max(), (song, stem, ab)
    print(f"real audio, {precision}: max |y - reference run| = {dev_y:.3e} (bound {tol:g}), max embedding deviation = {dev_emb:.3e} (relative to max(1, |emb|))")
