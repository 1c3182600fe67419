"""world_size-2 run of the segment-sharded engine on CPU (gloo backend + the SIMT emulator build of the kernels):
each rank encodes / converts its contiguous shard of segments, the only exchange is the all-gather of segment
embeddings, and the result equals the single-process run bit for bit (canonical-order mean)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ENC_CFG = {"channels": [16, 40, 64], "kernels": [25, 5, 4], "strides": [4, 2, 2], "dilation": [1, 1, 1],
           "bias": True, "norm": "batch", "conv_block": "res", "activation": "relu"}
SEG, L_IN, L_REF = 512, 2300, 2900          # 5 input segments, 6 reference segments (uneven over 2 ranks)


def _models(nblocks=3, enc_cfg=None):
    from music_mixing_style_transfer_amd.networks import FXencoder, TCNModel
    from music_mixing_style_transfer_amd.utils import synth
    enc_cfg = ENC_CFG if enc_cfg is None else enc_cfg
    enc = FXencoder({k: (list(v) if isinstance(v, list) else v) for k, v in enc_cfg.items()})
    enc.load_state_dict(synth.fxencoder_state_dict(enc_cfg, seed=1))
    tcn = TCNModel(nparams=64, ninputs=2, noutputs=2, nblocks=nblocks, dilation_growth=2, kernel_size=15, channel_width=128,
                   stack_size=15, cond_dim=64, causal=False)
    tcn.load_state_dict(synth.tcn_state_dict(nblocks=nblocks, cond_dim=64, seed=2))
    return enc.eval(), tcn.eval()


def _worker(rank, world, port, out_path, pass_samples=None):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MST_EMU_THREADS="2" if world <= 2 else "1")
    torch.set_num_threads(1)
    from music_mixing_style_transfer_amd import _lib
    from music_mixing_style_transfer_amd.inference import StyleTransferEngine
    from music_mixing_style_transfer_amd.utils import synth
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from emu_binding import bind_emulator
    b = bind_emulator(build=False)
    _lib.set_default_binding(b)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    enc, tcn = _models()
    eng = StyleTransferEngine(enc, tcn) if pass_samples is None else StyleTransferEngine(enc, tcn, pass_samples=pass_samples)
    x_in = synth.synth_audio((2, L_IN), seed=5)
    x_ref = synth.synth_audio((2, L_REF), seed=6)
    res = eng.transfer_stem(x_in, x_ref, SEG, SEG)
    if world == 1:
        torch.save({"full": res}, out_path)
    else:
        y, rng = res                               # this rank's time range only
        ranges = [None] * world
        dist.all_gather_object(ranges, tuple(rng))
        full = eng.gather_stem(y, rng, L_IN)       # collected on rank 0 only
        assert (full is None) == (rank != 0)
        if rank == 0:
            torch.save({"full": full, "ranges": ranges}, out_path)
        dist.destroy_process_group()


def test_two_ranks_equal_one(tmp_path, emu):
    from music_mixing_style_transfer_amd import _lib
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    prev = _lib._default
    try:
        _worker(0, 1, 0, one)          # single-process reference run (binds the emulator in this process)
    finally:
        _lib.set_default_binding(prev)
    mp.spawn(_worker, args=(2, 29641, two), nprocs=2, join=True)
    full = torch.load(one)["full"]
    res = torch.load(two, weights_only=False)
    # contiguous shards of the 5 input segments: rank 0 owns segments [0, 2), rank 1 [2, 5) -> time ranges cropped to L_IN
    assert res["ranges"] == [(0, 2 * SEG), (2 * SEG, L_IN)]
    assert res["full"].shape == full.shape == (2, L_IN)
    assert torch.equal(res["full"], full)


def test_eight_ranks_three_of_them_without_segments(tmp_path, emu):
    """world size 8 over 5 input / 6 reference segments: shards floor(r S / 8) .. floor((r + 1) S / 8) leave three ranks without an input
    segment and two without a reference segment - the empty-rows path of mean_embedding, the ragged branch of gather_embeddings and
    zero-length time ranges in gather_stem; the result is the single-process one bit for bit."""
    from music_mixing_style_transfer_amd import _lib
    from music_mixing_style_transfer_amd.inference import segmentation as seg
    one, eight = str(tmp_path / "one.pt"), str(tmp_path / "eight.pt")
    prev = _lib._default
    try:
        _worker(0, 1, 0, one)
    finally:
        _lib.set_default_binding(prev)
    mp.spawn(_worker, args=(8, 29683, eight), nprocs=8, join=True)
    full = torch.load(one)["full"]
    res = torch.load(eight, weights_only=False)
    shards = [seg.shard_range(5, r, 8) for r in range(8)]
    assert sum(1 for lo, hi in shards if lo == hi) == 3
    assert res["ranges"] == [(min(L_IN, lo * SEG), min(L_IN, hi * SEG)) for lo, hi in shards]
    assert torch.equal(res["full"], full)


def test_two_ranks_with_passes_smaller_than_a_shard(tmp_path, emu):
    """pass_samples = one segment: every rank walks its shard (2 and 3 input segments, 3 and 3 reference segments) in several network
    passes; bit-equal to the single process that runs everything in one pass."""
    from music_mixing_style_transfer_amd import _lib
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    prev = _lib._default
    try:
        _worker(0, 1, 0, one)
    finally:
        _lib.set_default_binding(prev)
    mp.spawn(_worker, args=(2, 29689, two, SEG), nprocs=2, join=True)
    assert torch.equal(torch.load(two, weights_only=False)["full"], torch.load(one)["full"])


def _cli_worker(rank, world, port, out_path):
    """The style_transfer runner's inference() loop: plain (world 1) or sharded over the ranks (world 2)."""
    import types
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MST_EMU_THREADS="2")
    torch.set_num_threads(1)
    from music_mixing_style_transfer_amd import _lib
    from music_mixing_style_transfer_amd.inference import style_transfer as st
    from music_mixing_style_transfer_amd.utils import synth
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from emu_binding import bind_emulator
    b = bind_emulator(build=False)
    _lib.set_default_binding(b)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    enc, tcn = _models()
    runner = object.__new__(st.Mixing_Style_Transfer_Inference)
    runner.args = types.SimpleNamespace(normalize_input=False, instruments=["drums", "bass"], segment_length=SEG,
                                        segment_length_ref=SEG, batch_size=1, save_each_inst=True, sample_rate=44100)
    runner.device = torch.device("cpu")
    runner.target_dir, runner.output_dir = "/data/", out_path + "/"
    runner.models = {"effects_encoder": enc, "mixing_converter": tcn}
    stems_in = torch.stack([synth.synth_audio((2, L_IN), seed=5), synth.synth_audio((2, L_IN), seed=7)])
    stems_ref = torch.stack([synth.synth_audio((2, L_REF), seed=6), synth.synth_audio((2, L_REF), seed=8)])
    runner.data_loader = [(stems_in, stems_ref, "/data/song/")]
    runner.inference()                          # one process: plain wav writer; two: every rank writes its own time range
    if world > 1:
        dist.destroy_process_group()


def test_cli_runner_sharded_equals_single(tmp_path, emu):
    """`style_transfer` launched on two ranks shards every stem's segments, each rank writes the time range it produced
    into the output files, and the files are byte-identical to the single-process run's."""
    from music_mixing_style_transfer_amd import _lib
    from music_mixing_style_transfer_amd.data_loader import load_wav_length
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    prev = _lib._default
    try:
        _cli_worker(0, 1, 0, one)
    finally:
        _lib.set_default_binding(prev)
    mp.spawn(_cli_worker, args=(2, 29653, two), nprocs=2, join=True)
    names = ["bass_output_notnormed.wav", "drums_output_notnormed.wav", "mixture_output_notnormed.wav"]
    assert sorted(os.listdir(os.path.join(one, "song"))) == sorted(os.listdir(os.path.join(two, "song"))) == names
    for k in names:
        a, b = os.path.join(one, "song", k), os.path.join(two, "song", k)
        assert load_wav_length(a) == L_IN
        assert open(a, "rb").read() == open(b, "rb").read(), k


def _norm_features():
    import numpy as np
    k = np.arange(32769)
    eq = lambda a, b: (a / (1.0 + (k / b) ** 1.3) + 0.02).astype(np.float64)
    return {"eq": {"drums": eq(40.0, 900.0), "bass": eq(60.0, 150.0)}, "compression": {"drums": [-14.0, 2.0], "bass": [-12.0, 2.5]},
            "imager": {"drums": 0.8, "bass": 0.9}, "loudness": {"drums": -20.0, "bass": -22.0}}


def _cli_files_worker(rank, world, port, root, workers=0, order=("loudness", "eq", "imager", "loudness")):
    """The runner over FILES with --normalize_input True: Song_Dataset_Inference decodes the wavs and normalises the input stems -
    on two ranks stem j is normalised by rank j % 2 only and broadcast, every rank writes its time range of the output files."""
    import types
    import numpy as np
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MST_EMU_THREADS="8")
    torch.set_num_threads(4)
    from music_mixing_style_transfer_amd import _lib
    from music_mixing_style_transfer_amd.data_loader import Song_Dataset_Inference
    from music_mixing_style_transfer_amd.inference import style_transfer as st
    from music_mixing_style_transfer_amd.mixing_manipulator.data_normalization import Audio_Effects_Normalizer
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from emu_binding import bind_emulator
    _lib.set_default_binding(bind_emulator(build=False))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    orig = Audio_Effects_Normalizer.normalize_audio
    Audio_Effects_Normalizer.normalize_audio = lambda self, audio, src: (calls.append(src), orig(self, audio, src))[1]
    enc, tcn = _models(nblocks=2, enc_cfg={"channels": [16, 64], "kernels": [25, 5], "strides": [8, 4], "dilation": [1, 1], "bias": True,
                                           "norm": "batch", "conv_block": "res", "activation": "relu"})      # 20 000-sample stems: a light pair of nets
    a = types.SimpleNamespace(normalize_input=True, instruments=["drums", "bass"], segment_length=4096, segment_length_ref=4096, batch_size=1,
                              save_each_inst=True, sample_rate=44100, target_dir=os.path.join(root, "data") + "/", interpolation=False,
                              input_file_name="input", reference_file_name="reference", stem_level_directory_name="separated",
                              do_not_separate=True, precomputed_normalization_feature=_norm_features(),
                              normalization_order=list(order), workers=workers)      # (compression matching: a minute on the emulator; test_normalizer.py)
    runner = object.__new__(st.Mixing_Style_Transfer_Inference)
    runner.args, runner.device = a, torch.device("cpu")
    runner.target_dir, runner.output_dir = a.target_dir, os.path.join(root, f"out{world}{'w' if workers else ''}") + "/"
    runner.models = {"effects_encoder": enc, "mixing_converter": tcn}
    runner.data_loader = Song_Dataset_Inference(a)
    runner.data_loader.dist = runner._world()
    runner.inference()
    np.save(os.path.join(root, f"calls_w{world}_r{rank}.npy"), np.array(calls))
    if world > 1:
        dist.destroy_process_group()


def _write_songs(root, songs, L=20000):
    import wave
    import numpy as np
    from music_mixing_style_transfer_amd.utils import synth
    t = np.arange(L)
    for si, song in enumerate(songs):
        for kind, seed in (("input", 50 + 100 * si), ("reference", 60 + 100 * si)):
            d = os.path.join(root, "data", song, "separated", kind)
            os.makedirs(d)
            for k, s in enumerate(("drums", "bass")):
                base = synth.synth_music(2, L, seed=seed + k).numpy().T
                hits = sum(a * np.exp(-np.maximum(0, t - n0) / 1200.0) * (t >= n0) * np.sin(2 * np.pi * (90.0 + 40 * k) * t / 44100.0)
                           for n0, a in ((1500, 0.9), (7000, 0.6), (12500, 0.8)))
                x = np.clip(0.3 * base + np.stack([hits, 0.8 * hits], 1), -1, 1)
                with wave.open(os.path.join(d, s + ".wav"), "w") as w:
                    w.setnchannels(2)
                    w.setsampwidth(2)
                    w.setframerate(44100)
                    w.writeframes(np.clip(np.rint(x * 32767), -32768, 32767).astype("<i2").tobytes())


def test_cli_files_with_normalize_input_sharded_by_stems(tmp_path, emu):
    """--normalize_input True on two ranks: byte-identical output files, and each rank ran the normaliser on ONE of the two input stems."""
    import numpy as np
    from music_mixing_style_transfer_amd import _lib
    root = str(tmp_path)
    _write_songs(root, ["song"])
    prev = _lib._default
    try:
        _cli_files_worker(0, 1, 0, root)
    finally:
        _lib.set_default_binding(prev)
    mp.spawn(_cli_files_worker, args=(2, 29667, root), nprocs=2, join=True)
    names = ["bass_output.wav", "drums_output.wav", "mixture_output.wav"]
    for k in names:
        a, b = os.path.join(root, "out1", "song", k), os.path.join(root, "out2", "song", k)
        assert open(a, "rb").read() == open(b, "rb").read(), k
    assert list(np.load(os.path.join(root, "calls_w1_r0.npy"))) == ["drums", "bass"]
    assert list(np.load(os.path.join(root, "calls_w2_r0.npy"))) == ["drums"] and list(np.load(os.path.join(root, "calls_w2_r1.npy"))) == ["bass"]


def test_two_songs_two_ranks_with_prefetch_worker_requested(tmp_path, emu):
    """The reference CLI's defaults (--workers 1, --normalize_input True) under two ranks and TWO songs: the loader's broadcasts of song
    i + 1 must not run on a prefetch thread beside the engine's all-gather / the runner's barrier of song i (collectives of two threads
    on one process group pair up differently on different ranks) - multi-rank runs prepare their songs inline.  Byte-identical to the
    single process without prefetch; the single process WITH the prefetch thread gives the same files too."""
    from music_mixing_style_transfer_amd import _lib
    root = str(tmp_path)
    _write_songs(root, ["song_a", "song_b"], L=8500)          # three segments of 4096 per stem
    prev = _lib._default
    try:
        order = ("loudness", "imager")               # the subject is the order of the collectives, not the normaliser (EQ matching: 20 s per song on the emulator)
        _cli_files_worker(0, 1, 0, root, 0, order)
        _cli_files_worker(0, 1, 0, root, 1, order)   # one process, prefetch thread active (and joined when the loop ends)
    finally:
        _lib.set_default_binding(prev)
    mp.spawn(_cli_files_worker, args=(2, 29671, root, 1, order), nprocs=2, join=True)
    import threading
    assert not [t for t in threading.enumerate() if t.name.startswith("mst-prefetch")]
    for song in ("song_a", "song_b"):
        for k in ["bass_output.wav", "drums_output.wav", "mixture_output.wav"]:
            a = open(os.path.join(root, "out1", song, k), "rb").read()
            assert a == open(os.path.join(root, "out1w", song, k), "rb").read(), (song, k)
            assert a == open(os.path.join(root, "out2w", song, k), "rb").read(), (song, k)


def _interp_worker(rank, world, port, out_path):
    """inference_interpolation(): every rank converts its shard of the `interpolate_segments` pieces and writes its time range."""
    import types
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MST_EMU_THREADS="2")
    torch.set_num_threads(1)
    from music_mixing_style_transfer_amd import _lib
    from music_mixing_style_transfer_amd.inference import style_transfer as st
    from music_mixing_style_transfer_amd.utils import synth
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from emu_binding import bind_emulator
    _lib.set_default_binding(bind_emulator(build=False))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    enc, tcn = _models()
    runner = object.__new__(st.Mixing_Style_Transfer_Inference)
    runner.args = types.SimpleNamespace(normalize_input=False, instruments=["drums", "bass"], segment_length=SEG, segment_length_ref=SEG,
                                        batch_size=2, save_each_inst=True, sample_rate=44100, interpolate_segments=5)
    runner.device = torch.device("cpu")
    runner.target_dir, runner.output_dir = "/data/", out_path + "/"
    runner.models = {"effects_encoder": enc, "mixing_converter": tcn}
    stems_in = torch.stack([synth.synth_audio((2, L_IN), seed=5), synth.synth_audio((2, L_IN), seed=7)])
    ref_a = torch.stack([synth.synth_audio((2, 4 * SEG - 7), seed=6), synth.synth_audio((2, 4 * SEG - 7), seed=8)])       # 4 segments
    ref_b = torch.stack([synth.synth_audio((2, SEG - 30), seed=16), synth.synth_audio((2, SEG - 30), seed=18)])           # encoded whole
    runner.data_loader = [(stems_in, ref_a, ref_b, "/data/song/")]
    runner.inference_interpolation()
    if world > 1:
        dist.destroy_process_group()


def test_interpolation_mode_sharded_equals_single(tmp_path, emu):
    from music_mixing_style_transfer_amd import _lib
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    prev = _lib._default
    try:
        _interp_worker(0, 1, 0, one)
    finally:
        _lib.set_default_binding(prev)
    mp.spawn(_interp_worker, args=(2, 29679, two), nprocs=2, join=True)
    names = ["bass_output_notnormed_interpolation.wav", "drums_output_notnormed_interpolation.wav", "mixture_output_notnormed_interpolation.wav"]
    assert sorted(os.listdir(os.path.join(one, "song"))) == sorted(os.listdir(os.path.join(two, "song"))) == names
    for k in names:
        assert open(os.path.join(one, "song", k), "rb").read() == open(os.path.join(two, "song", k), "rb").read(), k
