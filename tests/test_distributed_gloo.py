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


def _models():
    from music_mixing_style_transfer_amd.networks import FXencoder, TCNModel
    from music_mixing_style_transfer_amd.utils import synth
    enc = FXencoder({k: (list(v) if isinstance(v, list) else v) for k, v in ENC_CFG.items()})
    enc.load_state_dict(synth.fxencoder_state_dict(ENC_CFG, seed=1))
    tcn = TCNModel(nparams=64, ninputs=2, noutputs=2, nblocks=3, dilation_growth=2, kernel_size=15, channel_width=128,
                   stack_size=15, cond_dim=64, causal=False)
    tcn.load_state_dict(synth.tcn_state_dict(nblocks=3, cond_dim=64, seed=2))
    return enc.eval(), tcn.eval()


def _worker(rank, world, port, out_path):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MST_EMU_THREADS="2")
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
    eng = StyleTransferEngine(enc, tcn)
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
