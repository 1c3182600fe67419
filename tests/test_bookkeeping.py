"""Segment bookkeeping: product (inference/segmentation.py) vs oracle (oracle/segmentation_ref.py) vs tables
recorded from the reference's batchwise_segmentization.  Integer work: bit-exact."""
import os

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from music_mixing_style_transfer_amd.inference import segmentation as S
from oracle import segmentation_ref as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_tables_from_reference():
    g = np.load(os.path.join(GOLD, "bookkeeping.npz"))
    for L, seg, bs, pad, n_seg, n_batches, last in g["table"]:
        plan = O.segment_plan(int(L), int(seg), int(bs))
        assert (plan["pad"], plan["n_seg"], len(plan["batch_sizes"]), plan["batch_sizes"][-1]) == (pad, n_seg, n_batches, last)
        song = torch.arange(2 * L, dtype=torch.float32).reshape(2, int(L)) + 1.0
        batches = S.batchwise_segmentization(song, "s", int(seg), int(bs))
        assert (sum(b.shape[0] for b in batches), len(batches), batches[-1].shape[0]) == (n_seg, n_batches, last)
        if seg == 8 and bs == 3:
            cat = torch.cat([torch.cat(torch.unbind(b, 0), -1) for b in batches], -1)
            assert np.array_equal(cat.numpy(), g[f"cat_L{L}"])
    assert int(g["assert_short"]) == 1


def test_exact_multiple_gets_an_extra_zero_segment():
    song = torch.ones(2, 16)
    b = S.batchwise_segmentization(song, "s", 8, 4)
    assert b[0].shape == (3, 2, 8) and float(b[0][2].abs().sum()) == 0.0


def test_duration_assert_uses_min_length():
    with pytest.raises(AssertionError, match="Insufficient duration"):
        S.batchwise_segmentization(torch.zeros(2, 10), "s", 4, 2, min_length=16)
    with pytest.raises(AssertionError):
        O.segment_plan(10, 4, 2, min_length=16)


@settings(max_examples=60, deadline=None)
@given(L=st.integers(1, 300), seg=st.integers(1, 40), bs=st.integers(1, 7))
def test_product_equals_oracle_and_roundtrips(L, seg, bs):
    if L < seg:
        return
    x = torch.arange(2 * L, dtype=torch.float32).reshape(2, L) + 1.0
    pb = S.batchwise_segmentization(x, "s", seg, bs)
    ob = O.batchwise_segmentization(x.numpy(), seg, bs)
    assert len(pb) == len(ob)
    for p, o in zip(pb, ob):
        assert np.array_equal(p.numpy(), o)
    assert torch.equal(S.reassemble(pb, L), x)                  # identity "model": concat + crop restores the stem
    assert np.array_equal(O.reassemble(ob, L), x.numpy())


def test_thresholds():
    seg = 8
    assert len(S.segment_input(torch.zeros(2, 8), "s", seg, 4)) == 1 and S.segment_input(torch.zeros(2, 8), "s", seg, 4)[0].shape == (1, 2, 8)
    assert S.segment_input(torch.zeros(2, 9), "s", seg, 4)[0].shape == (2, 2, 8)
    assert S.segment_reference(torch.zeros(2, 16), "s", seg, 4, 4)[0].shape == (1, 2, 16)     # not > 2*seg: unsegmented
    assert S.segment_reference(torch.zeros(2, 17), "s", seg, 4, 16)[0].shape == (5, 2, 4)      # cut by segment_length_ref
    for L in (8, 9, 16, 17, 61):
        a = S.segment_reference(torch.zeros(2, L), "s", seg, 4, 3)
        b = O.reference_batches(np.zeros((2, L), np.float32), seg, 4, 3)
        assert [tuple(t.shape) for t in a] == [t.shape for t in b]


def test_stack_embeddings_ragged_raises_like_reference():
    with pytest.raises(RuntimeError):
        S.stack_embeddings([torch.zeros(3, 4), torch.zeros(2, 4)])
    assert S.stack_embeddings([torch.zeros(3, 4), torch.ones(3, 4)]).shape == (6, 4)


def test_shard_range_partitions():
    for n in (0, 1, 7, 16, 1212):
        for w in (1, 2, 3, 8):
            r = [S.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_inference_and_interpolation_orchestration_match_reference(emu_default, monkeypatch):
    """Rows a-A8 / f-2: inference_interpolation against the REFERENCE's own run (tests/golden/interp.npz: the reference method
    executed with closed-form stand-in networks): L//S+1 input segmentation, blend weight per BATCH index, reference B cut by
    segment_length, stack/mean of segment embeddings, concatenation, crop, per-stem and mixture outputs."""
    import types
    import torch
    from music_mixing_style_transfer_amd.inference import style_transfer as st
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "interp.npz"))

    class Enc:
        def __call__(self, x):
            return torch.cat([x.mean(-1), x.abs().mean(-1), (x * x).mean(-1)], dim=1)

    class Conv:
        """the golden run's closed-form converter; like the real TCN it takes one condition row for the batch or one row per item (the
        engine converts all segments of a stem in one pass, every segment with the row of ITS batch of the reference's loop)"""

        def __call__(self, x, cond):
            c = cond.expand(x.shape[0], -1) if cond.shape[0] == 1 else cond
            col = lambda k: c[:, k][:, None, None]
            return x * (1.0 + col(0) - 0.5 * col(3)) + 0.1 * col(1) - 0.2 * col(4) + 0.05 * (col(2) + col(5)) * torch.flip(x, dims=(1,))

    written = {}
    monkeypatch.setattr(st, "save_wav_pcm16", lambda path, data, sr: written.__setitem__(os.path.basename(path), np.array(data)))
    monkeypatch.setattr(st.os, "makedirs", lambda *a, **k: None)
    for ci in (0, 1):
        L, La, Lb, S, seg_len, seg_ref, bs = (int(v) for v in g[f"c{ci}_cfg"])
        args = types.SimpleNamespace(normalize_input=False, instruments=["drums", "bass", "other", "vocals"], interpolate_segments=S,
                                     segment_length=seg_len, segment_length_ref=seg_ref, batch_size=bs, save_each_inst=True,
                                     sample_rate=44100)
        runner = object.__new__(st.Mixing_Style_Transfer_Inference)
        runner.args, runner.device = args, torch.device("cpu")
        runner.target_dir, runner.output_dir = "/data/", "/tmp/mst_interp_out/"
        runner.models = {"effects_encoder": Enc(), "mixing_converter": Conv()}
        runner.data_loader = [(torch.from_numpy(g[f"c{ci}_input"]), torch.from_numpy(g[f"c{ci}_ref_a"]),
                               torch.from_numpy(g[f"c{ci}_ref_b"]), "/data/song/")]
        written.clear()
        runner.inference_interpolation()
        names = [k[len(f"c{ci}_"):] for k in g.files if k.startswith(f"c{ci}_") and k.endswith(".wav")]
        assert sorted(written) == sorted(names) and len(names) == 5
        for name in names:
            ref = g[f"c{ci}_{name}"]
            assert written[name].shape == ref.shape == (L, 2)
            # same bookkeeping and weights; the mean embedding is summed in canonical row order on the device path
            assert np.abs(written[name] - ref).max() <= 2e-6
    # the plain inference() loop (style_transfer.py:112-177): segmented and unsegmented input / reference
    for ni in (0, 1):
        L, Lr, seg_len, seg_ref, bs = (int(v) for v in g[f"n{ni}_cfg"])
        args = types.SimpleNamespace(normalize_input=False, instruments=["drums", "bass", "other", "vocals"], segment_length=seg_len,
                                     segment_length_ref=seg_ref, batch_size=bs, save_each_inst=True, sample_rate=44100)
        runner = object.__new__(st.Mixing_Style_Transfer_Inference)
        runner.args, runner.device = args, torch.device("cpu")
        runner.target_dir, runner.output_dir = "/data/", "/tmp/mst_interp_out/"
        runner.models = {"effects_encoder": Enc(), "mixing_converter": Conv()}
        runner.data_loader = [(torch.from_numpy(g[f"n{ni}_input"]), torch.from_numpy(g[f"n{ni}_ref"]), "/data/song/")]
        written.clear()
        runner.inference()
        names = [k[len(f"n{ni}_"):] for k in g.files if k.startswith(f"n{ni}_") and k.endswith(".wav")]
        assert sorted(written) == sorted(names) and len(names) == 5
        for name in names:
            assert np.abs(written[name] - g[f"n{ni}_{name}"]).max() <= 2e-6


def test_save_args_record(tmp_path):
    """The configuration record the reference writes next to its outputs (style_transfer.py:305-322): same file name, one
    block per argument group, one '- name: value' line per argument."""
    from music_mixing_style_transfer_amd.inference import style_transfer as stm
    args = stm.build_parser().parse_args(["--target_dir", "/data/", "--batch_size", "3", "--do_not_separate", "True"])
    runner = object.__new__(stm.Mixing_Style_Transfer_Inference)
    runner.output_dir = str(tmp_path) + "/"
    runner.save_args(args)
    txt = open(tmp_path / "style_transfer_inference_configurations.txt").read()
    assert txt.startswith("\n[args]\n  Directory args (") and "  Inference args (" in txt and "  Device args (" in txt
    assert "      - batch_size          : 3\n" in txt and "      - target_dir          : /data/\n" in txt
    assert "      - segment_length      : 524288\n" in txt
