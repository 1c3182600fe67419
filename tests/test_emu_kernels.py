"""The gfx950 kernel sources + C-ABI host code, compiled for the host against the SIMT emulator (tests/emu),
checked against the oracle.  Catches index math / MFMA fragment layout / LDS swizzle / packing bugs without a
GPU.  Same sources as libmst_hip.so, same C ABI, driven through the same module API."""
import ctypes as C

import numpy as np
import pytest
import torch

from music_mixing_style_transfer_amd.utils import synth
from oracle import fx_ref as F
from oracle import networks_ref as R


def _tcn(nblocks, cond_dim=64, growth=2, seed=0):
    from music_mixing_style_transfer_amd.networks import TCNModel
    sd = synth.tcn_state_dict(nblocks=nblocks, cond_dim=cond_dim, seed=seed)
    m = TCNModel(nparams=cond_dim, ninputs=2, noutputs=2, nblocks=nblocks, dilation_growth=growth, kernel_size=15,
                 channel_width=128, stack_size=15, cond_dim=cond_dim, causal=False)
    m.load_state_dict(sd)
    return m, sd


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("bf16", 4e-2), ("bf16x3", 3e-5)])
def test_tcn_blocks_emulated(emu_default, prec, tol):
    m, sd = _tcn(4)
    m.precision = prec
    x = synth.synth_audio((2, 2, 777), seed=1)
    cond = synth.synth_audio((1, 64), seed=2)
    col = []
    y_ref = R.tcn_forward(sd, x, cond, nblocks=4, collect=col)
    for n in (1, 2, 4):
        a = m.forward_blocks(x, cond, n)
        assert float((a - col[n - 1]).abs().max()) <= tol * float(col[n - 1].abs().max())
    y = m(x, cond)
    assert float((y - y_ref).abs().max()) <= tol
    assert float(y.abs().max()) <= 1.0


def test_tcn_large_dilation_and_odd_growth_emulated(emu_default):
    """dilation 3**n exercises P = 1 tiles with non power-of-two d; a short segment exercises growing P."""
    m, sd = _tcn(4, growth=3)
    x = synth.synth_audio((1, 2, 300), seed=4)
    cond = synth.synth_audio((1, 64), seed=5)
    y_ref = R.tcn_forward(sd, x, cond, nblocks=4, dilation_growth=3)
    assert float((m(x, cond) - y_ref).abs().max()) <= 2e-5
    m2, sd2 = _tcn(6)          # d up to 32 on L = 200: tiles with P = 8 and 16
    x = synth.synth_audio((1, 2, 200), seed=6)
    y_ref = R.tcn_forward(sd2, x, cond, nblocks=6)
    assert float((m2(x, cond) - y_ref).abs().max()) <= 2e-5
    m2.precision = "bf16"
    assert float((m2(x, cond) - y_ref).abs().max()) <= 4e-2
    m2.precision = "bf16x3"          # split-bf16 mode: 8-phase tiles of 128 times for the large dilations
    y3 = m2(x, cond)
    assert float((y3 - y_ref).abs().max()) <= 3e-5
    # 256-time tiles instead of 128-time ones: the same products; the two-phase 128-time tiles sum them class-major (B fragments reused by the
    # two taps of a class), the 256-time tiles tap-major - equal to fp32 accumulation rounding
    emu_default.check(emu_default.mst_tcn_set_tuning(m2._handle, 0), "tuning")
    assert float((m2(x, cond) - y3).abs().max()) <= 1e-5
    x5 = synth.synth_audio((2, 2, 700), seed=16)
    y5 = m2(x5, cond)
    y5_ref = R.tcn_forward(sd2, x5, cond, nblocks=6)
    assert float((y5 - y5_ref).abs().max()) <= 3e-5
    emu_default.check(emu_default.mst_tcn_set_tuning(m2._handle, 1), "tuning")
    y5s = m2(x5, cond)
    assert float((y5s - y5).abs().max()) <= 1e-5 and float((y5s - y5_ref).abs().max()) <= 3e-5


def test_tcn_bf16_whole_sequence_tiles_emulated(emu_default):
    """bf16 tiles of 8 phases x 16 steps that span their WHOLE phase sequence (d = 8 on 121 ... 128 samples: 16 steps per phase; at
    L = 131072 that is the d = 8192 block) run the unrolled class-major loop in which the (column tile, tap) pairs that only see zero padding do not
    exist.  Dropping a live pair would be an O(0.1) error: checked against the oracle at the bf16 tolerance, block by block, for the plain
    epilogue (block 3 of 5) and the fused output head (block 3 of 4), full and ragged last steps."""
    cond = synth.synth_audio((2, 64), seed=21)
    # (d = 8 on 249 ... 256 samples: 32 steps per phase = four-phase 128-time tiles that span their sequence: the d = 4096 block at L = 131072;
    #  200 samples: the same tiles not spanning it - the tap-major loop)
    for nb, L in ((4, 128), (4, 123), (5, 128), (5, 121), (4, 256), (5, 250), (5, 200)):
        m, sd = _tcn(nb)
        m.precision = "bf16"
        x = synth.synth_audio((2, 2, L), seed=30 + L)
        col = []
        y_ref = R.tcn_forward(sd, x, cond, nblocks=nb, collect=col)
        assert float((m(x, cond) - y_ref).abs().max()) <= 4e-2
        a = m.forward_blocks(x, cond, 4)
        assert float((a - col[3]).abs().max()) <= 4e-2 * float(col[3].abs().max())


def test_tcn_bf16_duo_kernel_emulated(emu_default):
    """The persistent LDS-DMA-fed form of the bf16 block kernel (mst_tcn_set_tuning bits 1-2 = 2, "duo") in its tap-major order (bit 4 off): the
    one-tile-per-workgroup kernel's arithmetic in the same order - bit-identical - and the oracle at the bf16 tolerance.  The emulated grid
    has 8 workgroups: several tiles per workgroup, XCD-ordered tile ranges, every phase count P, per-item FiLM rows, the fused output head.
    (Form 1, the "stream" kernel, and bit 3, the split-bf16 duo kernel, left the library in round 5: the setter rejects them.)"""
    cond = synth.synth_audio((1, 64), seed=2)
    cases = [(4, 2, (2, 2, 777), cond),                                   # P = 2, 4, 8; 2 x 8 tiles over 8 workgroups
             (4, 3, (1, 2, 300), cond),                                   # odd dilations: P = 1
             (6, 2, (1, 2, 200), cond),                                   # short segment: P = 8 / 16 tiles, zero rows, skipped column tiles
             (3, 2, (3, 2, 1500), synth.synth_audio((3, 64), seed=11))]   # 3 x 6 tiles: uneven walks, one FiLM row per item
    for nb, growth, shape, cnd in cases:
        m, sd = _tcn(nb, growth=growth)
        m.precision = "bf16"
        x = synth.synth_audio(shape, seed=1)
        col = []
        y_ref = R.tcn_forward(sd, x, cnd, nblocks=nb, dilation_growth=growth, collect=col)
        m._ensure(emu_default)
        emu_default.check(emu_default.mst_tcn_set_tuning(m._handle, 1), "tuning")           # form 0: one tile per workgroup
        y0 = m(x, cnd)
        a0 = m.forward_blocks(x, cnd, nb)
        assert float((y0 - y_ref).abs().max()) <= 4e-2
        assert float((a0 - col[nb - 1]).abs().max()) <= 4e-2 * float(col[nb - 1].abs().max())
        # form 2 ("duo": 4 matrix + 4 loader waves per CU, two tile buffers, the next tile by LDS-DMA during the main loop), tap-major:
        # the same bits as form 0
        emu_default.check(emu_default.mst_tcn_set_tuning(m._handle, 5), "tuning")
        assert torch.equal(m(x, cnd), y0) and torch.equal(m.forward_blocks(x, cnd, nb), a0)
    for gone in (3, 6, 13):          # the stream kernel (form 1), form 3, the split-bf16 duo kernel (bit 3)
        with pytest.raises(ValueError):
            emu_default.check(emu_default.mst_tcn_set_tuning(m._handle, gone), "tuning")


def test_tcn_bf16_duo_reuse_main_loop_emulated(emu_default):
    """The class-major main loop of the duo kernel's four-phase tiles (mst_tcn_set_tuning bit 4): every B fragment is read from LDS once per
    class of taps (j mod 4) and feeds up to eight MFMAs; same products as the tap-major loop, summed in another order - against the oracle at
    the bf16 tolerance, against the tap-major form to accumulation rounding.  d = 4 ... 32 at lengths that give four-phase tiles: several
    tiles per workgroup, ragged last tiles (zero rows), tiles of more than one phase group, one FiLM row per item."""
    cases = [(3, 2, (3, 2, 1500), synth.synth_audio((3, 64), seed=11)),        # d = 2, 4: 3 x 6 four-phase tiles
             (5, 2, (1, 2, 2100), synth.synth_audio((1, 64), seed=2)),         # d = 2 ... 16: phase groups 1, 2, 4
             (3, 2, (2, 2, 777), synth.synth_audio((1, 64), seed=3))]
    differs = 0
    for nb, growth, shape, cnd in cases:
        m, sd = _tcn(nb, growth=growth)
        m.precision = "bf16"
        x = synth.synth_audio(shape, seed=1)
        col = []
        y_ref = R.tcn_forward(sd, x, cnd, nblocks=nb, dilation_growth=growth, collect=col)
        m._ensure(emu_default)
        emu_default.check(emu_default.mst_tcn_set_tuning(m._handle, 5), "tuning")            # tap-major
        y0, a0 = m(x, cnd), m.forward_blocks(x, cnd, nb)
        emu_default.check(emu_default.mst_tcn_set_tuning(m._handle, 21), "tuning")           # class-major
        y1, a1 = m(x, cnd), m.forward_blocks(x, cnd, nb)
        assert float((y1 - y_ref).abs().max()) <= 4e-2
        assert float((a1 - col[nb - 1]).abs().max()) <= 4e-2 * float(col[nb - 1].abs().max())
        assert float((y1 - y0).abs().max()) <= 5e-3 and float((a1 - a0).abs().max()) <= 5e-2
        differs += int(not torch.equal(a1, a0))
    assert differs > 0          # the other summation order did run (bit-identical activations everywhere would mean the flag was ignored)


def test_tcn_bf16_one_tile_class_major_256_time_tiles_emulated(emu_default):
    """mst_tcn_set_tuning bit 7 (round 6, default): the four-phase class-major blocks on the ONE-TILE kernel's 256-time tiles (two workgroups
    per CU) instead of the duo kernel - the duo kernel's products in the duo kernel's order: bit-identical, several tiles per phase sequence,
    ragged last tiles, per-item FiLM rows; and against the oracle at the bf16 tolerance."""
    cases = [(4, (2, 2, 1500), synth.synth_audio((2, 64), seed=11)),        # d = 4, 8: 6 / 3 tiles per sequence, the last ragged
             (5, (1, 2, 2100), synth.synth_audio((1, 64), seed=2))]         # d = 4, 8, 16 (d = 16: 132 steps = 3 tiles, the last ragged)
    for nb, shape, cnd in cases:
        m, sd = _tcn(nb)
        m.precision = "bf16"
        x = synth.synth_audio(shape, seed=1)
        y_ref = R.tcn_forward(sd, x, cnd, nblocks=nb)
        m._ensure(emu_default)
        emu_default.check(emu_default.mst_tcn_set_tuning(m._handle, 21), "tuning")           # duo, class-major
        y0, a0 = m(x, cnd), m.forward_blocks(x, cnd, nb - 1)
        emu_default.check(emu_default.mst_tcn_set_tuning(m._handle, 21 | 128), "tuning")     # + bit 7
        fl = C.c_int(0)
        emu_default.check(emu_default.mst_tcn_get_tuning(m._handle, C.byref(fl), None), "get")
        assert fl.value == 21 | 128
        y1, a1 = m(x, cnd), m.forward_blocks(x, cnd, nb - 1)
        assert torch.equal(a0, a1)
        # the waveform: the LAST block (four phases here, fused head) runs the one-tile kernel in both settings - tap-major with bit 7 off, class-major
        # with it: the same products in another fp32 order, re-rounded to bf16 in front of the head
        assert float((y1 - y0).abs().max()) <= 5e-3
        assert float((y1 - y_ref).abs().max()) <= 4e-2
        aN0 = m.forward_blocks(x, cnd, nb)
        emu_default.check(emu_default.mst_tcn_set_tuning(m._handle, 21), "tuning")
        assert float((m.forward_blocks(x, cnd, nb) - aN0).abs().max()) <= 2.0 ** -7 * float(aN0.abs().max())
    assert emu_default.mst_tcn_set_tuning(m._handle, 256) != 0          # no flag bits beyond bit 7


def test_tcn_bf16_whole_sequence_256_time_tiles_emulated(emu_default):
    """Round 6 (with mst_tcn_set_tuning bit 7): a block whose phase sequences are EXACTLY one 256-time tile - 64 steps (four phases), 32 (eight) or 16
    (sixteen: d = 2048 / 4096 / 8192 at L = 131072) - runs the unrolled class-major loop on an LDS image that keeps only the halo steps a row window
    can straddle into (3 / 1 / 0): all-padding windows are neither staged nor read; the sixteen-phase form also with the fused output head (the last
    block).  At L = 512: d = 8, 16, 32.  The four-phase form sums in the duo kernel's order (bit-identical to bit 7 off); the other two agree with
    round 5's 128-time forms to accumulation rounding (one bf16 ulp on the activation); all within the bf16 tolerance of the oracle."""
    for nb, shape, cnd in [(6, (3, 2, 512), synth.synth_audio((3, 64), seed=11)),        # d = 32 is the last block: fused head
                           (5, (16, 2, 512), synth.synth_audio((1, 64), seed=2))]:      # d = 16 (eight phases) is the last block: its head runs round 5's form
        m, sd = _tcn(nb)
        m.precision = "bf16"
        x = synth.synth_audio(shape, seed=1)
        col = []
        y_ref = R.tcn_forward(sd, x, cnd, nblocks=nb, collect=col)
        m._ensure(emu_default)
        out = {}
        for flags in (53, 53 | 128):
            emu_default.check(emu_default.mst_tcn_set_tuning(m._handle, flags), "tuning")
            out[flags] = [m(x, cnd)] + [m.forward_blocks(x, cnd, n) for n in range(3, nb + 1)]
        assert float((out[181][0] - y_ref).abs().max()) <= 4e-2 and float((out[181][0] - out[53][0]).abs().max()) <= 5e-3
        differs = 0
        for k, n in enumerate(range(3, nb + 1)):
            a1, a0, r = out[181][1 + k], out[53][1 + k], col[n - 1]
            assert float((a1 - r).abs().max()) <= 4e-2 * float(r.abs().max()), n
            if n <= 4:          # d = 4 (several tiles per sequence), d = 8 (the whole-sequence four-phase tile): the duo kernel's order
                assert torch.equal(a1, a0), n
            else:
                assert float((a1 - a0).abs().max()) <= 2.0 ** -6 * float(a0.abs().max()), n
                differs += int(not torch.equal(a1, a0))
        assert differs > 0          # (the other summation order really ran)


def test_tcn_bf16_block0_fused_into_block1_emulated(emu_default):
    """mst_tcn_set_tuning bit 5 (default): block 0 is not launched - the d = 2 block computes its input rows from the waveform with
    tcn_block0_mfma_kernel's arithmetic, in the loader waves of the duo kernel (bit 7 off) or in the staging of the one-tile kernel (bit 7, the
    default since round 6): the same bits as the separate kernel on every activation and on the waveform, in both forms.  Several tiles per
    workgroup (the duo kernel's buffers are refilled), ragged lengths (zero rows on both sides, a last tile mostly outside the segment), segments
    shorter than a tile, per-item FiLM rows; probes of block 0 alone stay on the separate kernel."""
    cases = [(4, (2, 2, 777), synth.synth_audio((1, 64), seed=2)),
             (3, (3, 2, 1500), synth.synth_audio((3, 64), seed=11)),
             (3, (2, 2, 41), synth.synth_audio((2, 64), seed=4))]          # (more shapes: tools/emu_sweep_tcn.py --fuse0)
    for nb, shape, cnd in cases:
        m, sd = _tcn(nb)
        m.precision = "bf16"
        x = synth.synth_audio(shape, seed=1)
        m._ensure(emu_default)
        emu_default.check(emu_default.mst_tcn_set_tuning(m._handle, 21), "tuning")
        y0 = m(x, cnd)
        a0 = [m.forward_blocks(x, cnd, n) for n in (1, 2, nb - 1)]
        for flags in (21 | 32, 21 | 32 | 128, 21 | 128):          # fused in the duo kernel / in the one-tile kernel; the one-tile kernel unfused
            emu_default.check(emu_default.mst_tcn_set_tuning(m._handle, flags), "tuning")
            y1 = m(x, cnd)
            fl, fused = C.c_int(0), C.c_int(0)
            emu_default.check(emu_default.mst_tcn_get_tuning(m._handle, C.byref(fl), C.byref(fused)), "get")
            assert fl.value == flags and fused.value == (1 if flags & 32 else 0)
            a1 = [m.forward_blocks(x, cnd, n) for n in (1, 2, nb - 1)]
            for u, v in zip(a0, a1):
                assert torch.equal(u, v), (shape, flags)
            if flags & 128:          # bit 7 also moves the last block (fused head) from the tap-major to the class-major order: rounding
                assert float((y1 - y0).abs().max()) <= 5e-3, (shape, flags)
            else:
                assert torch.equal(y1, y0), (shape, flags)
        y_ref = R.tcn_forward(sd, x, cnd, nblocks=nb)
        assert float((y1 - y_ref).abs().max()) <= 4e-2
    # without the class-major duo form there is nothing to fuse into: the flag is ignored, block 0 runs on its own
    emu_default.check(emu_default.mst_tcn_set_tuning(m._handle, 1 | 32), "tuning")
    assert float((m(x, cnd) - y_ref).abs().max()) <= 4e-2


def test_tcn_bf16x3_half_tile_kernel_class_major_emulated(emu_default):
    """mst_tcn_set_tuning bit 6 (experimental, off by default): the eight-phase half-tile kernel of the split-bf16 mode (d a multiple of 8 with
    fewer than 64 steps per phase) with the class-major loop - the oracle at the mode's tolerance, bit 6 off to accumulation rounding.  Tiles
    that span their phase sequence, several tiles per sequence (first / last ones with all-padding tap tiles skipped), ragged lengths,
    the fused output head (last block) and the plain epilogue, per-item FiLM rows."""
    cases = [(6, (1, 2, 200), synth.synth_audio((1, 64), seed=5)),          # d = 8, 16, 32 on 25 / 13 / 7 steps
             (4, (2, 2, 400), synth.synth_audio((2, 64), seed=6)),          # d = 8: 50 steps = four tiles per sequence; the last block
             (5, (3, 2, 131), synth.synth_audio((1, 64), seed=7))]          # d = 8, 16: 17 / 9 steps  (more: tools/emu_sweep_tcn.py --tuning 85)
    for nb, shape, cnd in cases:
        m, sd = _tcn(nb)
        m.precision = "bf16x3"
        x = synth.synth_audio(shape, seed=3)
        col = []
        y_ref = R.tcn_forward(sd, x, cnd, nblocks=nb, collect=col)
        m._ensure(emu_default)
        emu_default.check(emu_default.mst_tcn_set_tuning(m._handle, 21), "tuning")
        y0, a0 = m(x, cnd), m.forward_blocks(x, cnd, nb)
        emu_default.check(emu_default.mst_tcn_set_tuning(m._handle, 21 | 64), "tuning")
        y1, a1 = m(x, cnd), m.forward_blocks(x, cnd, nb)
        assert float((y1 - y_ref).abs().max()) <= 3e-5
        assert float((a1 - col[nb - 1]).abs().max()) <= 3e-5 * float(col[nb - 1].abs().max())
        assert float((y1 - y0).abs().max()) <= 1e-5 and float((a1 - a0).abs().max()) <= 1e-5 * float(a0.abs().max())
        assert not torch.equal(a1, a0)          # the other loop did run


def test_tcn_condition_forms_emulated(emu_default):
    m, sd = _tcn(2)
    x = synth.synth_audio((2, 2, 260), seed=7)
    condB = synth.synth_audio((2, 64), seed=8)
    condL = [synth.synth_audio((1, 64), seed=9 + i) for i in range(2)]
    for cond in (condB, condL):
        assert float((m(x, cond) - R.tcn_forward(sd, x, cond, nblocks=2)).abs().max()) <= 2e-5
    with pytest.raises(RuntimeError):
        m(x, synth.synth_audio((3, 64), seed=1))          # rows must be 1 or B, as in torch broadcasting


def test_encoder_emulated(emu_default):
    from music_mixing_style_transfer_amd.networks import FXencoder
    cfg = {"channels": [4, 40, 72, 136], "kernels": [5, 4, 3, 10], "strides": [2, 2, 1, 2], "dilation": [1, 1, 1, 1],
           "bias": True, "norm": "batch", "conv_block": "res", "activation": "relu"}
    sd = synth.fxencoder_state_dict(cfg, seed=3)
    user_cfg = {k: (list(v) if isinstance(v, list) else v) for k, v in cfg.items()}
    enc = FXencoder(user_cfg)
    assert user_cfg["channels"][0] == 2        # the reference mutates the caller's list (architectures.py:30)
    enc.load_state_dict(sd)
    x = synth.synth_audio((3, 2, 333), seed=11)
    col = []
    R.fxencoder_blocks(x, sd, cfg, collect=col)
    for n in range(1, 5):
        a = enc.forward_blocks(x, n)
        assert a.shape == col[n - 1].shape
        assert float((a - col[n - 1]).abs().max()) <= 2e-5
    assert float((enc(x) - R.fxencoder_forward(sd, cfg, x)).abs().max()) <= 2e-5
    enc.precision = "bf16"                    # bf16 MFMA operands, fp32 accumulate, fp32 activations
    for n in (1, 2, 4):
        a = enc.forward_blocks(x, n)
        assert float((a - col[n - 1]).abs().max()) <= 3e-2 * float(col[n - 1].abs().max())
    assert float((enc(x) - R.fxencoder_forward(sd, cfg, x)).abs().max()) <= 3e-2
    enc.precision = "fp32"
    blk = enc.encoder[0]                      # a Res_ConvBlock runs stand-alone too
    assert float((blk(x) - col[0]).abs().max()) <= 2e-5
    with pytest.raises(ValueError):
        enc(synth.synth_audio((1, 2, 3), seed=1))          # reflection padding longer than the input


def test_encoder_nlc_bf16_pipeline_emulated(emu_default):
    """bf16 mode on a configs.yaml-like net (channels % 8 == 0): direct kernel for the stereo block, channel-minor
    bf16 activations afterwards, split-K + finalize on the short late layers."""
    from music_mixing_style_transfer_amd.networks import FXencoder
    cfg = {"channels": [16, 40, 72, 136, 264], "kernels": [25, 5, 4, 3, 10], "strides": [4, 2, 2, 1, 2],
           "dilation": [1, 1, 1, 1, 1], "bias": True, "norm": "batch", "conv_block": "res", "activation": "relu"}
    sd = synth.fxencoder_state_dict(cfg, seed=3)
    enc = FXencoder({k: (list(v) if isinstance(v, list) else v) for k, v in cfg.items()})
    enc.load_state_dict(sd)
    x = synth.synth_audio((3, 2, 1333), seed=11)
    col = []
    R.fxencoder_blocks(x, sd, cfg, collect=col)
    enc.precision = "bf16"
    for n in range(1, 6):
        a = enc.forward_blocks(x, n)
        assert a.shape == col[n - 1].shape
        assert float((a - col[n - 1]).abs().max()) <= 2e-2 * float(col[n - 1].abs().max())
    assert float((enc(x) - R.fxencoder_forward(sd, cfg, x)).abs().max()) <= 2e-2
    # bf16x3: the same pipeline in split mode (every activation as two bf16 planes x = hi + lo, folded weights likewise, three MFMAs
    # per product): fp32-class accuracy on the bf16 matrix cores
    enc.precision = "bf16x3"
    for n in range(1, 6):
        a = enc.forward_blocks(x, n)
        assert a.shape == col[n - 1].shape
        assert float((a - col[n - 1]).abs().max()) <= 5e-5 * float(col[n - 1].abs().max()), n
    e_ref = R.fxencoder_forward(sd, cfg, x)
    assert float((enc(x) - e_ref).abs().max()) <= 2e-5 * float(e_ref.abs().max())


def test_tiny_reference_goldens_through_the_product(emu_default):
    """The tiny FXencoder / TCNModel configurations whose outputs were recorded from the REAL reference run through the
    product's generic exact-fp32 path (any channel width / kernel size): product vs reference directly."""
    import os
    from music_mixing_style_transfer_amd.networks import FXencoder, TCNModel
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "nets_tiny.npz"))
    cfg = {"channels": [4, 8, 8], "kernels": [5, 4, 3], "strides": [2, 2, 1], "dilation": [1, 1, 1], "bias": True,
           "norm": "batch", "conv_block": "res", "activation": "relu"}
    enc = FXencoder({k: (list(v) if isinstance(v, list) else v) for k, v in cfg.items()})
    enc.load_state_dict(synth.fxencoder_state_dict(cfg, seed=3))
    assert float((enc(torch.from_numpy(g["tiny_enc_x"])) - torch.from_numpy(g["tiny_enc_out"])).abs().max()) <= 2e-6
    tcn = TCNModel(nparams=16, ninputs=2, noutputs=2, nblocks=4, dilation_growth=2, kernel_size=5, channel_width=8,
                   stack_size=15, cond_dim=16, causal=False)
    tcn.load_state_dict(synth.tcn_state_dict(nblocks=4, kernel_size=5, channel_width=8, cond_dim=16, seed=5))
    x = torch.from_numpy(g["tiny_tcn_x"])
    for name, cond in (("", torch.from_numpy(g["tiny_tcn_cond"])), ("_condB", torch.from_numpy(g["tiny_tcn_condB"])),
                       ("_condL", [torch.from_numpy(c) for c in g["tiny_tcn_condL"]])):
        assert float((tcn(x, cond) - torch.from_numpy(g["tiny_tcn_out" + name])).abs().max()) <= 2e-6
    # mono, 48 channels, k = 3, dilation 3**n
    sd = synth.tcn_state_dict(nblocks=3, ninputs=1, noutputs=1, kernel_size=3, channel_width=48, cond_dim=32, seed=7)
    m = TCNModel(nparams=32, ninputs=1, noutputs=1, nblocks=3, dilation_growth=3, kernel_size=3, channel_width=48,
                 stack_size=15, cond_dim=32, causal=False)
    m.load_state_dict(sd)
    xm, cm = synth.synth_audio((2, 1, 333), seed=1), synth.synth_audio((1, 32), seed=2)
    assert float((m(xm, cm) - R.tcn_forward(sd, xm, cm, nblocks=3, kernel_size=3, dilation_growth=3)).abs().max()) <= 2e-6


def test_embedding_mean_and_engine_emulated(emu_default):
    from music_mixing_style_transfer_amd.inference import embedding_mean
    e = synth.synth_audio((7, 40), seed=2)
    assert float((embedding_mean(e) - e.mean(0)).abs().max()) <= 1e-6


def test_fx_emulated(emu_default):
    from music_mixing_style_transfer_amd.mixing_manipulator import (AugmentationChain, Compressor, Equaliser, Gain,
                                                                     MidSideImager)
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "fx.npz"))
    x = g["x"][:1500]
    c = Compressor(44100)
    for th, at, rt, ra in g["comp_cases"]:
        c.parameters.threshold.value, c.parameters.attack_time.value = th, at
        c.parameters.release_time.value, c.parameters.ratio.value = rt, ra
        assert np.abs(c.process(x.copy()) - F.compressor(x.copy(), th, at, rt, ra)).max() <= 2e-7
    # attack slower than release (outside the reference's parameter ranges): the chunk maps of the time-parallel smoother
    # are concave instead of convex; equal coefficients: linear
    for th, at, rt, ra in ((-25.0, 300.0, 40.0, 6.0), (-25.0, 80.0, 80.0, 3.0)):
        c.parameters.threshold.value, c.parameters.attack_time.value = th, at
        c.parameters.release_time.value, c.parameters.ratio.value = rt, ra
        assert np.abs(c.process(x.copy()) - F.compressor(x.copy(), th, at, rt, ra)).max() <= 2e-7
    eq = Equaliser(2, 44100)
    for band, (gg, fc, q) in F.CONFIG4["eq"].items():
        getattr(eq.parameters, band + "_gain").value = gg
    assert np.abs(eq.process(x.copy()) - F.equaliser(x.copy(), F.CONFIG4["eq"])).max() <= 1e-7
    # more than 255 chunks: the 512-element scan over the chunk states (20000 samples -> 313 chunks of 64), ragged last chunk
    xl = (0.1 * np.random.default_rng(5).standard_normal((20000, 2))).astype(np.float32)
    assert np.abs(eq.process(xl.copy()) - F.equaliser(xl.copy(), F.CONFIG4["eq"])).max() <= 1e-7
    im = MidSideImager()
    for bal in (0.0, 0.4567, 1.5, 2.0):
        im.parameters.bal.value = bal
        assert np.abs(im.process(x.copy()) - F.midside_imager(x.copy(), bal)).max() <= 2e-6
    # the config-4 chain through AugmentationChain (probability 1, fixed parameters) vs the oracle chain
    comp = Compressor(44100)
    im.parameters.bal.value = F.CONFIG4["imager_bal"]
    gn = Gain()
    gn.parameters.gain.value = F.CONFIG4["gain_db"]
    chain = AugmentationChain([(eq, 1.0, True), (comp, 1.0, True), (im, 1.0, True), (gn, 1.0, False)],
                              randomize_param_value=False)
    y = chain([x.copy()])[0]
    assert np.abs(y - F.fx_chain(x.copy())).max() <= 5e-6
    # tail folding: an rms-normalised imager followed by a Gain runs as ONE pass (MstFxFuse.post_rms) - the same bits as the imager's
    # chain followed by a separate gain call; inverted gain; an imager followed by something else runs first (flush)
    one = lambda fxs: AugmentationChain(fxs, randomize_param_value=False)([x.copy()])[0]
    for invert in (False, True):
        gn.parameters.invert.value = invert
        assert np.array_equal(one([(im, 1.0, True), (gn, 1.0, False)]), gn.process(one([(im, 1.0, True)])))
        assert np.array_equal(one([(eq, 1.0, True), (im, 1.0, True), (gn, 1.0, False)]), gn.process(one([(eq, 1.0, True), (im, 1.0, True)])))
    gn.parameters.invert.value = False
    ref1 = AugmentationChain([(gn, 1.0, True)], randomize_param_value=False)([one([(im, 1.0, True)])])[0]      # a normalised gain does not fold
    assert np.abs(one([(im, 1.0, True), (gn, 1.0, True)]) - ref1).max() <= 1e-6
    ref2 = AugmentationChain([(eq, 1.0, True)], randomize_param_value=False)([one([(im, 1.0, True)])])[0]
    assert np.abs(one([(im, 1.0, True), (eq, 1.0, True)]) - ref2).max() <= 1e-6
    # the equaliser's apply pass leaves sum(x^2) of its raw input behind (the first rms-normalise of a chain): mst_fx_sumsq's value
    from music_mixing_style_transfer_amd.mixing_manipulator import common_audioeffects as CA
    dd = CA._Dev(xl.copy())
    eq._run(dd, None, True, want_in_sumsq=True)
    s_eq, s_ref = float(dd.last_in_sumsq.sum()), float(CA._sumsq(dd, dd.x).sum())
    assert abs(s_eq - s_ref) <= 1e-12 * s_ref and abs(s_ref - float((xl.astype(np.float64) ** 2).sum())) <= 1e-6 * s_ref
    # tail folding is the imager's: the other entry points refuse a descriptor that asks for it
    import ctypes as C
    from music_mixing_style_transfer_amd import _lib
    t = torch.from_numpy(x.copy())[None].contiguous()
    y_t, q = torch.empty_like(t), torch.zeros(64, dtype=torch.float64)
    f = _lib.MstFxFuse(None, None, q.data_ptr(), 1, 2.0)
    assert emu_default.mst_fx_gain(t.data_ptr(), y_t.data_ptr(), 1, t.shape[1], 2, 0.0, 0, C.byref(f), None) == -2          # MST_ERR_UNSUPPORTED
    # batched [n_items, L, C] input
    xb = np.stack([x, 0.5 * x[::-1].copy()])
    yb = comp.process(xb)
    assert np.abs(yb[1] - F.compressor(xb[1].copy(), -20.0, 2.0, 100.0, 4.0)).max() <= 2e-7


def test_haas_panner_emulated(emu_default):
    """a-D7: bit-exact against the reference's own outputs (tests/golden/fx.npz) and the oracle."""
    import os
    from music_mixing_style_transfer_amd.mixing_manipulator import Haas, Panner
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fx.npz"))
    x = g["x"]
    hp = Haas(44100)
    assert hp.parameters.delay.value == 1764 and hp.parameters.feedback.value == 0.35
    for key, delay, fb, wet in (("haas_left", 37, 0.35, "left"), ("haas_right", -12, 0.5, "right")):
        hp.parameters.delay.value, hp.parameters.feedback.value, hp.parameters.wet_channel.value = delay, fb, wet
        y = hp.process(x.copy())
        assert y.dtype == np.float32 and np.array_equal(y, g[key])
    # delays beyond the signal length wrap like np.roll; mono input is repeated to stereo
    hp.parameters.delay.value = -3 * len(x) - 5
    assert np.array_equal(hp.process(x.copy()), F.haas(x.copy(), -3 * len(x) - 5, 0.5, "right"))
    mono = x[:, :1].copy()
    assert np.array_equal(hp.process(mono), F.haas(np.repeat(mono, 2, axis=1), -3 * len(x) - 5, 0.5, "right"))
    pn = Panner()
    for i, (pan, law) in enumerate(((0.3, "-4.5dB"), (0.8, "linear"), (0.5, "constant_power"))):
        pn.parameters.pan.value, pn.parameters.pan_law.value = pan, law
        pn.update()
        assert np.array_equal(pn.gains, g[f"pan_gains_{i}"]) and pn.gains.dtype == np.float32
        assert np.array_equal(pn.process(x.copy()), x * F.panner_gains(pan, law))
        assert np.array_equal(pn.process(mono), np.repeat(mono, 2, axis=1) * F.panner_gains(pan, law))
    pn.parameters.pan_law.value = "-3dB"
    with pytest.raises(ValueError):
        pn.update()
    with pytest.raises(AssertionError):
        hp.process(np.zeros((16, 3), np.float32))


def test_tcn_bf16_per_item_rows_emulated(emu_default):
    """bf16 mode, one FiLM row per batch item, against the oracle."""
    m, sd = _tcn(4)
    m.precision = "bf16"
    x = synth.synth_audio((2, 2, 500), seed=3)
    cB = synth.synth_audio((2, 64), seed=10)
    assert float((m(x, cB) - R.tcn_forward(sd, x, cB, nblocks=4)).abs().max()) <= 4e-2


def test_conv_reverb_emulated(emu_default):
    """f-3: ConvolutionalReverb through mst_fx_convolve (host FFT stand-ins in the emulator) vs the reference's outputs."""
    import os
    from music_mixing_style_transfer_amd.mixing_manipulator import ConvolutionalReverb
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fx_reverb.npz"))
    x, h2, h1 = g["x"], g["h_stereo"], g["h_mono"]
    irs = [[{"impulse_response": (lambda: h2)}], [{"impulse_response": (lambda: h1)}, {"impulse_response": (lambda: h2)}]]
    rv = ConvolutionalReverb(irs, 44100)
    rv.update()
    y = rv.process(x.copy())
    assert y.dtype == np.float32 and np.abs(y - g["y_stereo"]).max() <= 5e-6 * np.abs(g["y_stereo"]).max()
    rv.parameters.index.value, rv.parameters.index_ir.value = 1, 2
    rv.parameters.decay.value, rv.parameters.pre_delay.value = 0.5, 3
    rv.parameters.dry.value, rv.parameters.wet.value = 0.3, 0.7
    rv.update()
    assert np.array_equal(rv.h, g["h_mono_faded"])
    y = rv.process(x.copy())
    assert np.abs(y - g["y_mono_fade_predelay_mix"]).max() <= 5e-6 * np.abs(g["y_mono_fade_predelay_mix"]).max()
    y = rv.process(x[:, :1].copy())
    assert y.shape == (len(x), 1) and np.abs(y - g["y_mono_input"]).max() <= 5e-6 * np.abs(g["y_mono_input"]).max()
    # batched items share the impulse response; wet = 0 returns the input
    xb = np.stack([x, 0.5 * x[::-1].copy()])
    rv2 = ConvolutionalReverb(irs, 44100)
    yb = rv2.process(xb)
    assert np.abs(yb[1] - F.conv_reverb(xb[1], h2)).max() <= 5e-6 * np.abs(yb[1]).max()
    # the public response edited IN PLACE must not leave the old one on the device (the device copy is cached: round-4 advice)
    rv2.parameters.dry.value, rv2.parameters.wet.value = 0.0, 1.0
    y_a = rv2.process(x.copy())
    rv2.h *= np.float32(0.5)
    y_b = rv2.process(x.copy())
    assert np.abs(y_b - 0.5 * y_a).max() <= 1e-6 * np.abs(y_a).max()
    rv2.parameters.wet.value = 0.0
    assert np.array_equal(rv2.process(x.copy()), x)
    with pytest.raises(ValueError):
        ConvolutionalReverb(None, 44100)


def _oracle_replay(chain, x_list):
    """Re-apply an AugmentationChain that has just run (probabilities 1) with the oracle, reading the parameter values its
    processors ended up with: the device chain and the numpy oracle must agree effect by effect."""
    from music_mixing_style_transfer_amd.mixing_manipulator import (AugmentationChain, Compressor, ConvolutionalReverb, Equaliser,
                                                                     Gain, MidSideImager, Panner)
    y_list = list(x_list)
    for fx, p, rms in chain.fxs:
        assert p >= 1
        if isinstance(fx, AugmentationChain):
            y_list = _oracle_replay(fx, y_list)
            continue
        out = []
        for x in y_list:
            P = fx.parameters
            if isinstance(fx, Equaliser):
                prm = {b: (getattr(P, b + "_gain").value, getattr(P, b + "_freq").value,
                           getattr(P, b + "_q").value if hasattr(P, b + "_q") else 0.707) for b in fx.bands}
                y = F.equaliser(x, prm, bands=fx.bands)
            elif isinstance(fx, Compressor):
                y = F.compressor(x, P.threshold.value, P.attack_time.value, P.release_time.value, P.ratio.value)
            elif isinstance(fx, Panner):
                y = x * F.panner_gains(P.pan.value, P.pan_law.value)
            elif isinstance(fx, MidSideImager):
                y = F.midside_imager(x, P.bal.value)
            elif isinstance(fx, ConvolutionalReverb):
                y = F.conv_reverb(x, fx.h, P.dry.value, P.wet.value, P.pre_delay.value).astype(np.float32)
            elif isinstance(fx, Gain):
                y = F.gain(x, P.gain.value, P.invert.value)
            else:
                raise AssertionError(type(fx))
            out.append(F.rms_normalize(x, y).astype(np.float32) if rms else np.asarray(y, np.float32))
        y_list = out
    if chain.parallel:
        w = chain.parallel_weight_factor
        y_list = [w * x + (1 - w) * y for x, y in zip(x_list, y_list)]
    return y_list


def test_fx_manipulator_chains_emulated(emu_default, tmp_path):
    """f-3: the instrument FX chains (create_inst_effects_augmentation_chain) - structure like the reference's, and the
    whole drums chain (shuffled eq/comp, pan/imager, low/high parallel convolution reverb, gain) against an oracle replay."""
    import os
    import random
    from music_mixing_style_transfer_amd.data_loader import save_wav_pcm16
    from music_mixing_style_transfer_amd.mixing_manipulator import (AugmentationChain, create_effects_augmentation_chain,
                                                                     create_inst_effects_augmentation_chain, load_impulse_responses)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fx_reverb.npz"))
    for rt, name, h in (("300-600", "roomA", g["h_stereo"][:400]), ("300-600", "roomB", g["h_mono"][:300]),
                        ("3000-6000", "hall", g["h_stereo"][:500]), ("0-300", "ignored", g["h_mono"][:50])):
        d = tmp_path / "IR_set1" / "RT60_avg" / rt / name
        d.mkdir(parents=True)
        save_wav_pcm16(str(d / "impulse_response.wav"), h)
    ir_dir = str(tmp_path / "IR_")
    groups = load_impulse_responses(ir_dir)
    assert [len(gr) for gr in groups] == [2, 1]            # "0-300" is skipped by the reference's glob, >= 3000 ms merged last
    assert groups[0][1]["impulse_response"]().shape == (300, 1) and groups[1][0]["impulse_response"]().shape == (500, 2)
    probs = {"eq": 1, "comp": 1, "pan": 1, "imager": 1, "reverb": 100, "gain": 1}      # reverb * 0.01 = 1 on the low branch
    chain = create_inst_effects_augmentation_chain("drums", probs, ir_dir_path=ir_dir)
    eq_comp, pan_img, rev, gain = chain.fxs
    assert eq_comp[0].shuffle and pan_img[0].shuffle and not eq_comp[2] and gain[0].name == "Gain" and gain[2] is False
    assert [type(f[0]).__name__ for f in eq_comp[0].fxs] == ["Equaliser", "Compressor"] and all(f[2] for f in eq_comp[0].fxs)
    low, high = rev[0].fxs[0][0], rev[0].fxs[1][0]
    assert low.parallel and low.parallel_weight_factor == 0.8 and high.parallel_weight_factor == 0.6
    assert low.fxs[0][0].bands == ["high_shelf"] and high.fxs[0][0].bands == ["low_shelf"] and low.fxs[1][1] == 1.0
    other = create_inst_effects_augmentation_chain("vocals", probs, ir_dir_path=ir_dir)
    assert other.fxs[2][0].parallel and other.fxs[2][0].parallel_weight_factor is None
    algo = create_inst_effects_augmentation_chain("bass", probs)                      # no impulse responses: the algorithmic reverb
    assert type(algo.fxs[2][0].fxs[0][0]).__name__ == "AlgorithmicReverb"
    assert type(create_inst_effects_augmentation_chain("bass", probs, ir_dir_path=ir_dir, algorithmic=True).fxs[2][0].fxs[0][0]).__name__ == \
        "AlgorithmicReverb"
    with pytest.raises(ValueError):
        create_effects_augmentation_chain(["flanger"])
    np.random.seed(3)
    random.seed(3)
    x = g["x"][:1200].copy()
    y = chain([x.copy(), 0.5 * x[::-1].copy()])
    ref = _oracle_replay(chain, [x.copy(), 0.5 * x[::-1].copy()])
    for a, b in zip(y, ref):
        assert a.shape == x.shape and np.isfinite(a).all()
        assert np.abs(a - b).max() <= 2e-5 * max(1e-3, np.abs(b).max())


@pytest.mark.parametrize("L,n_items,C", [(97, 1, 2), (128, 3, 1), (1000, 2, 2), (2049, 33, 2), (4100, 70, 1)])
def test_time_parallel_fx_shapes_emulated(emu_default, L, n_items, C):
    """The time-parallel compressor (chunk maps / chain / fill) and equaliser (chunk scan) against the oracle over ragged
    lengths (below four chunks -> serial form, exact multiples, ragged tails), mono audio and more sequences than one
    wave (n_items * C > 64, not a multiple of 64)."""
    from music_mixing_style_transfer_amd.mixing_manipulator import Compressor, Equaliser
    rng = np.random.default_rng(L + n_items)
    x = (0.2 * rng.standard_normal((n_items, L, C))).astype(np.float32)
    x[0, L // 3:L // 3 + 5] = 0.0                       # |x| < 1e-6 -> -120 dB floor
    c = Compressor(44100)
    for th, at, rt, ra in ((-30.0, 1.5, 60.0, 8.0), (-18.0, 15.0, 400.0, 0.6)):
        c.parameters.threshold.value, c.parameters.attack_time.value = th, at
        c.parameters.release_time.value, c.parameters.ratio.value = rt, ra
        y = c.process(x.copy())
        for i in (0, n_items - 1):
            ref = F.compressor(x[i].copy(), th, at, rt, ra)
            assert np.abs(y[i] - ref).max() <= 3e-7 * max(1.0, np.abs(ref).max()), (L, n_items, C, i)
    for bands in (("low_shelf",), ("first_band", "third_band"), ("low_shelf", "first_band", "second_band", "third_band", "high_shelf")):
        eq = Equaliser(C, 44100, bands=bands)
        prm = {}
        for k, b in enumerate(bands):
            g = float(rng.uniform(-12, 12))
            getattr(eq.parameters, b + "_gain").value = g
            fc = getattr(eq.parameters, b + "_freq").value
            q = getattr(eq.parameters, b + "_q").value if hasattr(eq.parameters, b + "_q") else 0.707
            prm[b] = (g, fc, q)
        y = eq.process(x.copy())
        for i in (0, n_items - 1):
            ref = F.equaliser(x[i].copy(), prm, bands=bands)
            assert np.abs(y[i] - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (L, bands, i)


@pytest.mark.parametrize("L,n_items,C", [(8 * 1024, 1, 2), (9 * 1024 + 517, 3, 2), (13 * 1024 + 31, 2, 1)])
def test_compressor_time_slices_are_bit_identical_emulated(emu_default, L, n_items, C):
    """The compressor's map / chain / apply kernels over THREE time slices (the pipelined form the product uses for large batches, here forced
    on a small one by the per-call hook MstFxFuse.forms = FX_FORM_COMP_SLICE_SMALL) against the single-slice run: the same bits - the
    smoother's value crosses a slice boundary as a float64 - and both against the oracle.  Slices of unequal batch counts, a ragged last
    batch, a short last chunk; the energy sums handed to the chain fusion (per-tile partials reduced in a fixed order) agree with a direct sum."""
    from music_mixing_style_transfer_amd import _lib
    from music_mixing_style_transfer_amd.mixing_manipulator import Compressor
    rng = np.random.default_rng(L)
    x = (0.25 * rng.standard_normal((n_items, L, C))).astype(np.float32)
    c = Compressor(44100)
    c.parameters.threshold.value, c.parameters.attack_time.value = -28.0, 3.0
    c.parameters.release_time.value, c.parameters.ratio.value = 120.0, 6.0
    y1 = c.process(x.copy())
    c.kernel_forms = _lib.FX_FORM_COMP_SLICE_SMALL
    y3 = c.process(x.copy())                                                                  # three slices: unequal batch counts
    assert np.array_equal(y1, y3)
    ref = F.compressor(x[n_items - 1].copy(), -28.0, 3.0, 120.0, 6.0)
    assert np.abs(y3[n_items - 1] - ref).max() <= 3e-7 * max(1.0, np.abs(ref).max())
    # the fused form: sum(y^2) per item (and for stereo the mid / side energies) left behind for the next rms-normalise / imager
    import torch
    from music_mixing_style_transfer_amd.mixing_manipulator.common_audioeffects import _Dev, SUMSQ_SLOTS
    for forms in (0, _lib.FX_FORM_COMP_SLICE_SMALL):
        c.kernel_forms = forms
        d = _Dev(torch.from_numpy(x.copy()))
        yt, sumsq = c._run(d, None, True)
        yy = yt.numpy().astype(np.float64)
        got = sumsq.numpy().reshape(n_items, SUMSQ_SLOTS).sum(1)
        assert np.allclose(got, (yy ** 2).sum((1, 2)), rtol=1e-12), forms
        if C == 2:
            ms = d.last_ms.numpy().reshape(n_items, SUMSQ_SLOTS, 2).sum(1)
            m32, s32 = (yt.numpy()[..., 0] + yt.numpy()[..., 1]), (yt.numpy()[..., 0] - yt.numpy()[..., 1])
            assert np.allclose(ms[:, 0], (m32 * m32).astype(np.float64).sum(1), rtol=1e-12) and np.allclose(ms[:, 1], (s32 * s32).astype(np.float64).sum(1), rtol=1e-12)


@pytest.mark.parametrize("L,n_items", [(1000, 2), (2049, 33), (4384, 3), (65, 1), (70000, 1)])
def test_equaliser_slab_apply_is_bit_identical_emulated(emu_default, L, n_items):
    """The stereo equaliser's apply pass on 16-frame slabs through LDS (in and out as 16-byte pieces; the default)
    against one lane per chunk straight from global memory (MstFxFuse.forms, per call): the same recursion on the same samples from the same start states - the
    same bits, with a short last chunk (guarded samples and pieces), whole chunks only, more chunk pairs than a workgroup and fewer than a
    wave; one, two and five bands."""
    from music_mixing_style_transfer_amd.mixing_manipulator import Equaliser
    rng = np.random.default_rng(L)
    x = (0.2 * rng.standard_normal((n_items, L, 2))).astype(np.float32)
    from music_mixing_style_transfer_amd import _lib
    for bands in (("low_shelf",), ("first_band", "third_band"), ("low_shelf", "first_band", "second_band", "third_band", "high_shelf")):
        eq = Equaliser(2, 44100, bands=bands)
        for b in bands:
            getattr(eq.parameters, b + "_gain").value = float(rng.uniform(-12, 12))
        eq.kernel_forms = _lib.FX_FORM_EQ_LANE_APPLY
        ref = eq.process(x.copy())
        eq.kernel_forms = 0
        got = eq.process(x.copy())
        assert np.array_equal(got, ref), (L, n_items, bands, float(np.abs(got - ref).max()))
        # the state pass on the float64 matrix cores (default) against the VALU dot products with the table in LDS: the same products in
        # the same order - the same chunk start states, the same output bits
        eq.kernel_forms = _lib.FX_FORM_EQ_VALU_ENDS
        ref2 = eq.process(x.copy())
        assert np.array_equal(got, ref2), (L, n_items, bands, float(np.abs(got - ref2).max()))


def test_encoder_rows_kernel_matches_im2col_emulated(emu_default):
    """bf16 FXencoder: the LDS-resident-rows convolution kernel (long early layers) against the im2col kernel it replaces -
    same operands in the same k order, so bit-identical - over strides 1 / 2 / 4, even kernels, ragged last tiles."""
    from music_mixing_style_transfer_amd.networks import FXencoder
    cfg = {"channels": [16, 32, 32, 64], "kernels": [25, 10, 15, 5], "strides": [4, 2, 1, 2], "dilation": [1, 1, 1, 1],
           "bias": True, "norm": "batch", "conv_block": "res", "activation": "relu"}
    sd = synth.fxencoder_state_dict(cfg, seed=11)
    enc = FXencoder({k: (list(v) if isinstance(v, list) else v) for k, v in cfg.items()})
    enc.load_state_dict(sd)
    enc.precision = "bf16"
    for shape in ((2, 2, 5000), (1, 2, 4097)):
        x = synth.synth_audio(shape, seed=shape[2])
        run = enc._get_runner()
        run._ensure(emu_default)
        emu_default.check(emu_default.mst_enc_set_tuning(run.handle, -1), "tuning")     # im2col form only
        ref = enc(x).clone()
        emu_default.check(emu_default.mst_enc_set_tuning(run.handle, 0), "tuning")      # rows form wherever it qualifies
        got = enc(x)
        assert torch.equal(got, ref), shape
        emb = R.fxencoder_forward(sd, cfg, x)
        assert float((got - emb).abs().max()) <= 3e-2 * float(emb.abs().max())
        enc.precision = "bf16x3"          # split mode: both forms again bit for bit, and fp32-class accuracy
        emu_default.check(emu_default.mst_enc_set_tuning(run.handle, -1), "tuning")
        ref3 = enc(x).clone()
        emu_default.check(emu_default.mst_enc_set_tuning(run.handle, 0), "tuning")
        got3 = enc(x)
        assert torch.equal(got3, ref3), shape
        assert float((got3 - emb).abs().max()) <= 2e-5 * float(emb.abs().max())
        enc.precision = "bf16"


def test_encoder_wave_tilings_and_workgroup_orders_match_emulated(emu_default):
    """bf16 / split-bf16 FXencoder, 128-channel layers: the conv kernel with its waves 2 x 2 (two MFMAs per LDS read) and the weight-major
    workgroup order (mst_enc_set_schedule) against the 4 x 1 / column-major forms - same operands, same k order per accumulator: the same bits;
    split-K slices included (short wide layers)."""
    from music_mixing_style_transfer_amd.networks import FXencoder
    cfg = {"channels": [16, 32, 128, 256], "kernels": [25, 10, 5, 5], "strides": [4, 2, 2, 1], "dilation": [1, 1, 1, 1],
           "bias": True, "norm": "batch", "conv_block": "res", "activation": "relu"}
    sd = synth.fxencoder_state_dict(cfg, seed=12)
    enc = FXencoder({k: (list(v) if isinstance(v, list) else v) for k, v in cfg.items()})
    enc.load_state_dict(sd)
    x = synth.synth_audio((3, 2, 3000), seed=77)
    emb = R.fxencoder_forward(sd, cfg, x)
    for precision, tol in (("bf16", 3e-2), ("bf16x3", 2e-5)):
        enc.precision = precision
        run = enc._get_runner()
        run._ensure(emu_default)
        outs = []
        for flags in (0, 1, 2, 3):
            emu_default.check(emu_default.mst_enc_set_schedule(run.handle, flags), "schedule")
            outs.append(enc(x).clone())
        emu_default.check(emu_default.mst_enc_set_schedule(run.handle, 1), "schedule")          # the default
        assert all(torch.equal(o, outs[0]) for o in outs[1:]), precision
        assert float((outs[0] - emb).abs().max()) <= tol * float(emb.abs().max())
    # exact-fp32 mode: the buffer-load gathers against the 64-bit-address fallback (schedule bit 2) - the same bits
    enc.precision = "fp32"
    ref32 = enc(x).clone()
    emu_default.check(emu_default.mst_enc_set_schedule(run.handle, 1 | 4), "schedule")
    assert torch.equal(enc(x), ref32)
    emu_default.check(emu_default.mst_enc_set_schedule(run.handle, 1), "schedule")
    assert float((ref32 - emb).abs().max()) <= 2e-5 * float(emb.abs().max())
    with pytest.raises(ValueError):
        emu_default.check(emu_default.mst_enc_set_schedule(run.handle, 64), "schedule")


def test_encoder_raw_rows_conv_kernel_emulated(emu_default):
    """The 128-channel layers (blocks 4 ... 11 of the default encoder) on enc_conv_taps_kernel - 256-column tiles, the input rows of a 64-channel block
    staged once by LDS-DMA and every tap read from them, loader + matrix waves, split-K over channel blocks - against the four-wave im2col kernel
    (mst_enc_set_schedule bit 5): same bf16 operands, another fp32 summation order (block-major instead of tap-major chunks, other k-slices):
    agreement to accumulation rounding, and the oracle at the bf16 tolerance.  All four instantiations (k = 5 / 10, stride 1 / 2), tiles that end
    inside the batch (columns beyond N), mirrored rows at both ends of every item, one and several channel blocks, items of 32 columns (the shortest the kernel takes: eight items per tile), layers that keep
    the old kernel (output lengths that are not multiples of 32)."""
    from music_mixing_style_transfer_amd.networks import FXencoder
    for cfg, shapes in (({"channels": [16, 64, 128, 256], "kernels": [25, 10, 10, 5], "strides": [4, 2, 2, 1]}, ((3, 2, 4096), (1, 2, 2048))),
                        ({"channels": [16, 64, 128, 128, 256], "kernels": [25, 10, 10, 5, 5], "strides": [4, 2, 1, 2, 1]}, ((2, 2, 4096), (5, 2, 1024), (1, 2, 3000), (3, 2, 512))),
                        # channel counts that are no multiples of 128 (a last channel tile of 64 valid rows) and three channel blocks (uneven k-slices)
                        ({"channels": [16, 64, 192, 320], "kernels": [25, 10, 5, 10], "strides": [4, 2, 2, 1]}, ((2, 2, 2048), (3, 2, 1024)))):
        cfg = dict(cfg, dilation=[1] * len(cfg["kernels"]), bias=True, norm="batch", conv_block="res", activation="relu")
        sd = synth.fxencoder_state_dict(cfg, seed=31)
        enc = FXencoder({k: (list(v) if isinstance(v, list) else v) for k, v in cfg.items()})
        enc.load_state_dict(sd)
        enc.precision = "bf16"
        nb = len(cfg["kernels"])
        for shape in shapes:
            x = synth.synth_audio(shape, seed=shape[2] + shape[0])
            col = []
            R.fxencoder_blocks(x, sd, cfg, collect=col)
            run = enc._get_runner()
            run._ensure(emu_default)
            for n in range(3, nb + 1):
                emu_default.check(emu_default.mst_enc_set_schedule(run.handle, 1 | 32), "schedule")      # four-wave im2col kernel
                ref = enc.forward_blocks(x, n).clone()
                emu_default.check(emu_default.mst_enc_set_schedule(run.handle, 1), "schedule")           # raw rows + loader waves (default)
                got = enc.forward_blocks(x, n)
                scale = float(col[n - 1].abs().max())
                assert got.shape == col[n - 1].shape
                assert float((got - ref).abs().max()) <= 1.6e-2 * scale, (shape, n, float((got - ref).abs().max()), scale)
                assert float((got - ref).abs().mean()) <= 3e-4 * scale, (shape, n)
                assert float((got - col[n - 1]).abs().max()) <= 3e-2 * scale, (shape, n)
            emb = R.fxencoder_forward(sd, cfg, x)
            assert float((enc(x) - emb).abs().max()) <= 3e-2 * float(emb.abs().max())


def test_encoder_fused_stereo_block_emulated(emu_default):
    """The default encoder's stereo block (2 -> 2, k = 25, skip; 2 -> 16, k = 25, stride 4) as ONE kernel (intermediate in LDS, weights through the
    scalar cache, two channels per packed multiply-add) against the two direct-kernel launches it replaces (mst_enc_set_schedule bit 3): same
    operands in the same order - the same bits, in bf16 and in split mode (low plane) - over one-tile and many-tile lengths, lengths that are not
    multiples of the stride, a last tile of one output, and the shortest legal input (reflection padding 12: 13 samples); and the oracle."""
    from music_mixing_style_transfer_amd.networks import FXencoder
    cfg = {"channels": [16], "kernels": [25], "strides": [4], "dilation": [1],
           "bias": True, "norm": "batch", "conv_block": "res", "activation": "relu"}
    sd = synth.fxencoder_state_dict(cfg, seed=21)
    enc = FXencoder({k: (list(v) if isinstance(v, list) else v) for k, v in cfg.items()})
    enc.load_state_dict(sd)
    for shape in ((2, 2, 2037), (1, 2, 4001), (3, 2, 1000), (1, 2, 13), (2, 2, 14), (1, 2, 41), (1, 2, 999)):
        x = synth.synth_audio(shape, seed=shape[2])
        col = []
        R.fxencoder_blocks(x, sd, cfg, collect=col)
        for precision, tol in (("bf16", 2e-2), ("bf16x3", 5e-5)):
            enc.precision = precision
            run = enc._get_runner()
            run._ensure(emu_default)
            emu_default.check(emu_default.mst_enc_set_schedule(run.handle, 1 | 8), "schedule")       # two launches, intermediate in HBM
            ref = enc.forward_blocks(x, 1).clone()
            emu_default.check(emu_default.mst_enc_set_schedule(run.handle, 1), "schedule")           # the fused kernel (default)
            got = enc.forward_blocks(x, 1)
            assert got.shape == col[0].shape
            assert torch.equal(got, ref), (shape, precision)
            assert float((got - col[0]).abs().max()) <= tol * float(col[0].abs().max()), (shape, precision)


def test_encoder_fused_block1_emulated(emu_default):
    """Blocks 1 and 2 of the default encoder (16 -> 16, k = 25, skip; 16 -> 32, k = 25, stride 4 / 32 -> 32, k = 15, skip; 32 -> 64, k = 15, stride 2) in
    bf16 mode as ONE kernel each (input rows by LDS-DMA, the intermediate in LDS, weights resident as A fragments of v_mfma_f32_16x16x32_bf16;
    the second conv of block 2 in two passes of two row tiles) against their two conv launches each (mst_enc_set_schedule bit 4):
    same bf16 operands, fp32 accumulation in k-steps of 32 instead of 16 - agreement to accumulation rounding (a rounding flip of the bf16
    intermediate / output moves an element by one bf16 ulp), and the oracle at the bf16 tolerance.  One-tile and many-tile lengths, lengths
    that are not multiples of the strides, border tiles at both ends, the shortest input the reflection padding allows."""
    from music_mixing_style_transfer_amd.networks import FXencoder
    cfg = {"channels": [16, 32, 64], "kernels": [25, 25, 15], "strides": [4, 4, 2], "dilation": [1, 1, 1],
           "bias": True, "norm": "batch", "conv_block": "res", "activation": "relu"}          # blocks 1 and 2 are the default encoder's (16 / 32 channels, k = 25 / 15)
    sd = synth.fxencoder_state_dict(cfg, seed=22)
    enc = FXencoder({k: (list(v) if isinstance(v, list) else v) for k, v in cfg.items()})
    enc.load_state_dict(sd)
    enc.precision = "bf16"
    for shape in ((2, 2, 8150), (1, 2, 16003), (3, 2, 4000), (1, 2, 200), (2, 2, 3997), (1, 2, 33000)):
        x = synth.synth_audio(shape, seed=shape[2])
        col = []
        R.fxencoder_blocks(x, sd, cfg, collect=col)
        run = enc._get_runner()
        run._ensure(emu_default)
        for nb in (2, 3):
            emu_default.check(emu_default.mst_enc_set_schedule(run.handle, 1 | 16), "schedule")      # two launches per block
            ref = enc.forward_blocks(x, nb).clone()
            emu_default.check(emu_default.mst_enc_set_schedule(run.handle, 1), "schedule")           # the fused kernels (default)
            got = enc.forward_blocks(x, nb)
            scale = float(col[nb - 1].abs().max())
            assert got.shape == col[nb - 1].shape
            assert float((got - ref).abs().max()) <= 1.6e-2 * scale, (shape, nb, float((got - ref).abs().max()), scale)
            assert float((got - ref).abs().mean()) <= 2e-4 * scale, (shape, nb)                      # isolated rounding flips, not a shifted result
            assert float((got - col[nb - 1]).abs().max()) <= 2e-2 * scale, (shape, nb)


def test_algorithmic_reverb_emulated(emu_default):
    """f-3 / AlgorithmicReverb (comb bank as block-wise linear scans, all-pass sections as D independent recurrences) against the
    oracle's sample-by-sample restatement: stereo, mono, a batch; block boundaries (several comb periods), extreme parameters."""
    from music_mixing_style_transfer_amd.mixing_manipulator import AlgorithmicReverb, create_effects_augmentation_chain
    L = 5000
    x = synth.synth_music(2, L, seed=8).numpy().T.copy()
    rv = AlgorithmicReverb()
    for prm in ({}, {"room_size": 0.85, "damping": 0.0, "wet_mix": 1.0, "dry_mix": 0.0, "width": 0.0},
                {"room_size": 0.05, "damping": 1.0, "width": 1.0}):
        for k, v in prm.items():
            getattr(rv.parameters, k).value = v
        kw = {k: getattr(rv.parameters, k).value for k in ("room_size", "damping", "dry_mix", "wet_mix", "width")}
        y = rv.process(x)
        ref = F.algorithmic_reverb(x, **kw)
        assert y.shape == (L, 2) and y.dtype == np.float32
        assert np.abs(y - ref).max() <= 2e-6 * max(1e-3, np.abs(ref).max()), prm
    mono = x[:, :1].copy()
    assert np.abs(rv.process(mono) - F.algorithmic_reverb(mono, **kw)).max() <= 2e-6
    xb = np.stack([x, 0.5 * x[::-1].copy()])
    yb = rv.process(xb)
    assert np.array_equal(yb[0], rv.process(x)) and np.abs(yb[1] - F.algorithmic_reverb(xb[1], **kw)).max() <= 2e-6
    chain = create_effects_augmentation_chain(["reverb"])            # no impulse-response directory: the algorithmic reverb
    assert type(chain.fxs[0][0]).__name__ == "AlgorithmicReverb" and chain.fxs[0][2] is True


def test_advice_round1_regressions_emulated(emu_default):
    """Round-1 advisor findings: (1) integer parameters draw with an exclusive upper bound - ConvolutionalReverb.randomize() never
    indexes past its impulse-response list; (2) a MONO stem through a chain with a panner (mono -> stereo) and rms-normalise."""
    from music_mixing_style_transfer_amd.mixing_manipulator import ConvolutionalReverb, create_effects_augmentation_chain
    h = (np.exp(-np.arange(200) / 40.0)[:, None] * np.ones((1, 2))).astype(np.float32)
    rv = ConvolutionalReverb([[{"impulse_response": (lambda: h)}], [{"impulse_response": (lambda: 0.5 * h)}]], 44100)
    for _ in range(300):
        rv.randomize()                                   # raised IndexError in a third of the calls before
        assert 0 <= rv.parameters.index.value < 2
    mono = synth.synth_music(1, 3000, seed=2).numpy().T.copy()          # [L, 1]
    chain = create_effects_augmentation_chain(["pan", "gain"])
    y = chain([mono])[0]
    assert y.shape == (3000, 2) and np.isfinite(y).all()
    # the panned stereo signal carries the mono input's mean square (rms-normalise across the channel change), up to the gain
    g = 10.0 ** (chain.fxs[1][0].parameters.gain.value / 20.0)
    assert abs(np.mean(y ** 2) / (g * g) - np.mean(mono ** 2)) <= 1e-5 * np.mean(mono ** 2) + 1e-9
