"""The library's own power-of-two real FFTs (csrc/fft_kernels.h: Stockham radix-2 passes + the real-sequence un-mix; they replaced hipFFT in
round 4) on the SIMT emulator against numpy.fft - through the two entry points that use them: the STFT mean magnitude (R2C) and the FFT
convolution (R2C, spectrum product, C2R).  Lengths cover the global Stockham passes (up to 256 points: odd and even pass counts - the inverse
starts in a different buffer - and the smallest transform) and the four-step form with in-LDS sub-transforms (from 512 points on: square and
2:1 splits, up to 2^17 points)."""
import numpy as np
import pytest
import torch


@pytest.mark.parametrize("n_fft", [4, 8, 64, 256, 512, 1024, 4096, 32768])
def test_stft_mean_magnitude_vs_numpy(emu_default, n_fft):
    from music_mixing_style_transfer_amd.mixing_manipulator import _device_ops as D
    rng = np.random.default_rng(n_fft)
    hop = max(1, n_fft // 4)
    L = n_fft + 6 * hop + 3
    x = rng.standard_normal((L, 2)).astype(np.float32)
    win = np.sqrt(np.hanning(n_fft + 1)[:-1]).astype(np.float32) if n_fft > 4 else np.ones(n_fft, np.float32)
    st = D.StftMeanMagnitude(n_fft, hop, win, max_batch=4)          # 7 frames in batches of 4: a full and a ragged batch
    for ch in (0, 1):
        got = st(torch.from_numpy(x), ch)
        n_frames = 1 + (L - n_fft) // hop
        frames = np.stack([x[f * hop:f * hop + n_fft, ch] * win for f in range(n_frames)]).astype(np.float32)
        want = np.abs(np.fft.rfft(frames.astype(np.float64), axis=1)).mean(0)
        assert got.shape == (n_fft // 2 + 1,)
        assert np.abs(got - want).max() <= 3e-6 * max(1.0, want.max()), (n_fft, np.abs(got - want).max())


@pytest.mark.parametrize("L,nt", [(5, 3), (40, 9), (100, 31), (200, 57), (400, 101), (700, 101), (3000, 257), (9000, 1001), (40000, 1001), (100000, 2001)])
def test_fft_convolution_vs_numpy(emu_default, L, nt):
    """fir_causal = one FFT convolution (transform lengths 8 ... 131072 here)."""
    from music_mixing_style_transfer_amd.mixing_manipulator import _device_ops as D
    rng = np.random.default_rng(L)
    x = rng.standard_normal((L, 1)).astype(np.float32)
    taps = (rng.standard_normal(nt) / nt).astype(np.float64)
    y = D.fir_causal(torch.from_numpy(x), taps).numpy()[:, 0]
    xe = np.concatenate([np.full(nt - 1, x[0, 0], np.float64), x[:, 0].astype(np.float64)])
    want = np.convolve(xe, taps.astype(np.float32).astype(np.float64))[nt - 1:nt - 1 + L]
    assert np.abs(y - want).max() <= 5e-6 * max(1.0, np.abs(want).max()), (L, nt, np.abs(y - want).max())


def test_fft_lengths_must_be_powers_of_two(emu_default):
    from music_mixing_style_transfer_amd.mixing_manipulator import _device_ops as D
    with pytest.raises(NotImplementedError):
        D.StftMeanMagnitude(1000, 250, np.ones(1000, np.float32), max_batch=2)


@pytest.mark.gpu
@pytest.mark.parametrize("L,nt", [(300000, 100001), (50000, 600001), (5000000, 600001)])
def test_long_transforms_and_long_responses_on_gpu(L, nt):
    """The largest four-step kernels (2^18 ... 2^21 real points: LOGMAX 9 and 10 column / row kernels, no emulator test reaches them) and the
    block choice for responses longer than 2^19 samples: (300000, 100001) is one 2^19-point transform, (50000, 600001) one 2^21-point
    transform, (5000000, 600001) overlap-save with blocks of TWICE the response (four times would need 2^22 points: round 4 returned
    MST_ERR_UNSUPPORTED there).  Against scipy's float64 FFT convolution; a response beyond 2^20 samples is refused with a message."""
    import scipy.signal
    from music_mixing_style_transfer_amd import _lib
    from music_mixing_style_transfer_amd.mixing_manipulator import _device_ops as D
    rng = np.random.default_rng(L + nt)
    x = rng.standard_normal((L, 1)).astype(np.float32)
    taps = (rng.standard_normal(nt) * np.exp(-np.arange(nt) / (0.2 * nt)) / np.sqrt(nt)).astype(np.float32)
    y = D.fir_causal(torch.from_numpy(x).cuda(), taps.astype(np.float64)).cpu().numpy()[:, 0]
    xe = np.concatenate([np.full(nt - 1, x[0, 0], np.float64), x[:, 0].astype(np.float64)])
    want = scipy.signal.fftconvolve(xe, taps.astype(np.float64))[nt - 1:nt - 1 + L]
    err = np.abs(y - want).max() / max(1.0, np.abs(want).max())
    print(f"fir_causal L = {L}, {nt} taps: max deviation {err:.2e} of max|ref|")
    assert err <= 2e-5, (L, nt, err)
    import ctypes as C
    lib = _lib.lib()
    h = C.c_void_p()
    assert lib.mst_fx_convolver_create(10000000, (1 << 20) + 1, 1, 1, C.byref(h)) == -2          # MST_ERR_UNSUPPORTED
    assert b"2^20" in lib.mst_last_error()
