"""WAV reader / writer of the inference data path against vectors recorded from the reference's
loader_utils.load_wav_segment (tests/golden/make_golden.py)."""
import os
import wave

import numpy as np
import pytest

from music_mixing_style_transfer_amd.data_loader import load_wav_length, load_wav_segment, save_wav_pcm16

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _write(path, pcm, width, rate=44100):
    with wave.open(path, "w") as w:
        w.setnchannels(pcm.shape[1])
        w.setsampwidth(width)
        w.setframerate(rate)
        w.writeframes(pcm.tobytes())


@pytest.mark.parametrize("name,width", [("pcm16", 2), ("pcm32", 4)])
def test_reader_matches_reference(tmp_path, name, width):
    g = np.load(os.path.join(GOLD, "wav.npz"))
    path = str(tmp_path / (name + ".wav"))
    _write(path, g[name], width)
    assert load_wav_length(path) == int(g[name + "_len"])
    a0 = load_wav_segment(path, axis=0)
    assert a0.dtype == np.float64 and np.array_equal(a0, g[name + "_axis0"])
    assert np.array_equal(load_wav_segment(path, start_point=10, duration=100, axis=1), g[name + "_axis1_seg"])


def test_reader_errors_like_reference(tmp_path):
    pcm = np.zeros((10, 2), np.int16)
    p = str(tmp_path / "a.wav")
    _write(p, pcm, 2, rate=48000)
    with pytest.raises(ValueError, match="sample rate should be 44100"):
        load_wav_segment(p)
    p8 = str(tmp_path / "b.wav")
    _write(p8, np.zeros((10, 2), np.uint8), 1)
    with pytest.raises(ValueError, match="bit depth should be 16 or 32-bit"):
        load_wav_segment(p8)


def test_pcm16_writer_roundtrip(tmp_path):
    x = np.stack([np.linspace(-1, 1, 1000), np.linspace(1, -1, 1000)], 1)
    p = str(tmp_path / "o.wav")
    save_wav_pcm16(p, x)
    y = load_wav_segment(p, axis=1)
    assert y.shape == x.shape and np.abs(y - x).max() <= 5e-5   # x*32767 on write, /32768 on read
