"""WAV I/O helpers of the inference data path (reference data_loader/loader_utils.py:40-70).

16/32-bit PCM -> float64 in [-1, 1) exactly like the reference (int16 / 2**15, int32 / 2**31), stereo
de-interleaved to [2, L] (axis=0) or [L, 2] (axis=1).  Same ValueErrors for a wrong sample rate / bit depth.
soundfile is not a dependency: writing uses the standard `wave` module (PCM_16).
"""
import wave

import numpy as np


def load_wav_length(audio_path):
    with wave.open(audio_path, "r") as w:
        return w.getnframes()


def load_wav_segment(audio_path, start_point=None, duration=None, axis=1, sample_rate=44100):
    start_point = 0 if start_point is None else start_point
    with wave.open(audio_path, "r") as w:
        duration = w.getnframes() if duration is None else duration
        if w.getframerate() != sample_rate:
            raise ValueError(f"ValueError: input audio's sample rate should be {sample_rate}")
        w.setpos(start_point)
        raw = w.readframes(duration)
        width, nch = w.getsampwidth(), w.getnchannels()
    if width == 2:
        X = np.frombuffer(raw, dtype=np.int16) / float(2 ** 15)
    elif width == 4:
        X = np.frombuffer(raw, dtype=np.int32) / float(2 ** 31)
    else:
        raise ValueError("ValueError: input audio's bit depth should be 16 or 32-bit")
    if nch == 2:
        X = np.concatenate((np.expand_dims(X[::2], axis=axis), np.expand_dims(X[1::2], axis=axis)), axis=axis)
    return X


def save_wav_pcm16(audio_path, data, sample_rate=44100):
    """data float [L, C] in [-1, 1] -> 16-bit PCM (what sf.write(..., 'PCM_16') is used for at
    inference/style_transfer.py:174,177): round(x * 32767), clipped."""
    data = np.asarray(data)
    if data.ndim == 1:
        data = data[:, None]
    pcm = np.clip(np.rint(data * 32767.0), -32768, 32767).astype("<i2")
    with wave.open(audio_path, "w") as w:
        w.setnchannels(pcm.shape[1])
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(pcm.tobytes())
