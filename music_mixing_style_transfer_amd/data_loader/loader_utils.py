"""WAV I/O helpers of the inference data path (reference data_loader/loader_utils.py:40-70).

16/32-bit PCM -> float64 in [-1, 1) exactly like the reference (int16 / 2**15, int32 / 2**31), stereo
de-interleaved to [2, L] (axis=0) or [L, 2] (axis=1).  Same ValueErrors for a wrong sample rate / bit depth.
soundfile is not a dependency: writing uses the standard `wave` module (PCM_16); `SlicedWavWriter` lets every rank of a
multi-GPU run write the time range it produced straight into the output file (no gather of the audio to one rank).
"""
import os
import struct
import wave

import numpy as np


def load_wav_length(audio_path):
    with wave.open(audio_path, "r") as w:
        return w.getnframes()


def read_wav_raw(audio_path):
    """(frame rate, bytes per sample, channels, frames, the raw PCM frames) of a wav file - the part of loading that is pure file I/O, which
    the dataset's prefetch thread does one song ahead (data_loader.py)."""
    with wave.open(audio_path, "r") as w:
        rate, width, nch, n = w.getframerate(), w.getsampwidth(), w.getnchannels(), w.getnframes()
        raw = w.readframes(n)
    return rate, width, nch, n, raw


def load_wav_segment(audio_path, start_point=None, duration=None, axis=1, sample_rate=44100, preread=None):
    start_point = 0 if start_point is None else start_point
    if preread is not None and audio_path in preread:          # the whole file is in memory already
        rate, width, nch, n, raw = preread[audio_path]
        duration = n if duration is None else duration
        if rate != sample_rate:
            raise ValueError(f"ValueError: input audio's sample rate should be {sample_rate}")
        a, b = min(max(0, start_point), n), min(n, max(0, start_point) + duration)
        raw = raw[a * width * nch:b * width * nch]
    else:
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


def load_wav_device(audio_path, device, sample_rate=44100, preread=None):
    """The whole file as a float32 DEVICE tensor [2, L] (load_wav_segment(path, axis=0) followed by the dataset's `.float()`): the raw
    PCM frames are uploaded as they are (2 or 4 bytes per sample instead of a float64 array made on the host) and de-interleaved /
    scaled on the device - int / 2**15 (2**31) in float64, then rounded to float32, exactly the reference's two steps.  Stereo only
    (what the stems are); other channel counts go through load_wav_segment.  preread: {path: read_wav_raw(path)} of files already in memory."""
    import torch
    rate, width, nch, n, raw = preread[audio_path] if (preread is not None and audio_path in preread) else read_wav_raw(audio_path)
    if rate != sample_rate:
        raise ValueError(f"ValueError: input audio's sample rate should be {sample_rate}")
    if width not in (2, 4):
        raise ValueError("ValueError: input audio's bit depth should be 16 or 32-bit")
    if nch != 2:
        raise ValueError("load_wav_device: stereo files only")
    import warnings
    with warnings.catch_warnings():          # the frames are only read (uploaded): no writable copy of the file's 30 MB
        warnings.simplefilter("ignore")
        host = torch.frombuffer(raw, dtype=torch.int16 if width == 2 else torch.int32)
    dev = host.to(device, non_blocking=False).view(-1, 2)
    x = dev.to(torch.float64) / float(2 ** 15 if width == 2 else 2 ** 31)
    return x.to(torch.float32).t().contiguous()


def pcm16_device(x):
    """float device tensor [..., L, C] in [-1, 1] -> int16 tensor, the arithmetic of pcm16(): round-half-even(x * 32767), clipped."""
    import torch
    return torch.clamp(torch.round(x.to(torch.float32) * 32767.0), -32768, 32767).to(torch.int16)


def save_wav_pcm16(audio_path, data, sample_rate=44100):
    """data float [L, C] in [-1, 1] -> 16-bit PCM (what sf.write(..., 'PCM_16') is used for at
    inference/style_transfer.py:174,177): round(x * 32767), clipped."""
    data = np.asarray(data)
    if data.ndim == 1:
        data = data[:, None]
    pcm = data.astype("<i2") if data.dtype == np.int16 else pcm16(data)          # int16 in: already converted (on the device)
    with wave.open(audio_path, "w") as w:
        w.setnchannels(pcm.shape[1])
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(pcm.tobytes())


def pcm16(data):
    """float [-1, 1] -> little-endian int16 exactly like save_wav_pcm16 (elementwise: slices convert independently)."""
    return np.clip(np.rint(np.asarray(data) * 32767.0), -32768, 32767).astype("<i2")


class SlicedWavWriter:
    """A 16-bit PCM wav file of known length written in time slices, possibly by several processes of one node:
    `create()` (one process) writes the 44-byte RIFF header the `wave` module would write and sizes the file;
    `write(t0, data[L_slice, C])` (any process, after create) stores rows [t0, t0 + L_slice).  The finished file is
    byte-identical to save_wav_pcm16 of the whole signal."""

    def __init__(self, path, n_frames, n_channels=2, sample_rate=44100):
        self.path, self.n_frames, self.nch, self.sr = path, int(n_frames), int(n_channels), int(sample_rate)

    def create(self):
        nbytes = self.n_frames * self.nch * 2
        header = b"RIFF" + struct.pack("<L4s4sLHHLLHH4sL", 36 + nbytes, b"WAVE", b"fmt ", 16, 1, self.nch, self.sr,
                                       self.nch * self.sr * 2, self.nch * 2, 16, b"data", nbytes)
        with open(self.path, "wb") as f:
            f.write(header)
            f.truncate(44 + nbytes)

    def write(self, t0, data):
        data = np.asarray(data)
        if data.ndim == 1:
            data = data[:, None]
        if data.shape[1] != self.nch or t0 < 0 or t0 + data.shape[0] > self.n_frames:
            raise ValueError("SlicedWavWriter.write: slice outside the file")
        if data.shape[0] == 0:
            return
        fd = os.open(self.path, os.O_WRONLY)
        try:
            os.pwrite(fd, (data.astype("<i2") if data.dtype == np.int16 else pcm16(data)).tobytes(), 44 + int(t0) * self.nch * 2)
        finally:
            os.close(fd)
