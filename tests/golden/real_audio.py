"""real_audio.npz helpers (test infrastructure), shared by make_golden.py and the tests.

Lossless packing of 16-bit PCM for the fixtures: per-channel first differences, the low and the high bytes as two
planes, xz.  Real music packs to about 40 % of its raw size this way (deflate on the interleaved samples: 65 %)."""
import lzma

import numpy as np


def pack(pcm):
    """int16 [L, C] -> uint8 array (an npz entry)"""
    pcm = np.ascontiguousarray(pcm, dtype="<i2")
    assert pcm.ndim == 2
    planar = pcm.T.astype(np.int32)
    d = np.ascontiguousarray((np.diff(planar, axis=1, prepend=0) & 0xFFFF).astype("<u2"))          # wraps like int16 arithmetic
    b = d.view(np.uint8).reshape(-1, 2)
    raw = b[:, 0].tobytes() + b[:, 1].tobytes()
    head = np.array(pcm.shape, dtype="<i8").tobytes()
    return np.frombuffer(head + lzma.compress(raw, preset=9 | lzma.PRESET_EXTREME), dtype=np.uint8)


def unpack(blob):
    """inverse of pack(): uint8 array -> int16 [L, C]"""
    blob = np.asarray(blob, dtype=np.uint8).tobytes()
    L, C = (int(v) for v in np.frombuffer(blob[:16], dtype="<i8"))
    raw = np.frombuffer(lzma.decompress(blob[16:]), dtype=np.uint8)
    n = L * C
    d = (raw[:n].astype(np.uint16) | (raw[n:].astype(np.uint16) << 8)).reshape(C, L)
    planar = np.cumsum(d.astype(np.uint32), axis=1, dtype=np.uint32).astype(np.uint16).view(np.int16)
    return np.ascontiguousarray(planar.T)


# ---- the two "songs" of real_audio.npz (make_golden.py real_audio) ---------------------------------------------------------------------
STEMS = ("drums", "bass")
SEG = 2 ** 19


def extremes_from(pcm):
    """The derived 'extremes' song, an integer recipe on the committed PCM of the real song (so that it costs no fixture bytes):
    input drums = one whole segment of exact digital silence, then the real drums x 16 saturated to the int16 range (-32768 decodes to exactly
    -1.0, +32767 to the largest sample a 16-bit file holds); input bass = the real bass x 64 saturated (a fifth of its samples on the rails);
    reference drums = exact digital silence (the FXencoder on an all-zero stem), reference bass = the real reference bass x 8 saturated."""
    sat = lambda x, g: np.clip(x.astype(np.int32) * g, -32768, 32767).astype("<i2")
    d = sat(pcm["input/drums"], 16)
    d[:SEG] = 0
    return {"input/drums": d, "input/bass": sat(pcm["input/bass"], 64),
            "reference/drums": np.zeros_like(pcm["reference/drums"]), "reference/bass": sat(pcm["reference/bass"], 8)}


def probe_index(length, n=4096):
    """sample positions of the stored output probes: both ends, both sides of the segment cut, and n evenly spread positions"""
    edges = [np.arange(0, 64), np.arange(length - 64, length), np.arange(SEG - 64, SEG + 64)]
    return np.unique(np.concatenate(edges + [np.linspace(0, length - 1, n).astype(np.int64)]))


def write_wav(path, pcm, sr=44100):
    import wave
    with wave.open(str(path), "w") as w:
        w.setnchannels(pcm.shape[1])
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(np.ascontiguousarray(pcm, dtype="<i2").tobytes())


def stage(root, songs):
    """songs: {song name: {"input/drums": int16 [L, 2], ...}} -> <root>/<song>/separated/{input,reference}/<stem>.wav"""
    import os
    for song, files in songs.items():
        for key, pcm in files.items():
            kind, stem = key.split("/")
            d = os.path.join(str(root), song, "separated", kind)
            os.makedirs(d, exist_ok=True)
            write_wav(os.path.join(d, stem + ".wav"), pcm)
