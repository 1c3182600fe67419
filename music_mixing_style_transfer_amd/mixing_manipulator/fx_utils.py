"""Loudness utilities of the input normaliser (reference mixing_manipulator/fx_utils.py:220-238 `lufs_normalize`).

The reference measures with pyloudnorm==0.1.0 (requirements.txt:9), which is not vendored and not installable here.  The
meter is restated from its published algorithm - ITU-R BS.1770-4 integrated loudness as pyloudnorm implements it:
K-weighting = a high-shelf (G = 4 dB, Q = 1/sqrt 2, fc = 1500 Hz) and a high-pass (Q = 0.5, fc = 38 Hz) RBJ biquad applied with
scipy.signal.lfilter, 400 ms gating blocks with 75 % overlap, absolute gate -70 LUFS, relative gate -10 LU (parity unpinned,
see DESIGN.md).  The two filters and the block energies run on the MI355X (mst_fx_biquad_cascade, mst_fx_range_reduce);
the gating over the few thousand block loudness values is host arithmetic.
"""
import math

import numpy as np

from . import _device_ops as D


def kweighting_coefficients(rate):
    """((b, a) high_shelf, (b, a) high_pass), a0-normalised float64, pyloudnorm IIRfilter.generate_coefficients."""
    out = []
    for G, Q, fc, kind in ((4.0, 1.0 / math.sqrt(2.0), 1500.0, "high_shelf"), (0.0, 0.5, 38.0, "high_pass")):
        A = 10.0 ** (G / 40.0)
        w0 = 2.0 * math.pi * (fc / rate)
        alpha = math.sin(w0) / (2.0 * Q)
        cw = math.cos(w0)
        if kind == "high_shelf":
            b0 = A * ((A + 1) + (A - 1) * cw + 2 * math.sqrt(A) * alpha)
            b1 = -2 * A * ((A - 1) + (A + 1) * cw)
            b2 = A * ((A + 1) + (A - 1) * cw - 2 * math.sqrt(A) * alpha)
            a0 = (A + 1) - (A - 1) * cw + 2 * math.sqrt(A) * alpha
            a1 = 2 * ((A - 1) - (A + 1) * cw)
            a2 = (A + 1) - (A - 1) * cw - 2 * math.sqrt(A) * alpha
        else:
            b0 = (1 + cw) / 2
            b1 = -(1 + cw)
            b2 = (1 + cw) / 2
            a0 = 1 + alpha
            a1 = -2 * cw
            a2 = 1 - alpha
        out.append((np.array([b0, b1, b2]) / a0, np.array([a0, a1, a2]) / a0))
    return out


def gated_loudness(z, G=(1.0, 1.0, 1.0, 1.41, 1.41)):
    """Block mean squares z [channels, blocks] -> integrated loudness (BS.1770-4 gating, pyloudnorm Meter.integrated_loudness).
    Array form of pyloudnorm's per-block Python loops: the same sums in the same order (channels added one after the other, block
    means by numpy's mean over the selected blocks), a few thousand blocks in microseconds instead of 7 ms per call."""
    z = np.asarray(z, dtype=np.float64)
    n_ch = z.shape[0]
    g = np.asarray(G[:n_ch], dtype=np.float64)[:, None]
    with np.errstate(divide="ignore", invalid="ignore"), _quiet():
        lj = -0.691 + 10.0 * np.log10(np.add.reduce(g * z, axis=0))
        gamma_a = -70.0
        Jg = lj >= gamma_a
        z_avg = np.array([np.mean(z[i, Jg]) for i in range(n_ch)])
        gamma_r = -0.691 + 10.0 * np.log10(np.add.reduce(g[:, 0] * z_avg)) - 10.0
        Jg = (lj > gamma_r) & (lj > gamma_a)
        z_avg = np.nan_to_num(np.array([np.mean(z[i, Jg]) for i in range(n_ch)]))
        return float(-0.691 + 10.0 * np.log10(np.add.reduce(g[:, 0] * z_avg)))


class _quiet:
    def __enter__(self):
        import warnings
        self._w = warnings.catch_warnings()
        self._w.__enter__()
        warnings.simplefilter("ignore")

    def __exit__(self, *a):
        return self._w.__exit__(*a)


_block_index = {}


class Meter:
    """BS.1770 loudness meter with pyloudnorm.Meter's interface (rate, block_size=0.400)."""

    def __init__(self, rate, filter_class="K-weighting", block_size=0.400):
        if filter_class != "K-weighting":
            raise ValueError("only the K-weighting filter class is provided")
        self.rate, self.block_size = rate, block_size
        self._filters = kweighting_coefficients(rate)

    def block_bounds(self, n_samples):
        T_g, step = self.block_size, 0.25
        T = n_samples / self.rate
        n_blocks = int(np.round(((T - T_g) / (T_g * step))) + 1)
        lo = [int(T_g * (j * step) * self.rate) for j in range(n_blocks)]
        hi = [int(T_g * (j * step + 1) * self.rate) for j in range(n_blocks)]
        return lo, hi

    def integrated_loudness(self, data):
        """data: numpy / device [L] or [L, C] float; returns LUFS (float)."""
        x = D.to_device(data)
        L, Cn = x.shape
        if Cn > 5:
            raise ValueError("Audio must have five channels or less.")
        if L < self.block_size * self.rate:
            raise ValueError("Audio must have length greater than the block size.")
        for b, a in self._filters:                       # each stage rounds to float32 like the reference's in-place filtering
            x = D.biquad(x, b, a)
        key = (L, self.rate, self.block_size)
        if key not in _block_index:          # the gating blocks of a signal length: built and uploaded once (four measurements per stem)
            if len(_block_index) > 8:
                _block_index.clear()
            lo, hi = self.block_bounds(L)
            _block_index[key] = (lo, hi, D.RangeIndex([0] * len(lo), lo, hi))
        lo, hi, index = _block_index[key]
        z = np.zeros((Cn, len(lo)))
        for c in range(Cn):
            z[c] = D.range_reduce(x, index, lo, hi, channel=c, mode="sumsq") / (self.block_size * self.rate)
        return gated_loudness(z)


def lufs_normalize(x, sr, lufs, log=True):
    """reference fx_utils.py:220-238: measure x + 1e-10, apply the gain to x, then divide by max(1, 1e-6 + peak).
    x: numpy [L] / [L, C] float32 -> numpy float32 of the same shape; a device tensor in gives a device tensor out."""
    is_np = isinstance(x, np.ndarray)
    xd = D.to_device(x)
    loudness = Meter(sr).integrated_loudness(xd + np.float32(1e-10))
    if log:
        print("original loudness: ", loudness, " max value: ", float(xd.abs().max()))
    gain = np.power(10.0, (lufs - loudness) / 20.0)             # pyloudnorm.normalize.loudness
    gain = float(np.asarray(gain).reshape(-1)[0])                # the reference's features file holds the targets as shape-(1,) arrays
    y = xd * np.float32(gain)                                    # float32 array times a scalar stays float32 (NumPy 1.x promotion)
    peak = float(D.range_reduce(y.reshape(1, -1, 1), [0], [0], [y.numel()], 0, "max")[0])
    y = y / np.float32(np.maximum(1.0, 1e-6 + peak))
    if not is_np:
        return y if x.dim() == 2 else y[:, 0]
    out = y.cpu().numpy()
    return out if x.ndim == 2 else out[:, 0]
