"""Stereo-image / panning normalisation (reference mixing_manipulator/normalization_imager.py: normalize_imager :22-80,
process_balance :86-99, lr_to_ms :103-106, ms_to_lr :110-113).

normalize_imager re-balances mid/side, then left/right, then mid/side again - every step is a pair of gains derived from
two energies, so the whole procedure is ONE 2x2 re-mix of (L, R) whose coefficients follow from the input's three second
moments (sum L^2, sum R^2, sum L*R).  On the MI355X: one reduction pass (mst_fx_stereo_moments) and one apply pass
(mst_fx_stereo_mix) instead of the reference's ~20 array passes; the moments are float64 sums (the reference sums float32
arrays), a <= 1e-6 relative difference.  The optional Haas widening of near-mono stems goes through the product's
AugmentationChain / Haas processor like the reference's.
"""
import numpy as np
import torch

from .. import _lib
from . import _device_ops as D
from .common_audioeffects import AugmentationChain, Haas


def lr_to_ms(left, right):
    return left + right, left - right


def ms_to_lr(mid, side):
    return (mid + side) / 2, (mid - side) / 2


def balance_gains(e_1, e_2, tgt_e1_bal=0.5, eps=1e-04):
    """The two gains of process_balance from the two energies."""
    total_e = e_1 + e_2
    g1 = np.sqrt(tgt_e1_bal * total_e / (e_1 + eps))
    left_e_1 = total_e - e_1 * (g1 ** 2)
    g2 = np.sqrt(left_e_1 / (e_2 + 1e-3))
    return g1, g2


def process_balance(data_1, data_2, tgt_e1_bal=0.5, eps=1e-04):
    """Balance the energies of two signals: data_1 gets the share tgt_e1_bal of the total (numpy arrays, like the reference)."""
    g1, g2 = balance_gains(np.sum(data_1 ** 2), np.sum(data_2 ** 2), tgt_e1_bal, eps)
    return data_1 * g1, data_2 * g2


def _moments(x):
    lib = _lib.lib()
    out = torch.empty(3, dtype=torch.float64, device=x.device)
    with lib.device_ctx(x):
        lib.check(lib.mst_fx_stereo_moments(x.data_ptr(), 1, x.shape[0], out.data_ptr(), lib.stream_ptr(x)), "mst_fx_stereo_moments")
    ll, rr, lr = out.cpu().numpy()
    return np.array([[ll, lr], [lr, rr]])


def _energy(M, v):
    return float(v @ M @ v)


def imager_matrix(M, target_side_mid_bal, eps=1e-04):
    """The 2x2 matrix normalize_imager applies to (L, R), from the second-moment matrix M of (L, R)."""
    mid, side = np.array([1.0, 1.0]), np.array([1.0, -1.0])                # rows: coefficients on (L, R)
    g1, g2 = balance_gains(_energy(M, mid), _energy(M, side), target_side_mid_bal, eps)
    mid, side = mid * g1, side * g2
    left, right = (mid + side) / 2, (mid - side) / 2
    g1, g2 = balance_gains(_energy(M, left), _energy(M, right), 0.5, eps)
    left, right = left * g1, right * g2
    mid, side = left + right, left - right
    g1, g2 = balance_gains(_energy(M, mid), _energy(M, side), target_side_mid_bal, eps)
    mid, side = mid * g1, side * g2
    return np.stack([(mid + side) / 2, (mid - side) / 2])


def normalize_imager(data, target_side_mid_bal=0.9, mono_threshold=0.95, sr=44100, eps=1e-04, verbose=False, haas=None):
    """data [L, 2] (numpy, or a device tensor that then stays on the device) -> the image-normalised signal [L, 2].
    haas (extension; the reference has no such argument): the chain applied to an almost-mono signal instead of a freshly randomised
    AugmentationChain([Haas]) - what makes that branch reproducible (tests pass a chain with fixed parameters)."""
    is_np = isinstance(data, np.ndarray)
    x = D.to_device(data)
    M = _moments(x)
    mid_e, side_e = _energy(M, np.array([1.0, 1.0])), _energy(M, np.array([1.0, -1.0]))
    if mid_e / (mid_e + side_e) > mono_threshold:               # Haas effect on an almost-mono signal (randomised parameters)
        chain = haas if haas is not None else AugmentationChain(fxs=[(Haas(sample_rate=sr), 1, True)])
        x = chain([x])[0]
        M = _moments(x)
    W = imager_matrix(M, target_side_mid_bal, eps)
    lib = _lib.lib()
    y = torch.empty_like(x)
    with lib.device_ctx(x):
        lib.check(lib.mst_fx_stereo_mix(x.data_ptr(), y.data_ptr(), 1, x.shape[0], float(W[0, 0]), float(W[0, 1]), float(W[1, 0]),
                                        float(W[1, 1]), lib.stream_ptr(x)), "mst_fx_stereo_mix")
    return y.cpu().numpy() if is_np else y
