"""Device calls under the input normaliser: thin typed wrappers over the C ABI (include/mst_hip.h) working on torch device
tensors.  Host-side control flow lives in fx_utils.py / utils_data_normalization.py / normalization_imager.py."""
import ctypes as C

import numpy as np
import torch

from .. import _lib


def to_device(x):
    """numpy / torch [L] or [L, C] float -> contiguous float32 device tensor [L, C]."""
    b = _lib.lib()
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if isinstance(x, np.ndarray) else x.to(torch.float32)
    if t.dim() == 1:
        t = t[:, None]
    return b.to_device(t).contiguous()


def _i64(values, device):
    return torch.from_numpy(np.asarray(values, dtype=np.int64)).to(device)


class RangeIndex:
    """(item, lo, hi) arrays of a set of ranges, uploaded once per device: pass it as `items` of range_reduce (with the same lo / hi lists)
    when the same ranges are reduced again and again - the BS.1770 gating blocks of a stem are measured four times per normalisation."""

    def __init__(self, items, lo, hi):
        self.items, self.lo, self.hi = np.asarray(items, dtype=np.int32), np.asarray(lo, dtype=np.int64), np.asarray(hi, dtype=np.int64)
        self._dev = {}

    def on(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(torch.from_numpy(a).to(device) for a in (self.items, self.lo, self.hi))
        return self._dev[key]


def biquad(x, b, a):
    """One second-order section over the whole signal from zero state (scipy.signal.lfilter recursion, float64 inside,
    float32 result): x device [L, C]."""
    lib = _lib.lib()
    coef = np.ascontiguousarray([[b[0], b[1], b[2], a[0], a[1], a[2]]], dtype=np.float64)
    L, Cn = x.shape
    y = torch.empty_like(x)
    with lib.device_ctx(x):
        nbytes = lib.mst_fx_biquad_scratch_bytes(1, L, Cn, 1)
        sc = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=x.device)
        lib.check(lib.mst_fx_biquad_cascade(x.data_ptr(), y.data_ptr(), 1, L, Cn, coef.ctypes.data_as(C.POINTER(C.c_double)), 1,
                                            sc.data_ptr(), nbytes, None, lib.stream_ptr(x)), "mst_fx_biquad_cascade")
    return y


_MAX_SPLIT = 1 << 15      # samples per workgroup of a max-reduction (fx_range_reduce_kernel runs ONE workgroup per range)


def range_reduce(x, items, lo, hi, channel=0, mode="sumsq"):
    """x device [n, L, C] (or [L, C]); float64 numpy [n_ranges]: sum of squares / max |x| over x[items[r], lo[r]:hi[r], channel].
    A max over a long range (the peak of a whole stem: 16 M samples on one workgroup took 17 ms) is cut into pieces of 2^15 samples
    that run as separate ranges and are combined on the host - a maximum does not depend on the order, the result is the same bits."""
    lib = _lib.lib()
    if x.dim() == 2:
        x = x[None]
    n, L, Cn = x.shape
    r = len(lo)
    if r == 0:
        return np.zeros(0)
    owner = None
    if mode != "sumsq" and not isinstance(items, RangeIndex) and any(min(int(b), L) - max(int(a), 0) > _MAX_SPLIT for a, b in zip(lo, hi)):
        items2, lo2, hi2, owner = [], [], [], []
        for k, (it, a, b) in enumerate(zip(items, lo, hi)):
            a, b = max(int(a), 0), min(int(b), L)
            starts = range(a, b, _MAX_SPLIT) if b > a else [a]
            for s0 in starts:
                items2.append(it)
                lo2.append(s0)
                hi2.append(min(b, s0 + _MAX_SPLIT))
                owner.append(k)
        items, lo, hi = items2, lo2, hi2
    dev = x.device
    if isinstance(items, RangeIndex):          # index arrays that already live on the device (the loudness meter's gating blocks)
        it, lo_t, hi_t = items.on(dev)
    else:
        it = torch.from_numpy(np.asarray(items, dtype=np.int32)).to(dev)
        lo_t, hi_t = _i64(lo, dev), _i64(hi, dev)
    out = torch.empty(len(lo), dtype=torch.float64, device=dev)
    with lib.device_ctx(x):
        lib.check(lib.mst_fx_range_reduce(x.data_ptr(), L, Cn, channel, it.data_ptr(), lo_t.data_ptr(), hi_t.data_ptr(), len(lo),
                                          0 if mode == "sumsq" else 1, out.data_ptr(), lib.stream_ptr(x)), "mst_fx_range_reduce")
    res = out.cpu().numpy()
    if owner is None:
        return res
    full = np.zeros(r)
    np.maximum.at(full, np.asarray(owner), res)
    return full


class StftMeanMagnitude:
    """Mean |STFT| over frames (librosa.stft(center=False) framing), one channel at a time."""

    def __init__(self, n_fft, hop, window, max_batch=64):
        self.lib = _lib.lib()
        self.n_fft, self.hop = int(n_fft), int(hop)
        win = np.ascontiguousarray(window, dtype=np.float32)
        h = C.c_void_p()
        self.lib.check(self.lib.mst_fx_stft_create(self.n_fft, self.hop, win.ctypes.data_as(C.POINTER(C.c_float)), max_batch, C.byref(h)),
                       "mst_fx_stft_create")
        self.handle = h
        self.ws = None

    def __call__(self, x, channel=0):
        lib = self.lib
        L, Cn = x.shape
        with lib.device_ctx(x):
            nbytes = lib.mst_fx_stft_workspace_bytes(self.handle)
            if self.ws is None or self.ws.numel() < nbytes or self.ws.device != x.device:
                self.ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            out = torch.empty(self.n_fft // 2 + 1, dtype=torch.float32, device=x.device)
            lib.check(lib.mst_fx_stft_mean_magnitude(self.handle, x.data_ptr(), L, Cn, channel, out.data_ptr(), self.ws.data_ptr(), nbytes,
                                                     lib.stream_ptr(x)), "mst_fx_stft_mean_magnitude")
        return out.cpu().numpy()

    def __del__(self):
        try:
            self.lib.mst_fx_stft_destroy(self.handle)
        except Exception:
            pass


_convolvers = {}


def fir_causal(x, taps):
    """y[n] = sum_k taps[k] * x[n - k] with x[m < 0] := x[0] (an FIR started from the steady state of its first sample, what
    scipy.signal.lfilter(b, 1, x, zi=lfilter_zi(b, 1) * x[0]) computes): x device [L, 1] -> [L, 1].  FFT convolution."""
    lib = _lib.lib()
    nt = len(taps)
    L = x.shape[0]
    xe = torch.cat((x[:1].expand(nt - 1, 1), x), 0).contiguous()
    Le = xe.shape[0]
    h = torch.from_numpy(np.ascontiguousarray(taps, dtype=np.float32)[:, None])
    h = lib.to_device(h).to(x.device)
    key = (lib.path, Le, nt, str(x.device))
    with lib.device_ctx(x):
        cv = _convolvers.get(key)
        if cv is None:
            if len(_convolvers) > 4:
                for old in list(_convolvers.values()):
                    lib.mst_fx_convolver_destroy(old)
                _convolvers.clear()
            hdl = C.c_void_p()
            lib.check(lib.mst_fx_convolver_create(Le, nt, 1, 1, C.byref(hdl)), "mst_fx_convolver_create")
            cv = _convolvers[key] = hdl
        nbytes = lib.mst_fx_convolver_workspace_bytes(cv)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        y = torch.empty_like(xe)
        lib.check(lib.mst_fx_convolve(cv, xe.data_ptr(), h.data_ptr(), nt, y.data_ptr(), nt - 1, 0.0, 1.0, ws.data_ptr(), nbytes,
                                      lib.stream_ptr(x)), "mst_fx_convolve")
    return y[:L]


def compressor_grid(x, thresholds, ratios, attack_ms, release_ms, sample_rate, clip=True):
    """All (threshold, ratio) candidates on ONE signal: x device [L, C] -> device [n, L, C]."""
    lib = _lib.lib()
    n = len(thresholds)
    L, Cn = x.shape
    dev = x.device
    th = torch.tensor(thresholds, dtype=torch.float64, device=dev)
    ra = torch.tensor(ratios, dtype=torch.float64, device=dev)
    y = torch.empty(n, L, Cn, dtype=torch.float32, device=dev)
    with lib.device_ctx(x):
        nbytes = lib.mst_fx_compressor_scratch_bytes(n, L, Cn)
        sc = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=dev)
        pk = torch.empty(n * 64, dtype=torch.float64, device=dev) if clip else None
        lib.check(lib.mst_fx_compressor_grid(x.data_ptr(), y.data_ptr(), n, L, Cn, th.data_ptr(), ra.data_ptr(), float(attack_ms),
                                             float(release_ms), float(sample_rate), sc.data_ptr(), nbytes,
                                             pk.data_ptr() if clip else None, lib.stream_ptr(x)), "mst_fx_compressor_grid")
    return y


def onset_hfc(x, win, channel=0):
    """x device [n, L, C] -> numpy float32 [n, L // win, 2]: (hfc, mean square) of every whole frame of `win` samples."""
    lib = _lib.lib()
    if x.dim() == 2:
        x = x[None]
    n, L, Cn = x.shape
    nf = L // win
    out = torch.zeros(n, nf, 2, dtype=torch.float32, device=x.device)
    if nf:
        with lib.device_ctx(x):
            lib.check(lib.mst_fx_onset_hfc(x.data_ptr(), n, L, Cn, channel, win, out.data_ptr(), lib.stream_ptr(x)), "mst_fx_onset_hfc")
    return out.cpu().numpy()
