"""Onset times the way aubio.onset('hfc', buf_size=N, hop_size=N, samplerate=sr) reports them to get_mean_peak
(reference utils_data_normalization.py:304-314).  aubio==0.4.9 (requirements.txt:1) is a native library that is neither
vendored nor installable here; its onset detector is restated from its published sources (src/onset/onset.c,
src/onset/peakpicker.c, src/spectral/specdesc.c, src/temporal/filter.c) - parity unpinned, see DESIGN.md:

  * detection function per hop (computed on the MI355X, mst_fx_onset_hfc): HFC of the log-compressed magnitude spectrum;
  * peak picker: a 7-value sliding window of the detection function, low-pass filtered forwards and backwards
    (biquad b = (0.15998789, 0.31997577, 0.15998789), a1 = 0.23484048, a2 = 0), thresholded with
    value[5] - median - 0.058 * mean, a peak where the middle of the last three thresholded values is a positive local maximum,
    refined by quadratic interpolation;
  * onset logic: silence gate (-70 dB), minimum inter-onset interval 50 ms, detection delay 4.3 hops, and the
    beginning-of-file rule.
Everything here runs on a sequence of one value per hop (a few thousand per stem): host arithmetic.  The peak picker is evaluated for all
hops at once (`peak_picker_all`: the same float32 / float64 operations in the same order as the hop-by-hop `PeakPicker`, on a
[hops, 7] matrix of sliding windows); only the handful of hops that carry a peak go through the sequential onset logic.
"""
import numpy as np

_B = (0.15998789, 0.31997577, 0.15998789)
_A1, _A2 = 0.23484048, 0.0


def _biquad_inplace(v):
    """aubio_filter_do on a short buffer: direct form I, zero state, float64 accumulators, float32 samples."""
    x1 = x2 = y1 = y2 = 0.0
    for j in range(len(v)):
        x0 = float(v[j])
        y0 = _B[0] * x0 + _B[1] * x1 - _A1 * y1 + _B[2] * x2 - _A2 * y2
        v[j] = np.float32(y0)
        x2, x1, y2, y1 = x1, x0, y1, y0


class PeakPicker:
    def __init__(self, threshold=0.058, win_post=5, win_pre=1):
        self.threshold, self.win_post = np.float32(threshold), win_post
        self.keep = np.zeros(win_post + win_pre + 1, np.float32)
        self.peek = np.zeros(3, np.float32)

    def __call__(self, value):
        self.keep[:-1] = self.keep[1:]
        self.keep[-1] = value
        proc = self.keep.copy()
        _biquad_inplace(proc)                   # forward
        tmp = proc[::-1].copy()
        _biquad_inplace(tmp)                    # backward
        proc = tmp[::-1].copy()
        mean = np.float32(proc.sum(dtype=np.float32) / np.float32(len(proc)))
        median = np.sort(proc)[len(proc) // 2]
        self.peek[:-1] = self.peek[1:]
        self.peek[2] = proc[self.win_post] - median - mean * self.threshold
        s0, s1, s2 = (np.float32(v) for v in self.peek)
        if not (s1 > s0 and s1 > s2 and s1 > 0.0):
            return 0.0
        return float(np.float32(1.0) + np.float32(0.5) * (s0 - s2) / (s0 - np.float32(2.0) * s1 + s2))


def _biquad_rows(v):
    """_biquad_inplace along axis 1 of a [rows, n] float32 matrix (every row from zero state)."""
    z = np.zeros(v.shape[0])
    x1, x2, y1, y2 = z, z.copy(), z.copy(), z.copy()
    for j in range(v.shape[1]):
        x0 = v[:, j].astype(np.float64)
        y0 = _B[0] * x0 + _B[1] * x1 - _A1 * y1 + _B[2] * x2 - _A2 * y2
        v[:, j] = y0.astype(np.float32)
        x2, x1, y2, y1 = x1, x0, y1, y0


def peak_picker_all(values, threshold=0.058, win_post=5, win_pre=1):
    """PeakPicker()(v) for every v of `values` in order, as one array (0.0 where there is no peak)."""
    values = np.asarray(values, dtype=np.float32)
    n, w = len(values), win_post + win_pre + 1
    if n == 0:
        return np.zeros(0)
    padded = np.concatenate((np.zeros(w - 1, np.float32), values))
    proc = np.lib.stride_tricks.sliding_window_view(padded, w).copy()        # row f: the picker's window after hop f
    _biquad_rows(proc)                                                        # forward
    proc = np.ascontiguousarray(proc[:, ::-1])
    _biquad_rows(proc)                                                        # backward
    proc = np.ascontiguousarray(proc[:, ::-1])
    tot = proc[:, 0].copy()
    for j in range(1, w):                                                     # float32 running sum in index order
        tot = tot + proc[:, j]
    mean = tot / np.float32(w)
    median = np.sort(proc, axis=1)[:, w // 2]
    thr = proc[:, win_post] - median - mean * np.float32(threshold)          # float32
    s0 = np.concatenate((np.zeros(2, np.float32), thr[:-2])) if n > 2 else np.concatenate((np.zeros(2, np.float32), thr))[:n]
    s1 = np.concatenate((np.zeros(1, np.float32), thr[:-1]))
    s2 = thr
    peak = (s1 > s0) & (s1 > s2) & (s1 > 0.0)
    out = np.zeros(n)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.float32(1.0) + np.float32(0.5) * (s0 - s2) / (s0 - np.float32(2.0) * s1 + s2)
    out[peak] = q[peak].astype(np.float64)
    return out


def onset_times(hfc, mean_square, hop, samplerate, silence_db=-70.0, minioi_ms=50.0, delay_hops=4.3, threshold=0.058):
    """hfc, mean_square: one value per hop -> list of onset sample positions (aubio_onset_get_last after every detected onset)."""
    picked = peak_picker_all(hfc, threshold)
    minioi = int(round(minioi_ms / 1000.0 * samplerate))
    delay = int(delay_hops * hop)
    total, last = 0, 0
    out = []
    with np.errstate(divide="ignore"):
        db = 10.0 * np.log10(np.asarray(mean_square, dtype=np.float32))
    # hops without a peak only matter while total <= delay (the beginning-of-file rule)
    for f in sorted(set(np.flatnonzero(picked > 0.0).tolist()) | set(range(min(len(hfc), delay // hop + 1)))):
        total = f * hop
        isonset = float(picked[f])
        silent = bool(db[f] < silence_db)
        if isonset > 0.0:
            if silent:
                isonset = 0.0
            else:
                new_onset = total + int(round(isonset * hop))
                if last + minioi < new_onset:
                    if last > 0 and delay > new_onset:
                        isonset = 0.0
                    else:
                        last = max(delay, new_onset)
                else:
                    isonset = 0.0
        elif total <= delay and not silent:
            if total == 0 or last + minioi < total:       # beginning of the file
                isonset = delay / hop
                last = total + delay
        if isonset:
            out.append(max(0, last - delay))
    return out
