"""EQ and compressor matching of the input normaliser (reference mixing_manipulator/utils_data_normalization.py:
get_eq_matching :65-107, get_mean_peak :284-338, compress :340-355, get_comp_matching :357-429).

Same function names, arguments and return values as the reference (numpy in, numpy out); the sample-rate work runs on
the MI355X:
  * get_eq_matching: loudness normalisation (BS.1770 meter), the mean STFT magnitude (the library's own FFT kernels), and the
    zero-phase 1001-tap FIR (scipy.signal.filtfilt semantics: odd extension by 3 * ntaps, forward + backward pass from the
    steady state of the first sample) as two FFT convolutions; scipy.signal.firwin2 designs the filter on the host like the
    reference does;
  * get_comp_matching: the reference tries ratio x threshold settings one at a time until the onset-peak statistic drops
    below the target; here a whole row of threshold candidates runs as ONE batch of the time-parallel compressor kernels on
    the same input (mst_fx_compressor_grid), their onset-detection functions and inter-onset peaks are reduced on the device,
    and the first candidate in the reference's scan order that satisfies the condition is returned - the same result as the
    sequential search.
"""
import numpy as np
import scipy.signal

from . import _device_ops as D
from . import fx_utils
from .onset import onset_times


def amp_to_db(x):
    return 20 * np.log10(x + 1e-30)


def db_to_amp(x):
    return 10 ** (x / 20)


# ------------------------------------------------------------------------------------------------ EQ matching
_stft_cache = {}


def _stft(n_fft, hop):
    key = (n_fft, hop)
    if key not in _stft_cache:
        _stft_cache.clear()
        _stft_cache[key] = D.StftMeanMagnitude(n_fft, hop, np.sqrt(np.hanning(n_fft + 1)[:-1]), max_batch=64)
    return _stft_cache[key]


def _filtfilt_fir(taps, x):
    """scipy.signal.filtfilt(taps, 1, x, padtype='odd', padlen=None, method='pad') on the device; x device [L, 1]."""
    import torch
    ntaps = len(taps)
    edge = 3 * ntaps
    L = x.shape[0]
    if L <= edge:
        raise ValueError(f"The length of the input vector x must be greater than padlen, which is {edge}.")
    left = 2 * x[0:1] - torch.flip(x[1:edge + 1], dims=(0,))
    right = 2 * x[L - 1:L] - torch.flip(x[L - edge - 1:L - 1], dims=(0,))
    ext = torch.cat((left, x, right), 0)
    y = D.fir_causal(ext, taps)
    y = D.fir_causal(torch.flip(y, dims=(0,)).contiguous(), taps)
    y = torch.flip(y, dims=(0,))
    return y[edge:edge + L]


def _is_device(x):
    import torch
    return isinstance(x, torch.Tensor)


def get_eq_matching(audio_t, ref_spec, sr=44100, n_fft=65536, hop_length=16384, min_db=-50, ntaps=101, lufs=-30):
    """audio_t: one channel [L]; ref_spec: the target mean magnitude spectrum [n_fft/2+1] -> the EQ-matched channel [L].
    numpy in -> numpy out (the reference's interface); a device tensor in -> a device tensor out (the device-resident normaliser)."""
    on_device = _is_device(audio_t)
    if on_device:
        max_db = amp_to_db(float(audio_t.abs().max()))
        if not max_db > min_db:
            return audio_t
        x = fx_utils.lufs_normalize(D.to_device(audio_t), sr, lufs, log=False)
    else:
        audio_t = np.copy(audio_t)
        max_db = amp_to_db(np.max(np.abs(audio_t)))
        if not max_db > min_db:
            return audio_t
        x = fx_utils.lufs_normalize(D.to_device(audio_t), sr, lufs, log=False)            # device [L, 1], float32
    audio_D_avg = _stft(n_fft, hop_length)(x, 0)
    m = ref_spec.shape[0]
    frq = np.arange(m) / (m / sr) / 2
    diff_eq = np.sqrt(db_to_amp(amp_to_db(ref_spec) - amp_to_db(audio_D_avg)))
    diff_filter = scipy.signal.firwin2(ntaps, frq / np.max(frq), diff_eq, nfreqs=None, window="hamming", antisymmetric=False)
    y = _filtfilt_fir(diff_filter, x)[:, 0]
    return y if on_device else y.cpu().numpy()


# ------------------------------------------------------------------------------------------------ compressor matching
_WIN = 2 ** 10


def _peak_stats(p_value, percentile):
    """mean / std of the onset peaks above the given percentile (all of them when none is above), or None."""
    if not len(p_value):
        return None
    thr = np.percentile(p_value, percentile)
    sel = [p for p in p_value if p > thr]
    use = sel if sel else list(p_value)
    return float(np.mean(use)), float(np.std(use))


def _mean_peak_device(y, sr, percentile):
    """y device [n, L, C] -> list (per item) of [mean peak dB, mean std] over channels, or None like get_mean_peak."""
    n, L, Cn = y.shape
    per_item = [[] for _ in range(n)]
    failed = [False] * n
    for ch in range(Cn):
        od = D.onset_hfc(y, _WIN, ch)                                     # [n, frames, 2]
        onsets = [onset_times(od[i, :, 0], od[i, :, 1], _WIN, sr) for i in range(n)]
        items, lo, hi = [], [], []
        for i, on in enumerate(onsets):
            for k, t in enumerate(on):
                items.append(i)
                lo.append(t)
                hi.append(on[k + 1] if k + 1 < len(on) else L)
        peaks = D.range_reduce(y, items, lo, hi, channel=ch, mode="max") if items else np.zeros(0)
        pos = 0
        for i, on in enumerate(onsets):
            st = _peak_stats(amp_to_db(peaks[pos:pos + len(on)]), percentile)
            pos += len(on)
            if st is None:
                failed[i] = True
            else:
                per_item[i].append(st)
    return [None if failed[i] else [float(np.mean([s[0] for s in per_item[i]])), float(np.mean([s[1] for s in per_item[i]]))]
            for i in range(n)]


def get_mean_peak(audio, sr=44100, true_peak=False, n_mels=128, percentile=75):
    """Mean onset-peak level in dB (peaks above the given percentile) and its spread; audio [samples, channels]."""
    if true_peak:
        raise NotImplementedError("true_peak=True (4x resampled peaks) is not used by the normaliser and not provided")
    return _mean_peak_device(D.to_device(np.asarray(audio))[None], sr, percentile)[0]


def compress(processor, audio, sr, th, ratio, attack, release):
    processor.parameters.threshold.value = th
    processor.parameters.ratio.value = ratio
    processor.parameters.attack_time.value = attack
    processor.parameters.release_time.value = release
    processor.update()
    output = processor.process(audio)
    if np.max(np.abs(output)) >= 1.0:
        output = np.clip(output, -1.0, 1.0)
    return output


def get_comp_matching(audio, ref_peak, ref_std, ratio, attack, release, sr=44100, min_db=-50, comp_peak_norm=-10.0, min_th=-40,
                      max_ratio=20, n_mels=128, true_peak=False, percentile=75, expander=True, batch=16):
    on_device = _is_device(audio)
    out = (lambda t: t) if on_device else (lambda t: t.cpu().numpy())
    if on_device:                                   # device in -> device out: the same steps, the same float32 arithmetic
        x = D.to_device(audio)
        mx = float(x.abs().max())
        if not amp_to_db(mx) > min_db:
            return x
        x = x * np.float32(np.power(10.0, comp_peak_norm / 20.0) / np.float32(mx))
        xd = x
    else:
        x = audio.copy()
        if x.ndim < 2:
            x = np.expand_dims(x, 1)
        max_db = amp_to_db(np.max(np.abs(x)))
        if not max_db > min_db:
            return x
        gain = np.power(10.0, comp_peak_norm / 20.0) / np.max(np.abs(x))              # pyloudnorm.normalize.peak
        x = x * (np.float32(gain) if x.dtype == np.float32 else gain)                 # a float32 signal stays float32 (NumPy 1.x promotion)
        xd = D.to_device(x)
    peak, std = _mean_peak_device(xd[None], sr, percentile)[0]
    if (ref_peak - ref_std) < peak < (ref_peak + ref_std):
        return x
    down = peak > (ref_peak - ref_std)
    if not down and not (expander and peak < (ref_peak + ref_std)):
        return x
    ratios = np.linspace(ratio, max_ratio, max_ratio - ratio + 1)
    if down:
        ths = np.linspace(-1 - 9, min_th, 2 * np.abs(min_th) - 1 - 18)
    else:
        ths = np.linspace(-1, min_th, 2 * np.abs(min_th) - 1)[::-1]
    # the reference's scan order: ratios outer, thresholds inner, stop at the first setting that meets the target.  Here a
    # chunk of `batch` consecutive settings of that order is evaluated at once; the answer is the first that qualifies.
    last = xd
    for rt in ratios:
        for k0 in range(0, len(ths), batch):
            th_chunk = ths[k0:k0 + batch]
            y = D.compressor_grid(xd, list(th_chunk), [rt if down else 1.0 / rt] * len(th_chunk), attack, release, sr, clip=True)
            stats = _mean_peak_device(y, sr, percentile)
            for i, st in enumerate(stats):
                if st is None:                      # the reference would fail on `peak, std = None` here (caught by the caller)
                    raise TypeError("cannot unpack non-iterable NoneType object")
                if (down and st[0] < (ref_peak + ref_std)) or (not down and st[0] > (ref_peak - ref_std)):
                    return out(y[i])
            last = y[len(th_chunk) - 1]
    return out(last)                                # no setting qualified: the reference returns the last one it tried
