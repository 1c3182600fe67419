"""ORACLE (test infrastructure, NOT the product): numpy restatement of the FX-manipulator processors.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Restated from mixing_manipulator/common_audioeffects.py:
  compressor       <- :529-587 compressor_process (+ :637-649 Compressor.process per-channel call,
                      makeup 0, float32 result array because np.zeros_like(x) on float32 input)
  midside_imager   <- :964-1007 MidSideImager.process
  gain             <- :1041-1051 Gain.process
  rms_normalize    <- :143-146 AugmentationChain.apply_processor (rms_normalize branch)
  haas             <- :768-787 haas_process (np.roll => circular delay)
  panner_gains     <- :882-915 Panner._calculate_pan_coefficents
  equaliser        <- :500-525 Equaliser.process: cascade low_shelf, first/second/third_band (peaking),
                      high_shelf; each band reset_state() then whole-signal filtering; float32 cast.
All layouts are [L, C] (time-major, interleaved channels) like the reference's processors.
float64 internal arithmetic, float32 at processor boundaries.

Parity status:
  * compressor / imager / gain / haas / panner / rms: PINNED - tests/golden/make_golden.py imports the
    reference module (third-party imports stubbed, numba.jit -> identity) and commits vectors.
  * equaliser: PARITY UNPINNED.  The biquad coefficients and recursion live in the third-party
    dependency pymixconsole==0.0.1 (requirements.txt:10; components/iirfilter.py), which is neither
    vendored under /root/reference nor installed.  Restated here from its published algorithm:
    RBJ Audio-EQ-Cookbook coefficients (A = 10^(G/40), w0 = 2*pi*fc/rate, alpha = sin(w0)/(2Q)),
    normalised by a0, applied with scipy.signal.lfilter semantics (transposed direct form II,
    zero initial state per band because Equaliser.process resets state first).  Anchored on the
    reference's call sites common_audioeffects.py:441-462 (filter types, shelves Q=0.707) and :511-513.
"""
import numpy as np

BANDS = ("low_shelf", "first_band", "second_band", "third_band", "high_shelf")


# ----------------------------------------------------------------------------- compressor
def compressor_channel(x, threshold, attack_time, release_time, ratio, makeup_gain, sample_rate):
    """One channel, float64 arithmetic.  Quirks kept: yL_prev forced to 0 on entry (:553);
    |x| < 1e-6 -> -120 dB (:559-560); ratio == 1 leaves y_g = 0 (:564-573)."""
    x = np.asarray(x, dtype=np.float64)
    m = x.shape[0]
    a_att = np.exp(-1.0 / (0.001 * sample_rate * attack_time))
    a_rel = np.exp(-1.0 / (0.001 * sample_rate * release_time))
    ax = np.abs(x)
    x_g = np.where(ax < 0.000001, -120.0, 20.0 * np.log10(np.maximum(ax, 1e-300)))
    if ratio > 1:
        y_g = np.where(x_g >= threshold, threshold + (x_g - threshold) / ratio, x_g)
    elif ratio < 1:
        y_g = np.where(x_g <= threshold, threshold + (x_g - threshold) / (1.0 / ratio), x_g)
    else:
        y_g = np.zeros(m)
    x_l = x_g - y_g
    y_l = np.empty(m)
    prev = 0.0
    for i in range(m):
        xl = x_l[i]
        if xl > prev:
            prev = a_att * prev + (1.0 - a_att) * xl
        else:
            prev = a_rel * prev + (1.0 - a_rel) * xl
        y_l[i] = prev
    c = np.power(10.0, (makeup_gain - y_l) / 20.0)
    return x * c, prev


def compressor(x, threshold=-20.0, attack_time=2.0, release_time=100.0, ratio=4.0, sample_rate=44100):
    """Compressor.process: x [L, C] -> float32 [L, C]; bypass iff threshold == 0 and ratio == 1."""
    if threshold == 0.0 and ratio == 1.0:
        return x
    y = np.zeros_like(x)
    for ch in range(x.shape[1]):
        y[:, ch] = compressor_channel(x[:, ch], threshold, attack_time, release_time, ratio, 0.0, sample_rate)[0]
    return y


# ----------------------------------------------------------------------------- imager / gain / rms
def midside_imager(x, bal):
    left = x[:, 0]
    right = x[:, 1]
    mid = left + right
    side = left - right
    mid_e = np.sum(mid ** 2)
    side_e = np.sum(side ** 2)
    total_e = mid_e + side_e
    max_side = np.sqrt(total_e / (side_e + 1e-3))
    cur = round(bal, 3)
    side_gain = cur if cur <= 1.0 else max_side * (cur - 1)
    new_side = side * side_gain
    mid_gain = np.sqrt((total_e - side_e * side_gain ** 2) / (mid_e + 1e-3))
    new_mid = mid * mid_gain
    return np.stack([(new_mid + new_side) / 2, (new_mid - new_side) / 2], 1)


def gain(x, gain_db, invert=False):
    g = 10 ** (gain_db / 20.0)
    return (-g if invert else g) * x


def rms_normalize(x, y):
    scale = np.sqrt(np.mean(np.square(x)) / np.maximum(1e-7, np.mean(np.square(y))))
    return y * scale


def haas(x, delay, feedback, wet_channel="left"):
    y = np.copy(x)
    ch = 0 if wet_channel == "left" else 1
    y[:, ch] += feedback * np.roll(x[:, ch], delay)
    return y


def panner_gains(pan, pan_law="-4.5dB", dtype=np.float32):
    theta = pan * (np.pi / 2)
    g = np.zeros(2, dtype=dtype)
    if pan_law == "linear":
        g[0] = ((np.pi / 2) - theta) * (2 / np.pi)
        g[1] = theta * (2 / np.pi)
    elif pan_law == "constant_power":
        g[0] = np.cos(theta)
        g[1] = np.sin(theta)
    elif pan_law == "-4.5dB":
        g[0] = np.sqrt(((np.pi / 2) - theta) * (2 / np.pi) * np.cos(theta))
        g[1] = np.sqrt(theta * (2 / np.pi) * np.sin(theta))
    else:
        raise ValueError(f"Invalid pan_law {pan_law}.")
    return g


# ----------------------------------------------------------------------------- convolution reverb
def reverb_fade(h, decay, sample_rate, dtype=np.float32):
    """ConvolutionalReverb.update (common_audioeffects.py:714-725): 20 ms fade-out of the IR tail, starting `decay` of the way
    from the peak to the end; the IR is cut at the end of the fade."""
    h = np.array(h, copy=True)
    if decay >= 1.0:
        return h
    n = h.shape[0]
    peak = int(np.argmax(np.max(np.abs(h), axis=1), axis=0))
    # np.minimum yields NumPy integers: the ramp below then promotes exactly like the reference's (float64 under NumPy 2)
    fstart = np.minimum(n, peak + int(decay * (n - peak)))
    fstop = np.minimum(n, fstart + int(0.020 * sample_rate))
    flen = fstop - fstart
    fade = np.power(0.1, (np.arange(1, flen + 1, dtype=dtype) / flen) * 5)
    h[fstart:fstop, :] *= fade[:, np.newaxis]
    return h[:fstop]


def conv_reverb(x, h, dry=0.0, wet=1.0, pre_delay_ms=0, sample_rate=44100):
    """ConvolutionalReverb.process (:727-764) in float64: full linear convolution per channel (the reference calls
    scipy.signal.oaconvolve in the input precision), wet signal cut at the IR peak (+ pre-delay), dry/wet mix."""
    from scipy.signal import fftconvolve
    x64, h64 = np.asarray(x, np.float64), np.asarray(h, np.float64)
    if h64.shape[1] == 1 and x64.shape[1] > 1:
        h64 = np.hstack([h64] * x64.shape[1])
    if wet == 0.0:
        return np.asarray(x)
    y = fftconvolve(x64, h64, mode="full", axes=0)
    idx = int(np.argmax(np.max(np.abs(h64), axis=1), axis=0)) + int(0.001 * abs(pre_delay_ms) * sample_rate)
    idx = int(np.clip(idx, 0, h64.shape[0] - 1))
    return dry * x64 + wet * y[idx:idx + x64.shape[0], :]


# ----------------------------------------------------------------------------- equaliser (UNPINNED)
def rbj_biquad(filter_type, gain_db, q, fc, rate):
    """RBJ cookbook biquad -> (b[3], a[3]) normalised so a[0] == 1, float64."""
    A = 10.0 ** (gain_db / 40.0)
    w0 = 2.0 * np.pi * (fc / rate)
    alpha = np.sin(w0) / (2.0 * q)
    cw = np.cos(w0)
    sA = np.sqrt(A)
    if filter_type == "high_shelf":
        b0 = A * ((A + 1) + (A - 1) * cw + 2 * sA * alpha)
        b1 = -2 * A * ((A - 1) + (A + 1) * cw)
        b2 = A * ((A + 1) + (A - 1) * cw - 2 * sA * alpha)
        a0 = (A + 1) - (A - 1) * cw + 2 * sA * alpha
        a1 = 2 * ((A - 1) - (A + 1) * cw)
        a2 = (A + 1) - (A - 1) * cw - 2 * sA * alpha
    elif filter_type == "low_shelf":
        b0 = A * ((A + 1) - (A - 1) * cw + 2 * sA * alpha)
        b1 = 2 * A * ((A - 1) - (A + 1) * cw)
        b2 = A * ((A + 1) - (A - 1) * cw - 2 * sA * alpha)
        a0 = (A + 1) + (A - 1) * cw + 2 * sA * alpha
        a1 = -2 * ((A - 1) + (A + 1) * cw)
        a2 = (A + 1) + (A - 1) * cw - 2 * sA * alpha
    elif filter_type == "peaking":
        b0 = 1 + alpha * A
        b1 = -2 * cw
        b2 = 1 - alpha * A
        a0 = 1 + alpha / A
        a1 = -2 * cw
        a2 = 1 - alpha / A
    else:
        raise ValueError(filter_type)
    return np.array([b0, b1, b2]) / a0, np.array([1.0, a1 / a0, a2 / a0])


def biquad_tdf2(x, b, a):
    """scipy.signal.lfilter recursion for a 2nd-order section (transposed direct form II),
    zero initial state, along axis 0 of x [L] or [L, C].  float64."""
    x = np.asarray(x, dtype=np.float64)
    y = np.empty_like(x)
    z1 = np.zeros(x.shape[1:])
    z2 = np.zeros(x.shape[1:])
    for n in range(x.shape[0]):
        xn = x[n]
        yn = b[0] * xn + z1
        z1 = b[1] * xn - a[1] * yn + z2
        z2 = b[2] * xn - a[2] * yn
        y[n] = yn
    return y


EQ_DEFAULTS = {
    "low_shelf": (0.0, 80.0, 0.707), "first_band": (0.0, 400.0, 0.7), "second_band": (0.0, 2000.0, 0.7),
    "third_band": (0.0, 4000.0, 0.7), "high_shelf": (0.0, 8000.0, 0.707),
}


def equaliser_coeffs(params, rate=44100, bands=BANDS):
    """params: {band: (gain_db, freq, q)}; shelves always use Q = 0.707 (common_audioeffects.py:452-454).
    Returns float64 [n_bands, 6] rows (b0, b1, b2, 1, a1, a2)."""
    rows = []
    for band in bands:
        g, fc, q = params.get(band, EQ_DEFAULTS[band])
        if band in ("low_shelf", "high_shelf"):
            b, a = rbj_biquad(band, g, 0.707, fc, rate)
        else:
            b, a = rbj_biquad("peaking", g, q, fc, rate)
        rows.append(np.concatenate([b, a]))
    return np.stack(rows)


def equaliser(x, params, rate=44100, bands=BANDS, hard_clip=False, use_scipy=True):
    """Equaliser.process: x [L, C] -> float32 [L, C]."""
    coef = equaliser_coeffs(params, rate, bands)
    y = np.asarray(x, dtype=np.float64)
    for row in coef:
        if use_scipy:
            from scipy.signal import lfilter
            y = lfilter(row[:3], row[3:], y, axis=0)
        else:
            y = biquad_tdf2(y, row[:3], row[3:])
    if hard_clip:
        y = np.clip(y, -1.0, 1.0)
    return y.astype(np.float32)


# ----------------------------------------------------------------------------- config-4 chain
CONFIG4 = {
    "eq": {"low_shelf": (3.0, 80.0, 0.707), "first_band": (-4.0, 400.0, 0.7), "second_band": (2.0, 2000.0, 0.7),
           "third_band": (-3.0, 4000.0, 0.7), "high_shelf": (1.5, 8000.0, 0.707)},
    "comp": {"threshold": -20.0, "ratio": 4.0, "attack_time": 2.0, "release_time": 100.0},
    "imager_bal": 1.5,
    "gain_db": 3.0,
}


def fx_chain(x, cfg=CONFIG4, rate=44100, compressor_fn=None):
    """BASELINE config 4: EQ -> rms-norm -> compressor -> rms-norm -> imager -> rms-norm
    -> gain, the AugmentationChain.apply_processor sequence (common_audioeffects.py:115-148) with fixed
    parameters.  x float32 [L, 2] -> float32 [L, 2]."""
    comp = compressor_fn or compressor
    x = np.asarray(x, dtype=np.float32)
    y = rms_normalize(x, equaliser(x, cfg["eq"], rate))
    y = y.astype(np.float32)
    z = rms_normalize(y, comp(y, sample_rate=rate, **cfg["comp"]))
    z = z.astype(np.float32)
    w = rms_normalize(z, midside_imager(z, cfg["imager_bal"]))
    w = w.astype(np.float32)
    return gain(w, cfg["gain_db"]).astype(np.float32)


# ----------------------------------------------------------------------------- algorithmic reverb
def _comb(x, delay, damp, feedback):
    """Freeverb feedback comb with a one-pole low-pass in the loop (pymixconsole.components.comb, restated; parity unpinned)."""
    buf = np.zeros(delay)
    y = np.empty(len(x))
    store, idx = 0.0, 0
    for n in range(len(x)):
        out = buf[idx]
        store = out * (1.0 - damp) + store * damp
        buf[idx] = x[n] + store * feedback
        idx = idx + 1 if idx + 1 < delay else 0
        y[n] = out
    return y


def _allpass(x, delay, feedback):
    """Schroeder all-pass section in Freeverb's form (pymixconsole.components.allpass, restated; parity unpinned)."""
    buf = np.zeros(delay)
    y = np.empty(len(x))
    idx = 0
    for n in range(len(x)):
        out = buf[idx]
        y[n] = out - x[n]
        buf[idx] = x[n] + out * feedback
        idx = idx + 1 if idx + 1 < delay else 0
    return y


def algorithmic_reverb(x, room_size=0.5, damping=0.1, dry_mix=0.9, wet_mix=0.1, width=0.7, stereospread=23, scalegain=0.2):
    """AlgorithmicReverb.process (common_audioeffects.py:1447-1495) with the reference's quirks: the comb sum restarts at the fifth
    comb (:1467-1471), the fourth right all-pass is 255 + spread long (:1512).  x [L] / [L, 1] / [L, 2] -> float64 [L, 2]."""
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[:, None]
    dl, dr = x[:, 0].astype(np.float64), x[:, -1].astype(np.float64)
    wet = []
    for side, d in ((0, dl), (1, dr)):
        ss = stereospread * side
        v = sum(_comb(d * scalegain, D + ss, damping, room_size) for D in (1422, 1491, 1557, 1617))
        for D in (556, 441, 341, (225, 255)[side]):
            v = _allpass(v, D + ss, room_size)
        wet.append(v)
    wet1, wet2 = wet_mix * ((width / 2) + 0.5), wet_mix * ((1 - width) / 2)
    return np.stack([wet1 * wet[0] + wet2 * wet[1] + dry_mix * dl, wet1 * wet[1] + wet2 * wet[0] + dry_mix * dr], 1)
