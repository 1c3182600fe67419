"""FX-manipulator processors on MI355X, behind the reference's processor API
(mixing_manipulator/common_audioeffects.py of the reference: AugmentationChain :91-201, Equaliser :370-525,
Compressor :590-661, MidSideImager :956-1007, Gain :1011-1051).

Same vocabulary as the reference: `Processor.process(x)`, `processor.parameters.<name>.value`, `update()`,
`randomize()`, `AugmentationChain(fxs=[(processor, probability, rms_normalize), ...])(x_list)`.
pymixconsole is not a dependency: Parameter / ParameterList / Processor are minimal local equivalents.

Audio layout: [L, C] (time-major, interleaved) like the reference, or a batch [n_items, L, C]; float32.
numpy in -> numpy out (drop-in); a CUDA torch tensor in -> a CUDA tensor out (stays on the device).
All arithmetic runs in libmst_hip.so (csrc/fx_kernels.h); there is no numpy fallback.
"""
import ctypes as C
import math
import random

import numpy as np
import torch

from .. import _lib


SUMSQ_SLOTS = 64       # MST_SUMSQ_SLOTS of include/mst_hip.h: energy sums travel as 64 partial sums per item


class Parameter:
    def __init__(self, name, value, kind, units=None, minimum=None, maximum=None, options=None, processor=None, **kw):
        self.name, self.value, self.kind, self.units = name, value, kind, units
        self.min, self.max, self.options, self.default = minimum, maximum, options, value

    def randomize(self):
        if self.kind == "float":
            self.value = random.uniform(self.min, self.max)
        elif self.kind == "int":
            # exclusive upper bound (pymixconsole draws np.random.randint(min, max)): ConvolutionalReverb declares
            # index.maximum = len(impulse_responses) and indexes the list with the drawn value
            self.value = random.randrange(self.min, self.max) if self.max > self.min else self.min
        elif self.kind == "bool":
            self.value = random.random() < 0.5
        elif self.kind == "string":
            self.value = random.choice(self.options)

    def __repr__(self):
        return f"Parameter({self.name!r}={self.value!r})"


class ParameterList:
    def __init__(self):
        self._names = []

    def add(self, p):
        self._names.append(p.name)
        setattr(self, p.name, p)

    def __iter__(self):
        return (getattr(self, n) for n in self._names)

    def __repr__(self):
        return "ParameterList(" + ", ".join(repr(p) for p in self) + ")"


class Processor:
    def __init__(self, name, parameters, block_size, sample_rate, dtype="float32"):
        self.name, self.parameters = name, parameters
        self.block_size, self.sample_rate, self.dtype = block_size, sample_rate, dtype

    def randomize(self):
        for p in self.parameters:
            p.randomize()
        self.update(None)

    def update(self, parameter_name=None):
        pass

    def __repr__(self):
        return f"Processor(name={self.name!r}, parameters={self.parameters!r}"


# ---------------------------------------------------------------------------------------------- device glue
class _Dev:
    """Moves one processor call onto the device and back in the caller's container type."""

    def __init__(self, x):
        self.lib = _lib.lib()
        self.numpy = isinstance(x, np.ndarray)
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if self.numpy else x
        if t.dtype != torch.float32:
            t = t.float()
        self.batched = t.dim() == 3
        if t.dim() == 1:
            t = t[:, None]
        if not self.batched:
            t = t[None]
        self.x = self.lib.to_device(t).contiguous()
        self.n, self.L, self.C = self.x.shape
        self.stream = self.lib.stream_ptr(self.x)

    def rebind(self, t):
        """The batch this call works on changes along a chain (a processor may turn mono into stereo)."""
        self.x = t
        self.n, self.L, self.C = t.shape
        return self

    def scratch(self, n_doubles):
        return torch.empty(n_doubles, dtype=torch.float64, device=self.x.device)

    def fuse(self, in_scale, want_sumsq, in_sumsq=None, post_gain=None, want_in_sumsq=False, want_ms=False, in_ms=None, forms=0):
        """(MstFxFuse pointer or None, sumsq tensor or None) for a processor call inside a fused chain.
        in_sumsq + post_gain: tail folding (the rms-normalise behind the processor and a gain behind that in the processor's last pass).
        want_ms (compressor): the mid / side energies of the raw output are left in self.last_ms; in_ms (imager): such an array, instead of
        the imager's own energy pass over the audio."""
        self.last_in_sumsq = None
        self.last_ms = None
        if in_scale is None and not want_sumsq and in_sumsq is None and not want_in_sumsq and in_ms is None and not forms:
            return None, None
        if want_ms:             # [n][SUMSQ_SLOTS][2], cleared by the producer
            self.last_ms = torch.empty(self.n * SUMSQ_SLOTS * 2, dtype=torch.float64, device=self.x.device)
        sumsq = torch.empty(self.n * SUMSQ_SLOTS, dtype=torch.float64, device=self.x.device) if want_sumsq else None   # cleared by the producer
        if want_in_sumsq:       # sum(x_raw^2) of this call's input, left behind by the processor (the equaliser's apply pass)
            self.last_in_sumsq = torch.empty(self.n * SUMSQ_SLOTS, dtype=torch.float64, device=self.x.device)
        f = _lib.MstFxFuse(in_scale.data_ptr() if in_scale is not None else None, sumsq.data_ptr() if want_sumsq else None,
                           in_sumsq.data_ptr() if in_sumsq is not None else None, 1 if in_sumsq is not None else 0,
                           float(post_gain) if post_gain is not None else 1.0,
                           self.last_in_sumsq.data_ptr() if want_in_sumsq else None,
                           self.last_ms.data_ptr() if want_ms else None, in_ms.data_ptr() if in_ms is not None else None, forms=forms)
        self._keep = (f, in_scale, sumsq, in_sumsq, self.last_in_sumsq, self.last_ms, in_ms)    # alive until the launches are queued
        return C.byref(f), sumsq

    def out(self, y):
        if not self.batched:
            y = y[0]
        return y.cpu().numpy() if self.numpy else y


def rms_normalize_(x, y):
    """In-place on y: y *= sqrt(mean(x^2) / max(1e-7, mean(y^2))) per item (reference apply_processor :143-146)."""
    d = _Dev(x)
    dy = _Dev(y)                 # y may have another channel count than x (Panner / Haas turn mono into stereo): the
    if dy.x.device != d.x.device:    # reference takes scalar means over each array, so each side uses its own size
        dy.x = dy.x.to(d.x.device)
    if dy.n != d.n:
        raise ValueError("rms_normalize_: x and y hold different numbers of items")
    sc = d.scratch(4 * d.n)
    d.lib.check(d.lib.mst_fx_rms_normalize(d.x.data_ptr(), dy.x.data_ptr(), d.n, d.L * d.C, dy.L * dy.C, sc.data_ptr(), d.stream),
                "mst_fx_rms_normalize")
    return dy.out(dy.x)


# ---------------------------------------------------------------------------------------------- processors
def rbj_coefficients(filter_type, gain_db, q, fc, rate):
    """RBJ audio-EQ-cookbook biquad (b0,b1,b2,a0,a1,a2), un-normalised, float64 - the published algorithm of
    pymixconsole==0.0.1 components/iirfilter.py (not vendored by the reference; see DESIGN.md, parity unpinned)."""
    A = 10.0 ** (gain_db / 40.0)
    w0 = 2.0 * math.pi * (fc / rate)
    alpha = math.sin(w0) / (2.0 * q)
    cw, sA = math.cos(w0), math.sqrt(A)
    if filter_type == "high_shelf":
        return (A * ((A + 1) + (A - 1) * cw + 2 * sA * alpha), -2 * A * ((A - 1) + (A + 1) * cw),
                A * ((A + 1) + (A - 1) * cw - 2 * sA * alpha), (A + 1) - (A - 1) * cw + 2 * sA * alpha,
                2 * ((A - 1) - (A + 1) * cw), (A + 1) - (A - 1) * cw - 2 * sA * alpha)
    if filter_type == "low_shelf":
        return (A * ((A + 1) - (A - 1) * cw + 2 * sA * alpha), 2 * A * ((A - 1) - (A + 1) * cw),
                A * ((A + 1) - (A - 1) * cw - 2 * sA * alpha), (A + 1) + (A - 1) * cw + 2 * sA * alpha,
                -2 * ((A - 1) + (A + 1) * cw), (A + 1) + (A - 1) * cw - 2 * sA * alpha)
    if filter_type == "peaking":
        return (1 + alpha * A, -2 * cw, 1 - alpha * A, 1 + alpha / A, -2 * cw, 1 - alpha / A)
    raise ValueError(f"unknown filter type {filter_type}")


class Equaliser(Processor):
    """Five-band parametric EQ: low shelf, three peaking bands, high shelf (shelves at Q = 0.707), a cascade of
    RBJ biquads each applied to the whole signal from zero state."""

    def __init__(self, n_channels, sample_rate, gain_range=(-15.0, 15.0), q_range=(0.1, 2.0),
                 bands=("low_shelf", "first_band", "second_band", "third_band", "high_shelf"), hard_clip=False,
                 name="Equaliser", parameters=None):
        super().__init__(name, parameters=parameters, block_size=None, sample_rate=sample_rate)
        self.n_channels = n_channels
        lo, hi = gain_range
        qlo, qhi = q_range
        if not parameters:
            self.parameters = ParameterList()
            spec = [("low_shelf", 80.0, 30.0, 200.0, None), ("first_band", 400.0, 200.0, 1000.0, 0.7),
                    ("second_band", 2000.0, 1000.0, 3000.0, 0.7), ("third_band", 4000.0, 3000.0, 8000.0, 0.7),
                    ("high_shelf", 8000.0, 5000.0, 10000.0, None)]
            for band, f0, fmin, fmax, q0 in spec:
                self.parameters.add(Parameter(band + "_gain", 0.0, "float", minimum=lo, maximum=hi))
                self.parameters.add(Parameter(band + "_freq", f0, "float", minimum=fmin, maximum=fmax))
                if q0 is not None:
                    self.parameters.add(Parameter(band + "_q", q0, "float", minimum=qlo, maximum=qhi))
        self.bands = list(bands)
        self.hard_clip = hard_clip

    def coefficients(self):
        rows = []
        for band in self.bands:
            g = getattr(self.parameters, band + "_gain").value
            fc = getattr(self.parameters, band + "_freq").value
            if band in ("low_shelf", "high_shelf"):
                rows.append(rbj_coefficients(band, g, 0.707, fc, self.sample_rate))
            else:
                rows.append(rbj_coefficients("peaking", g, getattr(self.parameters, band + "_q").value, fc, self.sample_rate))
        return np.asarray(rows, dtype=np.float64)

    def reset_state(self):
        pass   # every process() call starts each band from zero state, like the reference (:511-513)

    def process(self, x):
        d = _Dev(x)
        y, _ = self._run(d, None, False)
        return d.out(y)

    def fusable(self, d):
        return not self.hard_clip and d.L > 1024 and len(self.bands) >= 1       # the time-parallel path, output left as filtered

    def _run(self, d, in_scale, want_sumsq, want_in_sumsq=False):
        coef = np.ascontiguousarray(self.coefficients())
        y = torch.empty_like(d.x)
        nbytes = d.lib.mst_fx_biquad_scratch_bytes(d.n, d.L, d.C, coef.shape[0])      # time-parallel (chunked scan) path
        sc = d.scratch((nbytes + 7) // 8)
        fuse, sumsq = d.fuse(in_scale, want_sumsq, want_in_sumsq=want_in_sumsq, forms=getattr(self, "kernel_forms", 0))      # kernel_forms: _lib.FX_FORM_* (tests)
        d.lib.check(d.lib.mst_fx_biquad_cascade(d.x.data_ptr(), y.data_ptr(), d.n, d.L, d.C,
                                                coef.ctypes.data_as(C.POINTER(C.c_double)), coef.shape[0],
                                                sc.data_ptr(), nbytes, fuse, d.stream), "mst_fx_biquad_cascade")
        if self.hard_clip:
            y = y.clamp_(-1.0, 1.0)
        return y, sumsq


class Compressor(Processor):
    """Single-band dynamic range compressor: log-domain gain computer + branchy one-pole attack/release
    smoother per channel (makeup gain 0)."""

    def __init__(self, sample_rate, name="Compressor", parameters=None):
        super().__init__(name=name, parameters=parameters, block_size=None, sample_rate=sample_rate)
        if not parameters:
            self.parameters = ParameterList()
            self.parameters.add(Parameter("threshold", -20.0, "float", units="dB", minimum=-80.0, maximum=-5.0))
            self.parameters.add(Parameter("attack_time", 2.0, "float", units="ms", minimum=1.0, maximum=20.0))
            self.parameters.add(Parameter("release_time", 100.0, "float", units="ms", minimum=50.0, maximum=500.0))
            self.parameters.add(Parameter("ratio", 4.0, "float", minimum=4.0, maximum=40.0))
        self.yL_prev = None

    def process(self, x):
        p = self.parameters
        if p.threshold.value == 0.0 and p.ratio.value == 1.0:
            return x
        d = _Dev(x)
        y, _ = self._run(d, None, False)
        return d.out(y)

    def fusable(self, d):
        p = self.parameters
        return not (p.threshold.value == 0.0 and p.ratio.value == 1.0)

    def _run(self, d, in_scale, want_sumsq):
        p = self.parameters
        y = torch.empty_like(d.x)
        nbytes = d.lib.mst_fx_compressor_scratch_bytes(d.n, d.L, d.C)
        sc = d.scratch((nbytes + 7) // 8)
        # stereo, inside a chain: the apply pass also leaves the mid / side energies of its output behind (d.last_ms) - an imager that
        # follows needs no energy pass of its own
        fuse, sumsq = d.fuse(in_scale, want_sumsq, want_ms=bool(want_sumsq and d.C == 2), forms=getattr(self, "kernel_forms", 0))      # kernel_forms: _lib.FX_FORM_* (tests)
        d.lib.check(d.lib.mst_fx_compressor(d.x.data_ptr(), y.data_ptr(), d.n, d.L, d.C, float(p.threshold.value),
                                            float(p.attack_time.value), float(p.release_time.value), float(p.ratio.value),
                                            float(self.sample_rate), sc.data_ptr(), nbytes, fuse, d.stream), "mst_fx_compressor")
        return y, sumsq

    def update(self, parameter_name=None):
        self.yL_prev = None


class MidSideImager(Processor):
    """Mid/side energy re-balancing; bal in [0, 1] narrows, (1, 2] widens.  Stereo input only."""

    def __init__(self, name="IMAGER", parameters=None):
        super().__init__(name, parameters=parameters, block_size=None, sample_rate=None)
        if not parameters:
            self.parameters = ParameterList()
            self.parameters.add(Parameter("bal", 0.0, "float", processor=self, minimum=0.0, maximum=2.0))

    def process(self, data):
        d = _Dev(data)
        y, _ = self._run(d, None, False)
        return d.out(y)

    def fusable(self, d):
        return d.C == 2

    def _run(self, d, in_scale, want_sumsq, bal=None, in_sumsq=None, post_gain=None, in_ms=None):
        if d.C != 2:
            raise ValueError("MidSideImager needs stereo audio [L, 2]")
        y = torch.empty_like(d.x)
        sc = d.scratch(2 * SUMSQ_SLOTS * d.n)
        fuse, sumsq = d.fuse(in_scale, want_sumsq, in_sumsq, post_gain, in_ms=in_ms)
        d.lib.check(d.lib.mst_fx_midside_imager(d.x.data_ptr(), y.data_ptr(), d.n, d.L,
                                                float(self.parameters.bal.value if bal is None else bal),
                                                sc.data_ptr(), fuse, d.stream), "mst_fx_midside_imager")
        return y, sumsq


class Gain(Processor):
    """Gain in dB, optional polarity inversion."""

    def __init__(self, name="Gain", parameters=None):
        super().__init__(name, parameters=parameters, block_size=None, sample_rate=None)
        if not parameters:
            self.parameters = ParameterList()
            self.parameters.add(Parameter("gain", 1.0, "float", units="dB", minimum=-6.0, maximum=9.0))
            self.parameters.add(Parameter("invert", False, "bool"))

    def process(self, x):
        d = _Dev(x)
        y, _ = self._run(d, None, False)
        return d.out(y)

    def fusable(self, d):
        return True

    def factor(self):
        """The float32 multiplier mst_fx_gain applies: float(10 ** (gain_db / 20)), negated when inverting."""
        g = math.pow(10.0, float(self.parameters.gain.value) / 20.0)
        return -g if bool(self.parameters.invert.value) else g

    def _run(self, d, in_scale, want_sumsq):
        y = torch.empty_like(d.x)
        fuse, _ = d.fuse(in_scale, False)              # a gain leaves no energy sum behind (nothing downstream asks for one cheaply)
        d.lib.check(d.lib.mst_fx_gain(d.x.data_ptr(), y.data_ptr(), d.n, d.L, d.C, float(self.parameters.gain.value),
                                      int(bool(self.parameters.invert.value)), fuse, d.stream), "mst_fx_gain")
        return y, None


class ConvolutionalReverb(Processor):
    """Convolution reverb (reference common_audioeffects.py:665-764): the input is convolved (full linear convolution, per
    channel) with one of the given impulse responses, the wet signal is cut out starting at the IR's peak (+ pre-delay) and
    mixed `dry * x + wet * y`.  The convolution runs on the device (mst_fx_convolve: the library's own FFT kernels);
    IR selection, the optional decay fade and the mono/stereo adaptation are the reference's host-side steps.

    impulse_responses: list (one entry per RT60 group) of lists of dicts whose 'impulse_response' entry is a callable
    returning an [n_samples, n_channels] array - the structure `create_dataset` produces in the reference."""

    def __init__(self, impulse_responses, sample_rate, name="ConvolutionalReverb", parameters=None):
        super().__init__(name, parameters=parameters, block_size=None, sample_rate=sample_rate)
        if impulse_responses is None:
            raise ValueError("List of impulse responses must be provided for ConvolutionalReverb processor.")
        self.impulse_responses = impulse_responses
        self._convolvers = {}
        if not parameters:
            self.parameters = ParameterList()
            self.max_ir_num = len(max(impulse_responses, key=len))
            self.parameters.add(Parameter("index", 0, "int", minimum=0, maximum=len(impulse_responses)))
            self.parameters.add(Parameter("index_ir", 0, "int", minimum=0, maximum=self.max_ir_num))
            self.parameters.add(Parameter("wet", 1.0, "float", minimum=1.0, maximum=1.0))
            self.parameters.add(Parameter("dry", 0.0, "float", minimum=0.0, maximum=0.0))
            self.parameters.add(Parameter("decay", 1.0, "float", minimum=1.0, maximum=1.0))
            self.parameters.add(Parameter("pre_delay", 0, "int", units="ms", minimum=0, maximum=0))

    def update(self, parameter_name=None):
        group = self.impulse_responses[self.parameters.index.value]
        entry = group[self.parameters.index_ir.value % len(group)]
        h = np.array(entry["impulse_response"](), copy=True)
        decay = self.parameters.decay.value
        if decay < 1.0:          # fade the tail out over 20 ms, starting `decay` of the way from the peak to the end
            n = h.shape[0]
            peak = int(np.argmax(np.max(np.abs(h), axis=1), axis=0))
            # np.minimum yields NumPy integers: the ramp then promotes exactly like the reference's (float64 under NumPy 2)
            fstart = np.minimum(n, peak + int(decay * (n - peak)))
            fstop = np.minimum(n, fstart + int(0.020 * self.sample_rate))
            flen = fstop - fstart
            ramp = np.arange(1, flen + 1, dtype=self.dtype) / flen
            h[fstart:fstop, :] *= np.power(0.1, ramp * 5)[:, np.newaxis]
            h = h[:fstop]
        self.h = h

    def __del__(self):
        try:
            lib = _lib.lib()
            for cv, _ in self._convolvers.values():
                lib.mst_fx_convolver_destroy(cv)
        except Exception:
            pass

    def _convolver(self, lib, L, Lh, n, Cn, device):
        key = (lib.path, L, n, Cn, str(device))
        cur = self._convolvers.get(key)
        if cur is None or cur[1] < Lh:
            if cur is not None:
                lib.mst_fx_convolver_destroy(cur[0])
            h = C.c_void_p()
            cap = max(Lh, 1 << max(0, (Lh - 1).bit_length()))      # room for longer IRs of later calls
            lib.check(lib.mst_fx_convolver_create(L, cap, n, Cn, C.byref(h)), "mst_fx_convolver_create")
            cur = (h, cap)
            self._convolvers[key] = cur
        return cur[0]

    def process(self, x):
        d = _Dev(x)
        if not hasattr(self, "h"):
            self.update()
        if self.h.shape[1] == 1 and d.C > 1:
            self.h = np.hstack([self.h] * d.C)                      # mono IR on multi-channel audio
        if self.h.shape[1] > 1 and d.C == 1:
            self.h = self.h[:, np.random.randint(self.h.shape[1]), np.newaxis]     # one IR channel, chosen at random
        if self.parameters.wet.value == 0.0:
            return d.out(d.x.clone())
        if self.h.shape[1] != d.C:
            raise ValueError(f"impulse response has {self.h.shape[1]} channels, audio has {d.C}")
        # the response on the device and its peak position are kept until update() (or the channel adaptation above) replaces self.h:
        # the float32 copy, the peak search and a 0.5 MB pageable upload cost as much per call as the transforms themselves
        # (an in-place edit of the public array - `rv.h *= g` - must not leave the old response on the device: the key carries a cheap
        #  content fingerprint, 64 strided samples and the array's sum, besides its identity; the reference recomputes from self.h every call)
        flat = self.h.reshape(-1)
        probe = flat[::max(1, flat.shape[0] // 64)][:64]
        key = (id(self.h), self.h.shape, self.parameters.pre_delay.value, str(d.x.device), probe.tobytes(), float(flat.sum(dtype=np.float64)))
        cached = getattr(self, "_h_cache", None)
        if cached is None or cached[0] != key:
            h32 = np.ascontiguousarray(self.h, dtype=np.float32)
            idx = int(np.argmax(np.max(np.abs(self.h), axis=1), axis=0))
            idx += int(0.001 * np.abs(self.parameters.pre_delay.value) * self.sample_rate)
            idx = int(np.clip(idx, 0, h32.shape[0] - 1))
            cached = self._h_cache = (key, torch.from_numpy(h32).to(d.x.device), idx, h32.shape[0], self.h)      # self.h kept alive: its id is the key
        _, hd, idx, lh, _ = cached
        cv = self._convolver(d.lib, d.L, lh, d.n, d.C, d.x.device)
        nbytes = d.lib.mst_fx_convolver_workspace_bytes(cv)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=d.x.device)
        y = torch.empty_like(d.x)
        d.lib.check(d.lib.mst_fx_convolve(cv, d.x.data_ptr(), hd.data_ptr(), lh, y.data_ptr(), idx,
                                          float(self.parameters.dry.value), float(self.parameters.wet.value), ws.data_ptr(),
                                          nbytes, d.stream), "mst_fx_convolve")
        return d.out(y)


class Haas(Processor):
    """Haas effect: one channel gets `feedback` times a circularly delayed copy of itself added (reference
    common_audioeffects.py:768-856; np.roll wraps around, the delay may be negative).  Mono input becomes stereo."""

    def __init__(self, sample_rate, delay_range=(-0.040, 0.040), name="Haas", parameters=None):
        super().__init__(name, parameters=parameters, block_size=None, sample_rate=sample_rate)
        if not parameters:
            self.parameters = ParameterList()
            self.parameters.add(Parameter("delay", int(delay_range[1] * sample_rate), "int", units="samples",
                                          minimum=int(delay_range[0] * sample_rate), maximum=int(delay_range[1] * sample_rate)))
            self.parameters.add(Parameter("feedback", 0.35, "float", minimum=0.33, maximum=0.66))
            self.parameters.add(Parameter("wet_channel", "left", "string", options=["left", "right"]))

    def process(self, x):
        d = _Dev(x)
        assert d.C == 1 or d.C == 2, "Haas effect only works with monaural or stereo audio."
        wet = self.parameters.wet_channel.value
        y = torch.empty((d.n, d.L, 2), dtype=torch.float32, device=d.x.device)
        if wet not in ("left", "right"):          # the reference's if/elif falls through: plain copy (:782-785)
            y.copy_(d.x.expand(d.n, d.L, 2))
            return d.out(y)
        d.lib.check(d.lib.mst_fx_haas(d.x.data_ptr(), y.data_ptr(), d.n, d.L, d.C, int(self.parameters.delay.value),
                                      float(self.parameters.feedback.value), 0 if wet == "left" else 1, d.stream),
                    "mst_fx_haas")
        return d.out(y)

    def update(self, parameter_name=None):
        self.reset_state()

    def reset_state(self):
        pass                                       # the reference's ring buffer fields are never read by process()


class Panner(Processor):
    """Stereo panner, pan in [0, 1] (0 = left), laws '-4.5dB' (default), 'linear', 'constant_power'
    (reference common_audioeffects.py:860-952).  Mono input becomes stereo."""

    def __init__(self, name="Panner", parameters=None):
        super().__init__(name, parameters=parameters, block_size=None, sample_rate=None)
        if not parameters:
            self.parameters = ParameterList()
            self.parameters.add(Parameter("pan", 0.5, "float", minimum=0.0, maximum=1.0))
            self.parameters.add(Parameter("pan_law", "-4.5dB", "string", options=["-4.5dB", "linear", "constant_power"]))
        self.update()

    def _calculate_pan_coefficents(self):
        """Left/right gains of the chosen law; kept in the processor dtype (float32) like the reference's self.gains."""
        frac = float(self.parameters.pan.value)                  # 0 = hard left ... 1 = hard right
        theta = frac * (np.pi / 2)
        lin = np.array([((np.pi / 2) - theta) * (2 / np.pi), theta * (2 / np.pi)])
        trig = np.array([np.cos(theta), np.sin(theta)])
        law = self.parameters.pan_law.value
        table = {"linear": lambda: lin, "constant_power": lambda: trig, "-4.5dB": lambda: np.sqrt(lin * trig)}
        if law not in table:
            raise ValueError(f"Invalid pan_law {law}.")
        self.gains = table[law]().astype(self.dtype)

    def process(self, x):
        d = _Dev(x)
        assert d.C == 1 or d.C == 2, "Panner only works with monaural or stereo audio."
        y = torch.empty((d.n, d.L, 2), dtype=torch.float32, device=d.x.device)
        d.lib.check(d.lib.mst_fx_panner(d.x.data_ptr(), y.data_ptr(), d.n, d.L, d.C, float(self.gains[0]),
                                        float(self.gains[1]), d.stream), "mst_fx_panner")
        return d.out(y)

    def update(self, parameter_name=None):
        self._calculate_pan_coefficents()

    def reset_state(self):
        self.update()



class AlgorithmicReverb(Processor):
    """Schroeder / Freeverb-style algorithmic reverb (reference common_audioeffects.py:1429-1537): per side eight damped feedback
    comb filters (the right side's delays are 23 samples longer) and four all-pass sections in series, then a width-dependent
    wet cross-mix plus the dry signal.  Kept from the reference: the comb sum restarts at the fifth comb (`xL = combL5...`
    overwrites the sum of the first four, :1467-1471), so only combs 5..8 reach the output; the fourth right all-pass is 255 + 23
    samples long (:1512).  The filters start from silence on every call (the reference rebuilds them in update()).
    The comb / all-pass arithmetic is pymixconsole's (not vendored): restated from the published structure, parity unpinned.
    Output float32 [L, 2] (the reference returns float64)."""

    COMB_DELAYS = (1116, 1188, 1277, 1356, 1422, 1491, 1557, 1617)
    ALLPASS_DELAYS = ((556, 556), (441, 441), (341, 341), (225, 255))      # (left, right before the stereo spread)

    def __init__(self, name="algoreverb", parameters=None, sample_rate=44100, **kwargs):
        super().__init__(name=name, parameters=parameters, block_size=None, sample_rate=sample_rate)
        if not parameters:
            self.parameters = ParameterList()
            self.parameters.add(Parameter("room_size", 0.5, "float", minimum=0.05, maximum=0.85))
            self.parameters.add(Parameter("damping", 0.1, "float", minimum=0.0, maximum=1.0))
            self.parameters.add(Parameter("dry_mix", 0.9, "float", minimum=0.0, maximum=1.0))
            self.parameters.add(Parameter("wet_mix", 0.1, "float", minimum=0.0, maximum=1.0))
            self.parameters.add(Parameter("width", 0.7, "float", minimum=0.0, maximum=1.0))
        self.stereospread = 23
        self.scalegain = 0.2

    def process(self, data):
        d = _Dev(data)
        if d.C > 2:
            raise ValueError("AlgorithmicReverb needs mono or stereo audio")
        p = self.parameters
        wet1 = p.wet_mix.value * ((p.width.value / 2) + 0.5)
        wet2 = p.wet_mix.value * ((1 - p.width.value) / 2)
        combs = (C.c_int * 4)(*self.COMB_DELAYS[4:])                         # combs 1..4 never reach the output (see above)
        ap = (C.c_int * 8)(*[v + (self.stereospread if k % 2 else 0) for pair in self.ALLPASS_DELAYS for k, v in enumerate(pair)])
        y = torch.empty(d.n, d.L, 2, dtype=torch.float32, device=d.x.device)
        nbytes = d.lib.mst_fx_algorithmic_reverb_scratch_bytes(d.n, d.L, 4)
        sc = d.scratch((nbytes + 7) // 8)
        d.lib.check(d.lib.mst_fx_algorithmic_reverb(d.x.data_ptr(), y.data_ptr(), d.n, d.L, d.C, combs, 4, ap, 4, self.stereospread,
                                                    float(p.damping.value), float(p.room_size.value), float(self.scalegain), float(wet1),
                                                    float(wet2), float(p.dry_mix.value), sc.data_ptr(), nbytes, d.stream),
                    "mst_fx_algorithmic_reverb")
        return d.out(y)

def _sumsq(d, t):
    out = torch.empty(d.n * SUMSQ_SLOTS, dtype=torch.float64, device=t.device)
    d.lib.check(d.lib.mst_fx_sumsq(t.data_ptr(), d.n, t.shape[1] * t.shape[2], out.data_ptr(), d.stream), "mst_fx_sumsq")
    return out


class _Pending:
    """An array on its way through an AugmentationChain: device batch t [n, L, C], the per-item factor still to be applied to it
    (scale, float64 [n] or None) and sum(t^2) per item when its producer left it behind (sumsq, float64 [n] or None)."""
    __slots__ = ("dev", "t", "scale", "sumsq", "deferred", "ms")

    def __init__(self, dev, t, scale, sumsq, deferred=None, ms=None):
        self.dev, self.t, self.scale, self.sumsq = dev, t, scale, sumsq
        self.ms = ms            # mid / side energies of t left behind by its producer (the compressor), or None
        # deferred = (imager, bal): an rms-normalised MidSideImager step that has NOT run yet - if a Gain is next, the imager's pass
        # applies the rms factor and the gain itself (MstFxFuse tail folding); anything else runs it first (flush)
        self.deferred = deferred

    def flush(self, gain=None):
        """Run the deferred imager step (+ its rms-normalise); with `gain` (a float factor): and that gain, all in one pass."""
        if self.deferred is None:
            return self
        imager, bal = self.deferred
        d = self.dev.rebind(self.t)
        sumsq_x = self.sumsq if self.sumsq is not None else _sumsq(d, self.t)
        if gain is not None:
            y, _ = imager._run(d, self.scale, False, bal=bal, in_sumsq=sumsq_x, post_gain=gain, in_ms=self.ms)
            return _Pending(d, y, None, None)
        y, sumsq_y = imager._run(d, self.scale, True, bal=bal, in_ms=self.ms)
        scale = torch.empty(d.n, dtype=torch.float64, device=y.device)
        d.lib.check(d.lib.mst_fx_rms_pending(self.scale.data_ptr() if self.scale is not None else None, sumsq_x.data_ptr(),
                                             self.t.shape[1] * self.t.shape[2], sumsq_y.data_ptr(), y.shape[1] * y.shape[2],
                                             scale.data_ptr(), d.n, d.stream), "mst_fx_rms_pending")
        return _Pending(d, y, scale, sumsq_y)

    @staticmethod
    def wrap(x):
        d = _Dev(x)
        return _Pending(d, d.x, None, None)

    def materialize(self):
        """The true batch [n, L, C] on the device."""
        if self.deferred is not None:
            return self.flush().materialize()
        if self.scale is None:
            return self.t
        d = self.dev.rebind(self.t)
        y = torch.empty_like(self.t)
        d.lib.check(d.lib.mst_fx_scale_items(self.t.data_ptr(), y.data_ptr(), d.n, self.t.shape[1] * self.t.shape[2],
                                             self.scale.data_ptr(), d.stream), "mst_fx_scale_items")
        return y

    def result(self):
        """In the caller's container type and rank."""
        return self.dev.out(self.materialize())


class AugmentationChain:
    """Apply (processor, probability, rms_normalize) entries in order to every array of a list; optional shuffle
    and parallel dry/wet mix - the reference's chain semantics (:156-192)."""

    def __init__(self, fxs=None, shuffle=False, parallel=False, parallel_weight_factor=None, randomize_param_value=True):
        self.fxs = fxs if fxs is not None else []
        self.shuffle, self.parallel = shuffle, parallel
        self.parallel_weight_factor = parallel_weight_factor
        self.randomize_param_value = randomize_param_value

    def apply_processor(self, x, processor, rms_normalize):
        """x: an array / tensor, or a _Pending left by the previous processor of a fused chain.  Plain use (one processor,
        array in -> array out) is the reference's apply_processor (:115-148)."""
        if processor.block_size is not None:
            raise NotImplementedError("block-wise processors are not on the gfx950 path")
        if not isinstance(x, _Pending):
            y = processor.process(x)
            if rms_normalize:
                y = rms_normalize_(x, y)
            return y
        if x.deferred is not None:          # an imager + rms step waiting for its successor: a plain Gain folds into its pass
            if isinstance(processor, Gain) and not rms_normalize:
                return x.flush(gain=processor.factor())
            x = x.flush()
        d = x.dev
        if isinstance(processor, MidSideImager) and rms_normalize and processor.fusable(d.rebind(x.t)):
            return _Pending(d, x.t, x.scale, x.sumsq, deferred=(processor, float(processor.parameters.bal.value)), ms=x.ms)
        if hasattr(processor, "fusable") and processor.fusable(d):
            # the pending rms factor of the previous step is folded into this processor's loads; its output leaves sum(y^2) behind
            d.rebind(x.t)
            d.last_in_sumsq = None
            d.last_ms = None          # only what THIS processor's run leaves behind may travel on (a run that never reaches d.fuse() leaves nothing)
            if rms_normalize and x.sumsq is None and isinstance(processor, Equaliser):
                y, sumsq_y = processor._run(d, x.scale, True, want_in_sumsq=True)      # the equaliser leaves sum(x^2) of its input behind too
            else:
                y, sumsq_y = processor._run(d, x.scale, rms_normalize)
            ms = getattr(d, "last_ms", None)
            if not rms_normalize:
                return _Pending(d, y, None, sumsq_y, ms=ms)
            sumsq_x = x.sumsq if x.sumsq is not None else (d.last_in_sumsq if d.last_in_sumsq is not None else _sumsq(d, x.t))
            if sumsq_y is None:
                sumsq_y = _sumsq(d, y)
            scale = torch.empty(d.n, dtype=torch.float64, device=y.device)
            d.lib.check(d.lib.mst_fx_rms_pending(x.scale.data_ptr() if x.scale is not None else None, sumsq_x.data_ptr(),
                                                 x.t.shape[1] * x.t.shape[2], sumsq_y.data_ptr(), y.shape[1] * y.shape[2],
                                                 scale.data_ptr(), d.n, d.stream), "mst_fx_rms_pending")
            return _Pending(d, y, scale, sumsq_y, ms=ms)
        xm = x.materialize()
        y = processor.process(xm)
        if rms_normalize:
            y = rms_normalize_(xm, y)
        d.rebind(y if y.dim() == 3 else y[None])
        return _Pending(d, d.x, None, None)

    def apply_same_processor(self, x_list, processor, rms_normalize):
        return [self.apply_processor(x, processor, rms_normalize) for x in x_list]

    def __call__(self, x_list):
        if self.shuffle:
            random.shuffle(self.fxs)
        # inside the chain every array travels as a _Pending: a device batch, the rms factor still to be applied to it and the
        # sum of squares its producer left behind - a step then costs no extra pass over the audio (see MstFxFuse)
        y_list = [_Pending.wrap(x) for x in x_list]
        for fx, p, rms in self.fxs:
            if np.random.rand() < p:
                if isinstance(fx, Processor):
                    if self.randomize_param_value:
                        fx.randomize()
                    else:
                        fx.update(None)
                    y_list = self.apply_same_processor(y_list, fx, rms)
                else:
                    y_list = [_Pending.wrap(y) for y in fx([y.result() for y in y_list])]
        y_list = [y.result() for y in y_list]
        if self.parallel:
            w = self.parallel_weight_factor if self.parallel_weight_factor else np.random.rand() / 2.0
            y_list = [w * x + (1 - w) * y for x, y in zip(x_list, y_list)]
        return y_list

    def __repr__(self):
        return f"AugmentationChain(fxs={self.fxs!r}, shuffle={self.shuffle!r})"
