#!/usr/bin/env python
"""BASELINE.json configs[3]: the FX-manipulator chain (biquad EQ -> rms-norm -> compressor -> rms-norm -> mid/side
imager -> rms-norm -> gain) on a batch of 64 stereo segments of 131072 samples on one MI355X, with fixed
parameters (oracle/fx_ref.py CONFIG4).  Prints one JSON line: segments/s, algorithmic GB/s, per-processor ms,
max deviation from the oracle on a checked sample, and the oracle's C restatement timed on one host core."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    from music_mixing_style_transfer_amd.mixing_manipulator import Compressor, Equaliser, Gain, MidSideImager, rms_normalize_
    from oracle import fx_ref as F
    n, L = 64, 131072
    g = torch.Generator().manual_seed(0)
    x = (0.1 * torch.randn(n, L, 2, generator=g)).clamp_(-1, 1).cuda()
    eq = Equaliser(2, 44100)
    for band, (gg, fc, q) in F.CONFIG4["eq"].items():
        getattr(eq.parameters, band + "_gain").value = gg
    comp, im, gn = Compressor(44100), MidSideImager(), Gain()
    im.parameters.bal.value = F.CONFIG4["imager_bal"]
    gn.parameters.gain.value = F.CONFIG4["gain_db"]

    def chain(x, times=None):
        def timed(name, fn):
            if times is None:
                return fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = fn()
            b.record()
            torch.cuda.synchronize()
            times[name] = times.get(name, 0.0) + a.elapsed_time(b)
            return r
        y = timed("equaliser", lambda: eq.process(x))
        y = timed("rms_norm", lambda: rms_normalize_(x, y))
        z = timed("compressor", lambda: comp.process(y))
        z = timed("rms_norm", lambda: rms_normalize_(y, z))
        w = timed("imager", lambda: im.process(z))
        w = timed("rms_norm", lambda: rms_normalize_(z, w))
        return timed("gain", lambda: gn.process(w))

    # the timed form: the product's AugmentationChain (pending rms factors folded into the next processor's loads, energy sums
    # left behind by the producers); `chain()` above is the same sequence processor by processor, used for the per-processor times
    from music_mixing_style_transfer_amd.mixing_manipulator import AugmentationChain
    for k, v in F.CONFIG4["comp"].items():
        getattr(comp.parameters, k).value = v
    fused = AugmentationChain(fxs=[(eq, 1.0, True), (comp, 1.0, True), (im, 1.0, True), (gn, 1.0, False)], randomize_param_value=False)
    out = fused([x])[0]
    torch.cuda.synchronize()
    if "--chain-only" in sys.argv:           # counter passes (tools/gpu_fx_pmc.sh): N chains and nothing else after the warm-up chain
        for _ in range(int(sys.argv[sys.argv.index("--chain-only") + 1])):
            out = fused([x])[0]
        torch.cuda.synchronize()
        return
    steps = 5
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fused([x])[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    times = {}
    chain(x, times)
    # parity on two items against the oracle chain (compressor via the oracle's C restatement for speed)
    import subprocess
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle")], check=True, capture_output=True)
    lib = C.CDLL(os.path.join(REPO, "oracle", "libfx_ref.so"))
    fp = C.POINTER(C.c_float)

    def c_comp(xx, threshold, attack_time, release_time, ratio, sample_rate):
        xx = np.ascontiguousarray(xx, dtype=np.float32)
        yy = np.empty_like(xx)
        lib.ref_compressor(xx.ctypes.data_as(fp), yy.ctypes.data_as(fp), C.c_long(xx.shape[0]), xx.shape[1], C.c_double(threshold),
                           C.c_double(attack_time), C.c_double(release_time), C.c_double(ratio), C.c_double(0.0), C.c_double(sample_rate))
        return yy
    dev = 0.0
    t1 = time.perf_counter()
    for i in (0, 17):
        ref = F.fx_chain(x[i].cpu().numpy(), compressor_fn=c_comp)
        dev = max(dev, float(np.abs(out[i].cpu().numpy() - ref).max()))
    cpu_dt = (time.perf_counter() - t1) / 2
    alg_bytes = 144 * L * n           # unfused per-processor read+write bytes of the chain
    # row f-3: convolution reverb, 1.5 s stereo impulse response, same batch
    from music_mixing_style_transfer_amd.mixing_manipulator import ConvolutionalReverb
    from music_mixing_style_transfer_amd.utils import synth
    Lh = 66150
    hl = (synth.synth_audio((Lh, 2), seed=9).numpy().astype(np.float64) * np.exp(-np.arange(Lh) / 12000.0)[:, None] * 0.05).astype(np.float32)
    hl[441] = (0.8, 0.7)
    rv = ConvolutionalReverb([[{"impulse_response": (lambda: hl)}]], 44100)
    rv.update()
    yr = rv.process(x)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(steps):
        yr = rv.process(x)
    torch.cuda.synchronize()
    rv_dt = (time.perf_counter() - t2) / steps
    t3 = time.perf_counter()
    ref = F.conv_reverb(x[3].cpu().numpy(), hl)
    rv_cpu = time.perf_counter() - t3
    rv_dev = float(np.abs(yr[3].cpu().numpy() - ref).max() / np.abs(ref).max())
    print(json.dumps({"metric": "FX chain segments/sec (EQ+compressor+imager+gain, rms-normalised)", "value": n / dt,
                      "unit": "segments/s", "n_items": n, "segment": [L, 2], "ms_per_chain": dt * 1e3,
                      "algorithmic_GBps": alg_bytes / dt / 1e9, "per_processor_ms": times,
                      "max_abs_dev_vs_oracle": dev,
                      "conv_reverb": {"segments_per_s": n / rv_dt, "ms_per_batch": rv_dt * 1e3, "ir_samples": Lh, "n_fft": 262144,
                                      "rel_dev_vs_oracle": rv_dev, "oracle_segments_per_s_1core": 1.0 / rv_cpu},
                      "cpu_baseline": {"value": 1.0 / cpu_dt, "unit": "segments/s", "cores": 1, "kind": "port",
                                       "sample": "2 segments, oracle/fx_ref.py chain (scipy lfilter EQ + oracle/fx_ref.c compressor)"}}))


if __name__ == "__main__":
    main()
