"""Deterministic synthetic weights / audio (no checkpoints or datasets are reachable offline).

Values come from a counter-based splitmix64 hash of (key, element index), so the very same
tensors are produced in the build container (where golden vectors are generated) and on the
GPU box, independent of torch / numpy RNG implementations.

State-dict key names and shapes follow the reference modules
(networks/architectures.py:26-70 FXencoder, :76-133 TCNModel, :177-220 TCNBlock,
networks/network_utils.py:15-89 Conv1d_layer, :156-160 FiLM).  Weight scales are chosen so that
activations stay O(1) through the 24 encoder convs and the 14 TCN blocks (plain nn.init gives
|embedding| ~ 3e-3).
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hashed_uniform(key, n, lo=-1.0, hi=1.0, seed=0):
    """n float64 values in [lo, hi), a pure function of (key, seed, index)."""
    base = np.uint64(((zlib.crc32(key.encode()) & 0xFFFFFFFF) * 0x100000001B3 + int(seed)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (base << np.uint64(20))
        bits = _splitmix64(_splitmix64(idx) ^ base)
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return lo + (hi - lo) * u


def _t(key, shape, lo, hi, seed, dtype=torch.float32):
    n = int(np.prod(shape)) if len(shape) else 1
    return torch.from_numpy(hashed_uniform(key, n, lo, hi, seed).reshape(shape)).to(dtype)


def _bn(sd, prefix, c, seed):
    sd[prefix + "weight"] = _t(prefix + "weight", (c,), 0.8, 1.2, seed)
    sd[prefix + "bias"] = _t(prefix + "bias", (c,), -0.1, 0.1, seed)
    sd[prefix + "running_mean"] = _t(prefix + "running_mean", (c,), -0.1, 0.1, seed)
    sd[prefix + "running_var"] = _t(prefix + "running_var", (c,), 0.8, 1.2, seed)
    sd[prefix + "num_batches_tracked"] = torch.tensor(1000, dtype=torch.int64)


def fxencoder_state_dict(cfg, seed=0, gain=1.0):
    """Synthetic FXencoder weights; cfg = configs.yaml Effects_Encoder entry (channels WITHOUT the
    leading 2 the reference inserts at architectures.py:30, or with it - both accepted)."""
    ch = list(cfg["channels"])
    if len(ch) == len(cfg["kernels"]):
        ch = [2] + ch
    sd = OrderedDict()
    for i, k in enumerate(cfg["kernels"]):
        for name, cin, cout in (("conv1", ch[i], ch[i]), ("conv2", ch[i], ch[i + 1])):
            p = f"encoder.{i}.{name}.conv1d."
            a = gain * (3.0 / (cin * k)) ** 0.5
            sd[p + "conv1d.weight"] = _t(p + "conv1d.weight", (cout, cin, k), -a, a, seed)
            sd[p + "conv1d.bias"] = _t(p + "conv1d.bias", (cout,), -0.05, 0.05, seed)
            _bn(sd, p + "batch_norm.", cout, seed)
    return sd


def tcn_state_dict(nblocks=14, ninputs=2, noutputs=2, channel_width=128, kernel_size=15,
                   cond_dim=2048, seed=0, gain=0.85, film_gain=0.2, out_gain=0.3):
    sd = OrderedDict()
    for n in range(nblocks):
        cin = ninputs if n == 0 else channel_width
        cout = channel_width
        p = f"blocks.{n}."
        a = gain * (3.0 / (cin * kernel_size)) ** 0.5
        sd[p + "conv1.weight"] = _t(p + "conv1.weight", (cout, cin, kernel_size), -a, a, seed)
        af = film_gain * (3.0 / cond_dim) ** 0.5
        sd[p + "film.film_fc.weight"] = _t(p + "film.film_fc.weight", (2 * cout, cond_dim), -af, af, seed)
        fb = _t(p + "film.film_fc.bias", (2 * cout,), -0.1, 0.1, seed)
        fb[:cout] += 1.0  # FiLM scale half (network_utils.py:181) centred on 1
        sd[p + "film.film_fc.bias"] = fb
        _bn(sd, p + "bn.", cout, seed)
        sd[p + "res.weight"] = _t(p + "res.weight", (cout, 1, 1), 0.4, 0.9, seed)
    a = out_gain * (3.0 / channel_width) ** 0.5
    sd["output.weight"] = _t("output.weight", (noutputs, channel_width, 1), -a, a, seed)
    sd["output.bias"] = _t("output.bias", (noutputs,), -0.05, 0.05, seed)
    return sd


def synth_audio(shape, seed=0, amp=1.0, key="audio"):
    """Seeded U(-amp, amp) float32 tensor."""
    n = int(np.prod(shape))
    return torch.from_numpy(hashed_uniform(key, n, -amp, amp, seed).reshape(shape).astype(np.float32))


def synth_music(n_channels, length, seed=0, sr=44100, key="music"):
    """Band-limited-ish stereo test signal: a few sines + decaying noise bursts, |x| <= 0.5
    (SURVEY.md 8d config 3).  float32 [n_channels, length]."""
    t = np.arange(length, dtype=np.float64) / sr
    out = np.zeros((n_channels, length))
    par = hashed_uniform(key + "/par", 64, 0.0, 1.0, seed)
    for c in range(n_channels):
        for h in range(6):
            f = 55.0 * 2 ** (par[8 * c + h] * 6.0)
            out[c] += (0.3 / (h + 1)) * np.sin(2 * np.pi * f * t + 6.28 * par[32 + 8 * c + h])
        noise = hashed_uniform(key + f"/n{c}", length, -1.0, 1.0, seed)
        env = 0.5 * (1.0 + np.sin(2 * np.pi * 2.0 * t + c)) ** 2 * 0.25
        out[c] += 0.2 * env * noise
    out *= 0.5 / max(1e-9, np.abs(out).max())
    return torch.from_numpy(out.astype(np.float32))


def save_reference_format_checkpoint(path, state_dict):
    """Write a checkpoint in the layout the reference's save_checkpoint produces
    (modules/training_utils.py:13-29) and reload_weights consumes (inference/style_transfer.py:94-108):
    {"model": {"module." + key: tensor}, "optimizer": .., "scheduler": .., "epoch": ..}."""
    model = OrderedDict(("module." + k, v) for k, v in state_dict.items())
    torch.save({"model": model, "optimizer": {}, "scheduler": {}, "epoch": 0}, path)
