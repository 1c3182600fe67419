"""ORACLE (test infrastructure, NOT the product): CPU restatement of the reference networks.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (music_mixing_style_transfer_amd.*) never does; it fails loudly without the
HIP library.

A *functional* restatement in plain torch-CPU fp32 of the reference's forward passes, operating
directly on reference-format state dicts (same key names), so it is independent both of the
reference's module classes and of the product's module classes:

  fxencoder_forward  <- networks/architectures.py:65-70 (FXencoder.forward), :36-45 (block build)
                        networks/network_utils.py:116-119 (Res_ConvBlock.forward: conv2(conv1(x)+x))
                        networks/network_utils.py:28-34,47-51,74-83 (Conv1d_layer: ReflectionPad1d(l,r)
                        with l=pad//2, r=pad-l, Conv1d(stride), BatchNorm1d(eval), ReLU)
  tcn_forward        <- networks/architectures.py:135-147 (TCNModel.forward, clamp(-1,1))
                        :222-234 (TCNBlock.forward: leaky(bn(conv)) -> FiLM -> += res(x_in))
                        :199-207 (zero padding ((k-1)*d)//2, bias=False), :216-220 (grouped 1x1 res)
                        networks/network_utils.py:180-182 (FiLM: split r,b ; r*x+b)
  tcn_receptive_field<- networks/architectures.py:149-155

Parity status: PINNED against the imported reference (tests/golden/make_golden.py runs the real
reference modules in the build container and commits input/output vectors; tests/test_oracle_golden.py
checks this file against them).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # torch.nn.BatchNorm1d default (reference passes none)
LEAKY_SLOPE = 0.01     # torch.nn.LeakyReLU default (architectures.py:215)


def _bn_eval(x, sd, p):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        training=False, eps=BN_EPS)


def same_pad(kernel, dilation=1):
    """(left, right) of the reference's "SAME" padding: total (k-1)*d, left = total//2
    (network_utils.py:30-34).  Even kernels pad asymmetrically, e.g. k=10 -> (4, 5)."""
    total = (kernel - 1) * dilation
    return total // 2, total - total // 2


def conv1d_layer(x, sd, p, kernel, stride=1, dilation=1):
    """Conv1d_layer in 'conv' mode with norm='batch', activation='relu'."""
    l, r = same_pad(kernel, dilation)
    x = F.pad(x, (l, r), mode="reflect")
    y = F.conv1d(x, sd[p + "conv1d.weight"], sd.get(p + "conv1d.bias"), stride=stride, dilation=dilation)
    y = _bn_eval(y, sd, p + "batch_norm.")
    return F.relu(y)


def fxencoder_blocks(x, sd, cfg, collect=None):
    n = len(cfg["kernels"])
    for i in range(n):
        k, s, d = cfg["kernels"][i], cfg["strides"][i], cfg["dilation"][i]
        c1 = conv1d_layer(x, sd, f"encoder.{i}.conv1.conv1d.", k, 1, d) + x
        x = conv1d_layer(c1, sd, f"encoder.{i}.conv2.conv1d.", k, s, d)
        if collect is not None:
            collect.append(x)
    return x


@torch.no_grad()
def fxencoder_forward(sd, cfg, x):
    """x float32 [B,2,L] -> float32 [B, channels[-1]] (global average over time)."""
    y = fxencoder_blocks(x, sd, cfg)
    return y.mean(dim=-1)


def tcn_dilation(n, dilation_growth=2, stack_size=15):
    return dilation_growth ** (n % stack_size)


def film_factors(sd, n, cond):
    """(r, b) each [Bc, C] for block n from cond [Bc, cond_dim]."""
    f = F.linear(cond, sd[f"blocks.{n}.film.film_fc.weight"], sd[f"blocks.{n}.film.film_fc.bias"])
    c = f.shape[1] // 2
    return f[:, :c], f[:, c:]


def tcn_block(x, sd, n, cond, kernel_size, dilation):
    p = f"blocks.{n}."
    w = sd[p + "conv1.weight"]
    pad = ((kernel_size - 1) * dilation) // 2
    y = F.conv1d(x, w, None, padding=pad, dilation=dilation)
    y = F.leaky_relu(_bn_eval(y, sd, p + "bn."), LEAKY_SLOPE)
    r, b = film_factors(sd, n, cond)
    y = r.unsqueeze(-1) * y + b.unsqueeze(-1)
    res = F.conv1d(x, sd[p + "res.weight"], None, groups=x.shape[1])
    return y + res


@torch.no_grad()
def tcn_forward(sd, x, cond, nblocks=14, kernel_size=15, dilation_growth=2, stack_size=15, collect=None):
    """x float32 [B,cin,L]; cond float32 [1|B, cond_dim] or a list with one such tensor per block
    (the reference's 'SeFa' branch, architectures.py:139-140) -> float32 [B,noutputs,L] in [-1,1]."""
    for n in range(nblocks):
        c = cond[n] if isinstance(cond, (list, tuple)) else cond
        x = tcn_block(x, sd, n, c, kernel_size, tcn_dilation(n, dilation_growth, stack_size))
        if collect is not None:
            collect.append(x)
    y = F.conv1d(x, sd["output.weight"], sd["output.bias"])
    return torch.clamp(y, -1.0, 1.0)


def tcn_receptive_field(nblocks=14, kernel_size=15, dilation_growth=2, stack_size=15):
    rf = kernel_size
    for n in range(1, nblocks):
        rf += (kernel_size - 1) * tcn_dilation(n, dilation_growth, stack_size)
    return rf


@torch.no_grad()
def style_transfer_segments(enc_sd, enc_cfg, tcn_sd, ref_segments, in_segments, tcn_kwargs=None):
    """Hot loops 1+2 of inference/style_transfer.py:144-162 on already-segmented tensors:
    embedding = mean over all reference segments, then the converter on every input segment."""
    emb = fxencoder_forward(enc_sd, enc_cfg, ref_segments)
    emb_avg = emb.reshape(-1, emb.shape[-1]).mean(dim=0)
    out = tcn_forward(tcn_sd, in_segments, emb_avg.unsqueeze(0), **(tcn_kwargs or {}))
    return emb, emb_avg, out
