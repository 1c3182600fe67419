"""Numerical gate for a transform-domain TCN block kernel: emulates, in float64 with explicit operand roundings, the
arithmetic a gfx950 kernel would do - per phase sequence N-point overlap-save: real DFT as a GEMM, one complex 128x128 GEMM per bin,
inverse DFT as a GEMM - and reports the max-abs deviation of the TCN output from the fp32 oracle, next to the direct forms.

    python tools/proto/fft_tcn_error.py [L] [B]

Operand formats: 'b' = one bf16 value, 'x' = split hi + lo (three MFMAs per product; the lo*lo term dropped), 'f' = exact fp32.
A configuration is (activation storage, forward DFT operands, bin GEMM operands, inverse DFT operands)."""
import sys
import os
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from music_mixing_style_transfer_amd.utils import synth
from oracle import networks_ref as R

torch.set_grad_enabled(False)
D = torch.float64


def bf16(x):
    return x.to(torch.float32).to(torch.bfloat16).to(D)


def split(x):
    hi = bf16(x)
    lo = bf16(x.to(torch.float32).to(D) - hi)
    return hi, lo


def mm(a, b, fa, fb):
    """a @ b with operand formats fa, fb in {'b','x','f'}; fp32 result (accumulation error of the fp32 accumulator not modelled)."""
    a32, b32 = a.to(torch.float32).to(D), b.to(torch.float32).to(D)
    if fa == 'f' and fb == 'f':
        r = a32 @ b32
    elif fa == 'b' and fb == 'b':
        r = bf16(a) @ bf16(b)
    else:
        ah, al = split(a) if fa == 'x' else (bf16(a), None)
        bh, bl = split(b) if fb == 'x' else (bf16(b), None)
        r = ah @ bh
        if al is not None:
            r = r + al @ bh
        if bl is not None:
            r = r + ah @ bl
    return r.to(torch.float32).to(D)


def dft_mats(N):
    n = torch.arange(N, dtype=D)
    k = torch.arange(N // 2 + 1, dtype=D)
    ang = 2 * np.pi * k[:, None] * n[None, :] / N
    Fr, Fi = torch.cos(ang), -torch.sin(ang)              # X[k] = sum x[n] (cos - i sin)
    s = 1.0 / np.sqrt(N)
    return Fr * s, Fi * s                                  # orthonormal-ish scaling on the forward side


def block_fft(x, w, d, N, V, fmt):
    """x [B, C, L] float64 (already storage-rounded), w [Co, Ci, 15] BN-folded.  Returns conv output [B, Co, L] (fp32-valued)."""
    f_fwd, f_bin, f_inv = fmt
    B, C, L = x.shape
    K = w.shape[2]
    M = (L + d - 1) // d
    xp = torch.zeros(B, C, M * d, dtype=D)
    xp[:, :, :L] = x
    seqs = xp.reshape(B, C, M, d).permute(0, 3, 1, 2).reshape(B * d, C, M)          # phase sequences
    nblk = (M + V - 1) // V
    tot = (nblk - 1) * V + N
    pad = torch.zeros(B * d, C, tot, dtype=D)
    pad[:, :, 7:7 + M] = seqs
    idx = (torch.arange(nblk)[:, None] * V + torch.arange(N)[None, :])
    u = pad[:, :, idx]                                                               # [S, C, nblk, N]
    Fr, Fi = dft_mats(N)
    Fm = torch.cat([Fr, Fi[1:N // 2]], 0)                                            # [N, N] real DFT matrix (N real outputs)
    U = mm(u.reshape(-1, N), Fm.t().contiguous(), f_fwd, f_fwd).reshape(B * d, C, nblk, N)
    Ur = U[..., :N // 2 + 1]
    Ui = torch.zeros_like(Ur)
    Ui[..., 1:N // 2] = U[..., N // 2 + 1:]
    # weights: conj(DFT(w zero padded)) * sqrt(N)-compensation folded into the inverse
    wz = torch.zeros(w.shape[0], w.shape[1], N, dtype=D)
    wz[:, :, :K] = w.to(D)
    Wf = torch.fft.rfft(wz, dim=-1)
    Wr, Wi = Wf.real, -Wf.imag                                                       # conj -> correlation
    Yr = torch.zeros(B * d, w.shape[0], nblk, N // 2 + 1, dtype=D)
    Yi = torch.zeros_like(Yr)
    for kbin in range(N // 2 + 1):
        ur = Ur[..., kbin].permute(1, 0, 2).reshape(C, -1)                           # [Ci, S*nblk]
        ui = Ui[..., kbin].permute(1, 0, 2).reshape(C, -1)
        # [Yr | Yi] = Wr [ur | ui] + Wi [-ui | ur]   (K-concatenated in the kernel: one accumulator)
        a = torch.cat([Wr[:, :, kbin], Wi[:, :, kbin]], 1)                           # [Co, 2Ci]
        bR = torch.cat([ur, -ui], 0)
        bI = torch.cat([ui, ur], 0)
        yr = mm(a, bR, f_bin, f_bin)
        yi = mm(a, bI, f_bin, f_bin)
        Yr[..., kbin] = yr.reshape(w.shape[0], B * d, nblk).permute(1, 0, 2)
        Yi[..., kbin] = yi.reshape(w.shape[0], B * d, nblk).permute(1, 0, 2)
    # inverse real DFT for outputs m = 0..V-1:  y[m] = (1/N) sum_k c_k (Yr cos(2 pi k m / N) - Yi sin(...)),  c = 1 for k = 0, N/2 else 2; forward was scaled 1/sqrt(N)
    m = torch.arange(V, dtype=D)
    k = torch.arange(N // 2 + 1, dtype=D)
    ang = 2 * np.pi * m[:, None] * k[None, :] / N
    c = torch.full((N // 2 + 1,), 2.0, dtype=D)
    c[0] = c[-1] = 1.0
    Gr = torch.cos(ang) * c / np.sqrt(N)
    Gi = -torch.sin(ang) * c / np.sqrt(N)
    G = torch.cat([Gr, Gi[:, 1:N // 2]], 1)                                          # [V, N]
    Yc = torch.cat([Yr, Yi[..., 1:N // 2]], -1)                                      # [S, Co, nblk, N]
    y = mm(Yc.reshape(-1, N), G.t().contiguous(), f_inv, f_inv).reshape(B * d, w.shape[0], nblk * V)[:, :, :M]
    y = y.reshape(B, d, w.shape[0], M).permute(0, 2, 3, 1).reshape(B, w.shape[0], M * d)[:, :, :L]
    return y


def block_direct(x, w, d, f):
    B, C, L = x.shape
    xp = torch.nn.functional.pad(x, (7 * d, 7 * d))
    cols = torch.stack([xp[:, :, j * d:j * d + L] for j in range(15)], 2)            # [B, C, 15, L]
    a = w.to(D).reshape(w.shape[0], -1)                                              # [Co, C*15]
    out = []
    for b in range(B):
        out.append(mm(a, cols[b].reshape(-1, L), f, f))
    return torch.stack(out, 0)


def run(sd, x, cond, act_fmt, conv, nblocks=14):
    """conv(xs, wfold, d) -> conv output; act_fmt 'b' (bf16 activations in HBM) or 'f'."""
    h = x.to(D)
    for n in range(nblocks):
        p = f"blocks.{n}."
        w = sd[p + "conv1.weight"].to(D)
        scale = sd[p + "bn.weight"].to(D) / torch.sqrt(sd[p + "bn.running_var"].to(D) + 1e-5)
        shift = sd[p + "bn.bias"].to(D) - sd[p + "bn.running_mean"].to(D) * scale
        wf = (w * scale[:, None, None]).to(torch.float32).to(D)
        d = R.tcn_dilation(n)
        if n == 0:
            y = torch.nn.functional.conv1d(h, wf, None, padding=7)                   # block 0: its own (hi/lo) kernel, taken as exact
        else:
            y = conv(h, wf, d)
        y = y + shift[None, :, None]
        y = torch.maximum(y, 0.01 * y)
        r, bb = R.film_factors(sd, n, cond)
        y = r.to(D).unsqueeze(-1) * y + bb.to(D).unsqueeze(-1)
        res = sd[p + "res.weight"].to(D).reshape(-1)
        if n == 0:
            y = y + res[:, None] * h[:, torch.arange(128) // 64, :]
        else:
            y = y + res[None, :, None] * h
        h = bf16(y) if act_fmt == 'b' else y.to(torch.float32).to(D)
    out = torch.nn.functional.conv1d(h, sd["output.weight"].to(D), sd["output.bias"].to(D))
    return out.clamp(-1, 1)


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sd = synth.tcn_state_dict(seed=0)
    x = synth.synth_audio((B, 2, L), seed=6)
    cond = synth.synth_audio((1, 2048), seed=9, amp=0.5)
    ref = R.tcn_forward(sd, x, cond).to(D)
    exact = run(sd, x, cond, 'f', lambda h, w, d: block_direct(h, w, d, 'f'))
    print(f"L={L} B={B}  fp32 oracle vs float64 restatement: {float((exact - ref).abs().max()):.2e}")
    cases = [("direct bf16 (the product's bf16 mode)", 'b', lambda h, w, d: block_direct(h, w, d, 'b')),
             ("direct bf16x3", 'f', lambda h, w, d: block_direct(h, w, d, 'x'))]
    for N, V in ((32, 16), (32, 18), (64, 48), (64, 50)):
        for act, fmt in (('f', 'xxx'), ('b', 'bbb'), ('b', 'xbx'), ('f', 'xbx'), ('b', 'xxx')):
            cases.append((f"fft N={N} V={V} act={act} fwd/bin/inv={fmt}", act, lambda h, w, d, N=N, V=V, fmt=fmt: block_fft(h, w, d, N, V, fmt)))
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None      # substrings of the case names to run
    for name, act, conv in cases:
        if only and not any(o in name for o in only):
            continue
        y = run(sd, x, cond, act, conv)
        print(f"{name:55s} max|y - oracle| = {float((y - ref).abs().max()):.3e}   rms = {float(((y - ref) ** 2).mean().sqrt()):.3e}", flush=True)


if __name__ == "__main__":
    main()
