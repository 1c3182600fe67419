#!/usr/bin/env python
"""Builds tools/_ab/probe.so (untracked): the product's csrc/ with s_memtime phase stamps patched into ONE wave of ONE workgroup of
  * tcn_block_bf16_duo_kernel (matrix wave 0, the d = 64 block): probe slots 0 .. 7  (tools/probe_tcn_phases.py --kernel duo)
  * tcn_block_bf16x3_kernel<2, 4> (wave 0, the d = 64 block):     probe slots 8 .. 15 (tools/probe_tcn_phases.py --kernel x3)
and an extra entry point mst_probe_read().  The product sources carry no probe code.   python tools/build_probe.py"""
import os
import subprocess

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(R, "music_mixing_style_transfer_amd", "csrc")
DST = os.path.join(R, "tools", "_ab", "probe_src")


def patch(s, old, new, begin=None, end=None):
    """Replace `old` (unique in s, or unique between the markers begin / end) by `new`."""
    lo = s.index(begin) if begin else 0
    hi = s.index(end, lo) if end else len(s)
    region = s[lo:hi]
    assert region.count(old) == 1, "anchor not found (or not unique): " + old[:70]
    return s[:lo] + region.replace(old, new) + s[hi:]


X3_BEGIN = "void tcn_block_bf16x3_kernel(TcnBlockArgs a) {"
X3_END = "// bf16x3, large dilations"


def main():
    os.makedirs(DST, exist_ok=True)
    for f in os.listdir(SRC):
        if f.endswith((".h", ".hip")) or f == "Makefile":
            open(os.path.join(DST, f), "w").write(open(os.path.join(SRC, f)).read())
    p = os.path.join(DST, "tcn_kernels.h")
    s = open(p).read()
    dev = os.path.join(DST, "mst_dev.h")
    open(dev, "a").write("\n__device__ long long mst_tcn_probe[32];      // probe build only\n")
    s = patch(s, "struct TcnBlockArgs {", """#define MST_PROBE(k) do { if (probe_on) { const long long now_ = mst_clock(); mst_tcn_probe[k] += now_ - probe_t; probe_t = now_; } } while (0)
struct TcnBlockArgs {""")
    # ---- duo kernel, matrix wave 0
    s = patch(s, """    const unsigned tiles_item = (unsigned)a.tiles_phase * (unsigned)a.tiles_step;
    for (;;) {
        const int b = tb, m0 = tm0, phi0 = tphi0;""", """    const unsigned tiles_item = (unsigned)a.tiles_phase * (unsigned)a.tiles_step;
    const bool probe_on = blockIdx.x == 8 && wv == 0 && lane == 0 && a.d == 64;
    long long probe_t = mst_clock();
    for (;;) {
        MST_PROBE(0);            // loop bookkeeping
        const int b = tb, m0 = tm0, phi0 = tphi0;""")
    s = patch(s, """            tcn_reuse_class<P, 15 / NCLS, true, NUMAX>(acc, A0, A1, ring, sm, wst, aoff, NCLS - 1, 0, l16, g);
        } else {""", """            MST_PROBE(1);        // acc init + ring preload + classes 0 .. NCLS - 2
            tcn_reuse_class<P, 15 / NCLS, true, NUMAX>(acc, A0, A1, ring, sm, wst, aoff, NCLS - 1, 0, l16, g);
            MST_PROBE(2);        // the last class
        } else {""")
    s = patch(s, """        read_xin(0);
        mst_dma_wait_barrier<63>();            // (1)""", """        read_xin(0);
        MST_PROBE(3);            // residual reads issued
        mst_dma_wait_barrier<63>();            // (1)""")
    s = patch(s, """        float hs0[NC], hs1[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) hs0[q] = hs1[q] = 0.0f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (m) {
                read_xin(1);""", """        MST_PROBE(4);            // barrier 1
        float hs0[NC], hs1[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) hs0[q] = hs1[q] = 0.0f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (m) {
                read_xin(1);""")
    s = patch(s, """        } else {
            mst_dma_wait_barrier<63>();          // (2) the transposed output tile is complete: the loader waves store it
        }""", """        } else {
            MST_PROBE(5);        // epilogue arithmetic + LDS writes
            mst_dma_wait_barrier<63>();          // (2) the transposed output tile is complete: the loader waves store it
            MST_PROBE(6);        // barrier 2
            if (probe_on) mst_tcn_probe[7] += 1;
        }""")
    # ---- split-bf16 one-tile kernel
    s = patch(s, """    const float *xb = (const float *)a.x + (size_t)b * a.Lp * 128;
    float *yb = (float *)a.y + (size_t)b * a.Lp * 128;

    // ---- stage the rows: a thread owns one 16-byte slot (8 channels) of rows prow, prow + 16, ...; fp32 in, (hi, lo) bf16 out""", """    const float *xb = (const float *)a.x + (size_t)b * a.Lp * 128;
    float *yb = (float *)a.y + (size_t)b * a.Lp * 128;
    const bool probe_on = (blockIdx.x & 63) == 8 && tid == 0 && a.d == 64 && P == 2 && NQ == 4;
    long long probe_t = mst_clock();

    // ---- stage the rows: a thread owns one 16-byte slot (8 channels) of rows prow, prow + 16, ...; fp32 in, (hi, lo) bf16 out""", X3_BEGIN, X3_END)
    s = patch(s, """    __syncthreads();

    // v_mfma_f32_16x16x32_bf16 like the bf16 kernel (same operand traffic per FLOP as the 32 x 32 x 16 form, more throughput under the
    // power limit): two row tiles of 16 channels x NC column tiles of 16 times per wave""", """    MST_PROBE(8);                // staging: loads, split, LDS writes
    __syncthreads();
    MST_PROBE(9);                // barrier behind the staging

    // v_mfma_f32_16x16x32_bf16 like the bf16 kernel (same operand traffic per FLOP as the 32 x 32 x 16 form, more throughput under the
    // power limit): two row tiles of 16 channels x NC column tiles of 16 times per wave""", X3_BEGIN, X3_END)
    s = patch(s, """    const float *frow = a.film + (a.film_rows > 1 ? (size_t)b * 256 : 0);
    __syncthreads();                       // every wave is done reading the input tiles
    float *st = (float *)smem;""", """    const float *frow = a.film + (a.film_rows > 1 ? (size_t)b * 256 : 0);
    MST_PROBE(10);               // main loop
    __syncthreads();                       // every wave is done reading the input tiles
    MST_PROBE(11);               // barrier behind the main loop
    float *st = (float *)smem;""", X3_BEGIN, X3_END)
    s = patch(s, """            *(f32x4 *)(st + o * 128 + (((co0 >> 2) ^ (o & 31)) << 2)) = z;
        }
    }
    __syncthreads();
    {
        const int s4 = tid & 31;                                   // this thread's 4 channels, the same in every pass""", """            *(f32x4 *)(st + o * 128 + (((co0 >> 2) ^ (o & 31)) << 2)) = z;
        }
    }
    MST_PROBE(12);               // LeakyReLU / FiLM + transposed LDS writes
    __syncthreads();
    MST_PROBE(13);               // barrier
    {
        const int s4 = tid & 31;                                   // this thread's 4 channels, the same in every pass""", X3_BEGIN, X3_END)
    s = patch(s, """                *(f32x4 *)(yb + t * 128 + 4 * s4) = out;
            }
        }
    }
}
""", """                *(f32x4 *)(yb + t * 128 + 4 * s4) = out;
            }
        }
        MST_PROBE(14);           // rows: residual from global memory + store
        if (probe_on) mst_tcn_probe[15] += 1;
    }
}
""", X3_BEGIN, X3_END)
    open(p, "w").write(s)
    # ---- encoder conv kernel (enc_conv_nlc_kernel<4>): wave 0 of workgroup 100 of the launches with Cin = Cout = 2048; slots 16 .. 23
    p = os.path.join(DST, "enc_kernels.h")
    s = open(p).read()
    s = patch(s, "struct EncNlcArgs {", """#define MST_EPROBE(k) do { if (eprobe_on) { const long long now_ = mst_clock(); mst_tcn_probe[k] += now_ - eprobe_t; eprobe_t = now_; } } while (0)
struct EncNlcArgs {""")
    NB, NE = "void enc_conv_nlc_kernel(EncNlcArgs a) {", "// The 128-channel x 128-column tile with its four waves 2 x 2"
    s = patch(s, """    if (kc_lo < kc_hi) {
        stab_entry(kc_lo);
        fetch(kc_lo);
    }
    for (int kc = kc_lo; kc < kc_hi; ++kc) {
        if (kc > kc_lo) __syncthreads();""", """    const bool eprobe_on = blockIdx.x == 100 && tid == 0 && a.Cin == 2048 && a.Cout == 2048 && MW == 4;
    long long eprobe_t = mst_clock();
    if (kc_lo < kc_hi) {
        stab_entry(kc_lo);
        fetch(kc_lo);
    }
    MST_EPROBE(16);              // prologue: descriptors, first fetch issued
    for (int kc = kc_lo; kc < kc_hi; ++kc) {
        if (kc > kc_lo) __syncthreads();
        MST_EPROBE(17);          // barrier: every wave is done with the previous chunk's tile""", NB, NE)
    s = patch(s, """        __syncthreads();
        if (kc + 1 < kc_hi) fetch(kc + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = 128 * ni + 32 * q + ln;""", """        MST_EPROBE(18);          // wait for this chunk's loads + LDS writes of the B tile
        __syncthreads();
        MST_EPROBE(19);          // barrier: the tile is complete
        if (kc + 1 < kc_hi) fetch(kc + 1);
        MST_EPROBE(20);          // next chunk's loads issued
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = 128 * ni + 32 * q + ln;""", NB, NE)
    s = patch(s, """                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[ks], bv, acc[q], 0, 0, 0);
            }
        }
    }
""", """                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[ks], bv, acc[q], 0, 0, 0);
            }
        }
        MST_EPROBE(21);          // 16 MFMAs + 16 ds_read_b128
        if (eprobe_on) mst_tcn_probe[23] += 1;
    }
""", NB, NE)
    open(p, "w").write(s)
    hp = os.path.join(DST, "mst_host.h")
    open(hp, "w").write(open(hp).read().replace('#include "../../include/mst_hip.h"', '#include "../../../include/mst_hip.h"'))
    p = os.path.join(DST, "mst_tcn.hip")          # the translation unit that holds the TCN kernels (and so the probe array)
    s = open(p).read()
    s += '''
extern "C" int mst_probe_read(long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mst_tcn_probe), 32 * sizeof(long long)) != hipSuccess) return -3;
    if (reset) { long long z[32] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(mst_tcn_probe), z, sizeof(z)) != hipSuccess) return -3; }
    return 0;
}
'''
    open(p, "w").write(s)
    mk = os.path.join(DST, "Makefile")
    text = open(mk).read().replace("../../include/mst_hip.h", "../../../include/mst_hip.h")
    open(mk, "w").write(text)
    subprocess.run(["make", "-j4", "-C", DST], check=True)
    subprocess.run(["cp", os.path.join(DST, "libmst_hip.so"), os.path.join(R, "tools", "_ab", "probe.so")], check=True)
    print("built tools/_ab/probe.so")


if __name__ == "__main__":
    main()
