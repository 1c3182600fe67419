// libmst_hip.so, FXencoder part of the C ABI (mst_enc_*, mst_global_avgpool, mst_embedding_mean) and the implicit-GEMM conv launch the
// generic TCN configuration shares: BatchNorm folding, weight packing and launches of csrc/enc_kernels.h.  See include/mst_hip.h.
#include "mst_host.h"
#include "enc_kernels.h"

// fp32 image for enc_conv_kernel: wpk[cot][kc][kr][m] = W[cot*MT+m][kc*16+kr] * scale[co], zero padded
int pack_conv_f32(MstEncConv &c, const float *w, const std::vector<float> &scale) {
    const int MT = 32 * c.mw, K = c.cin * c.ksz;
    const int co_tiles = (c.cout + MT - 1) / MT;
    std::vector<float> wp((size_t)co_tiles * c.nchunks * 16 * MT, 0.0f);
    host_parallel_for(co_tiles, [&](int cot) {
        for (int kc = 0; kc < c.nchunks; ++kc)
            for (int kr = 0; kr < 16; ++kr) {
                const int k = kc * 16 + kr;
                if (k >= K) continue;
                for (int m = 0; m < MT; ++m) {
                    const int co = cot * MT + m;
                    if (co < c.cout) wp[(((size_t)cot * c.nchunks + kc) * 16 + kr) * MT + m] = w[(size_t)co * K + k] * scale[co];
                }
            }
    });
    std::vector<int> kt((size_t)c.nchunks32 * 32 * 2);
    for (int k = 0; k < c.nchunks32 * 32; ++k) {
        kt[2 * k] = k < K ? k / c.ksz : -1;
        kt[2 * k + 1] = k < K ? (k % c.ksz) * c.dil - c.pad_l : 0;
    }
    int rc;
    if ((rc = upload(&c.wpk, wp))) return rc;
    if ((rc = upload(&c.ktab, kt))) return rc;
    return MST_OK;
}
void conv_geometry(MstEncConv &c, int cin, int cout, int ksz, int stride, int dil, int pad_l, int pad_r) {
    c.cin = cin;
    c.cout = cout;
    c.ksz = ksz;
    c.stride = stride;
    c.dil = dil;
    c.pad_l = pad_l;
    c.pad_r = pad_r;
    c.nchunks = (cin * ksz + 15) / 16;
    c.nchunks32 = (cin * ksz + 31) / 32;
    c.mw = cout <= 32 ? 1 : (cout <= 64 ? 2 : 4);
}

// enc_conv_kernel's gathers: 32-bit byte offsets on a descriptor that starts at the tile's first batch item (24-bit channel x length multiply)
int conv_buf32(int mw, int B, int cin, long Lin, long Lout) {
    const long NT = 128 * (4 / mw), span_items = std::min<long>(B, NT / std::max<long>(1, Lout) + 2);
    return (Lin < (1 << 24) && cin < (1 << 24) && (double)span_items * cin * Lin * 4.0 < 2147483647.0) ? 1 : 0;
}

// generic configuration: every block is one launch of the fp32 implicit-GEMM conv kernel (NCL activations, zero
// padding) with the TCN epilogue; the output head is the same kernel with k = 1 and the clamp epilogue
int tcn_launch_generic(const MstEncConv &c, const float *x, float *y, int B, int L, int epi, const float *film, int film_rows,
                       const float *res, int res_div, void *stream) {
    EncConvArgs a;
    a.x = x;
    a.y = y;
    a.wpk = c.wpk;
    a.shift = c.shift;
    a.ktab = c.ktab;
    a.wpk16 = nullptr;
    a.nchunks32 = c.nchunks32;
    a.B = B;
    a.Cin = c.cin;
    a.Lin = L;
    a.Cout = c.cout;
    a.Lout = L;
    a.stride = 1;
    a.nchunks = c.nchunks;
    a.residual = 0;
    a.Ntot = (long)B * L;
    a.pad_zero = 1;
    a.epi = epi;
    a.film = film;
    a.res = res;
    a.film_rows = film_rows;
    a.res_div = res_div;
    a.buf32 = conv_buf32(c.mw, B, c.cin, L, L);
    const int MT = 32 * c.mw, NT = 128 * (4 / c.mw);
    const dim3 grid((unsigned)((a.Ntot + NT - 1) / NT), (unsigned)((c.cout + MT - 1) / MT));
    switch (c.mw) {
        case 1: MST_LAUNCH((enc_conv_kernel<1>), grid, dim3(256), stream, a); break;
        case 2: MST_LAUNCH((enc_conv_kernel<2>), grid, dim3(256), stream, a); break;
        default: MST_LAUNCH((enc_conv_kernel<4>), grid, dim3(256), stream, a); break;
    }
    MST_CHECK_LAUNCH("enc_conv_kernel (generic TCN)");
    return MST_OK;
}

// =================================================================================================
// FXencoder
// =================================================================================================
struct MstEnc {
    MstEncDesc d;
    std::vector<MstEncConv> conv;   // 2 per block
    int schedule = 1;               // bit 5: the 128-channel layers on the four-wave im2col kernel instead of the raw-rows kernel with loader waves; bit 4: blocks 1 / 2 (bf16 mode) as two launches each instead of the fused kernel; bit 3: the stereo block as two direct-kernel launches instead of the fused kernel (the bit-identical reference form); bit 1: 2 x 2 wave tiling of the 128-channel conv kernel (measured slower: off); bit 0: weight-major workgroup order for the weight-heavy layers (mst_enc_set_schedule)
    void *zeros = nullptr;          // 256 bytes of zeros: what the channel-minor conv kernel fetches for rows / k-slots outside the problem
    long rows_min_tiles = 512;      // bf16 mode: layers with at least this many tiles keep their input rows resident in LDS (mst_enc_set_tuning)
};

extern "C" int mst_enc_create(const MstEncDesc *desc, MstEnc **out) {
    if (!desc || !out) return fail(MST_ERR_ARG, "mst_enc_create: null argument");
    if (desc->nblocks < 1 || desc->nblocks > MST_MAX_BLOCKS) return fail(MST_ERR_ARG, "mst_enc_create: nblocks out of range");
    if (!(desc->act_slope >= 0.0f && desc->act_slope <= 1.0f)) return fail(MST_ERR_ARG, "mst_enc_create: act_slope outside [0, 1]");
    for (int i = 0; i < desc->nblocks; ++i)
        if (desc->kernels[i] < 1 || desc->strides[i] < 1 || desc->dilations[i] < 1 || desc->channels[i] < 1 ||
            desc->channels[i + 1] < 1)
            return fail(MST_ERR_ARG, "mst_enc_create: bad layer description");
    MstEnc *e = new MstEnc();
    if (hipMalloc(&e->zeros, 256) != hipSuccess || hipMemset(e->zeros, 0, 256) != hipSuccess) {
        (void)hipFree(e->zeros);
        delete e;
        return fail(MST_ERR_HIP, "mst_enc_create: hipMalloc failed");
    }
    e->d = *desc;
    e->conv.resize(2 * desc->nblocks);
    for (int i = 0; i < desc->nblocks; ++i)
        for (int which = 0; which < 2; ++which) {
            MstEncConv &c = e->conv[2 * i + which];
            c.cin = desc->channels[i];
            c.cout = which ? desc->channels[i + 1] : desc->channels[i];
            c.ksz = desc->kernels[i];
            c.stride = which ? desc->strides[i] : 1;
            c.dil = desc->dilations[i];
            const int pad = desc->valid_padding ? 0 : (c.ksz - 1) * c.dil;   // "SAME": total (k-1)*d, left = total//2 (network_utils.py:30-34)
            c.pad_l = pad / 2;
            c.pad_r = pad - c.pad_l;
            c.nchunks = (c.cin * c.ksz + 15) / 16;
            c.nchunks32 = (c.cin * c.ksz + 31) / 32;
            c.mw = c.cout <= 32 ? 1 : (c.cout <= 64 ? 2 : 4);
            c.slope = desc->act_slope;
        }
    *out = e;
    return MST_OK;
}

extern "C" int mst_enc_destroy(MstEnc *e) {
    if (!e) return MST_OK;
    for (auto &c : e->conv) {
        (void)hipFree(c.wpk);
        (void)hipFree(c.wpk16);
        (void)hipFree(c.w_direct);
        (void)hipFree(c.w_frag);
        (void)hipFree(c.w_frag16);
        (void)hipFree(c.w_taps);
        (void)hipFree(c.wpk_nlc);
        (void)hipFree(c.wpk_nlc_lo);
        (void)hipFree(c.stab);
        (void)hipFree(c.shift);
        (void)hipFree(c.ktab);
    }
    (void)hipFree(e->zeros);
    delete e;
    return MST_OK;
}

extern "C" int mst_enc_load_conv(MstEnc *e, int block, int which, const float *w, const float *bias,
                                 const float *bn_weight, const float *bn_bias, const float *bn_mean,
                                 const float *bn_var, float bn_eps, void *) {
    if (!e || !w || !bn_weight || !bn_bias || !bn_mean || !bn_var) return fail(MST_ERR_ARG, "mst_enc_load_conv: null argument");
    if (block < 0 || block >= e->d.nblocks || which < 0 || which > 1) return fail(MST_ERR_ARG, "mst_enc_load_conv: index out of range");
    MstEncConv &c = e->conv[2 * block + which];
    std::vector<float> scale, shift;
    bn_fold(bn_weight, bn_bias, bn_mean, bn_var, bn_eps, c.cout, scale, shift);
    const int MT = 32 * c.mw, K = c.cin * c.ksz;
    const int co_tiles = (c.cout + MT - 1) / MT;
    std::vector<float> wp((size_t)co_tiles * c.nchunks * 16 * MT, 0.0f);
    for (int cot = 0; cot < co_tiles; ++cot)
        for (int kc = 0; kc < c.nchunks; ++kc)
            for (int kr = 0; kr < 16; ++kr) {
                const int k = kc * 16 + kr;
                if (k >= K) continue;
                for (int m = 0; m < MT; ++m) {
                    const int co = cot * MT + m;
                    if (co < c.cout) wp[(((size_t)cot * c.nchunks + kc) * 16 + kr) * MT + m] = w[(size_t)co * K + k] * scale[co];
                }
            }
    std::vector<float> sh((size_t)co_tiles * MT, 0.0f);
    for (int co = 0; co < c.cout; ++co) sh[co] = shift[co] + (bias ? bias[co] * scale[co] : 0.0f);
    // bf16 A fragments of v_mfma_f32_32x32x16_bf16: [cot][kc32][ks][mi][lane][e]
    std::vector<__bf16> wp16((size_t)co_tiles * c.nchunks32 * 2 * c.mw * 64 * 8);
    host_parallel_for(co_tiles, [&](int cot) {
        for (int kc = 0; kc < c.nchunks32; ++kc)
            for (int ks = 0; ks < 2; ++ks)
                for (int mi = 0; mi < c.mw; ++mi)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const int co = cot * MT + 32 * mi + (l & 31);
                            const int k = kc * 32 + ks * 16 + 8 * (l >> 5) + e;
                            const float v = (co < c.cout && k < K) ? w[(size_t)co * K + k] * scale[co] : 0.0f;
                            wp16[((((((size_t)cot * c.nchunks32 + kc) * 2 + ks) * c.mw + mi) * 64 + l) * 8) + e] = (__bf16)v;
                        }
    });
    std::vector<int> kt((size_t)c.nchunks32 * 32 * 2);
    for (int k = 0; k < c.nchunks32 * 32; ++k) {
        kt[2 * k] = k < K ? k / c.ksz : -1;
        kt[2 * k + 1] = k < K ? (k % c.ksz) * c.dil - c.pad_l : 0;
    }
    int rc;
    if (c.cin < 8) {
        std::vector<float> wd((size_t)c.cout * K);
        for (int co = 0; co < c.cout; ++co)
            for (int k = 0; k < K; ++k) wd[(size_t)co * K + k] = w[(size_t)co * K + k] * scale[co];
        if ((rc = upload(&c.w_direct, wd))) return rc;
        if (c.cin == 2 && c.ksz == ENC_STEREO_K && c.dil == 1 && c.cout == 2 && c.stride == 1) {
            std::vector<float> fr((size_t)ENC_STEREO_KS0 * 64);
            enc_stereo_pack_a0(wd.data(), fr.data());
            if ((rc = upload(&c.w_frag, fr))) return rc;
        } else if (c.cin == 2 && c.ksz == ENC_STEREO_K && c.dil == 1 && c.cout == 16 && c.stride == 4) {
            std::vector<float> fr((size_t)ENC_STEREO_KS1 * 64);
            enc_stereo_pack_a1(wd.data(), fr.data());
            if ((rc = upload(&c.w_frag, fr))) return rc;
        }
    } else if (c.cin % 8 == 0) {
        // NLC pipeline: contraction index k = j*Cin + ci; fragments [cot][kc64][ks 0..3][mi][lane][e]
        c.nchunks64 = (K + 63) / 64;
        std::vector<__bf16> wn((size_t)co_tiles * c.nchunks64 * 4 * c.mw * 64 * 8), wl(wn.size());
        host_parallel_for(co_tiles, [&](int cot) {
            for (int kc = 0; kc < c.nchunks64; ++kc)
                for (int ks = 0; ks < 4; ++ks)
                    for (int mi = 0; mi < c.mw; ++mi)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                const int co = cot * MT + 32 * mi + (l & 31);
                                const int k = kc * 64 + ks * 16 + 8 * (l >> 5) + e;
                                float v = 0.0f;
                                if (co < c.cout && k < K) v = w[((size_t)co * c.cin + (k % c.cin)) * c.ksz + k / c.cin] * scale[co];
                                const size_t at = ((((((size_t)cot * c.nchunks64 + kc) * 4 + ks) * c.mw + mi) * 64 + l) * 8) + e;
                                wn[at] = (__bf16)v;
                                wl[at] = (__bf16)(v - (float)wn[at]);
                            }
        });
        std::vector<int> st((size_t)c.nchunks64 * 8 * 2);
        for (int sidx = 0; sidx < c.nchunks64 * 8; ++sidx) {
            const int k0 = sidx * 8;
            st[2 * sidx] = k0 < K ? (k0 / c.cin) * c.dil - c.pad_l : 0;
            st[2 * sidx + 1] = k0 < K ? k0 % c.cin : -1;
        }
        if ((rc = upload(&c.wpk_nlc, wn))) return rc;
        if ((rc = upload(&c.wpk_nlc_lo, wl))) return rc;
        if (c.mw == 4 && c.dil == 1 && c.cin % 64 == 0 && (c.ksz == 5 || c.ksz == 10)) {          // the raw-rows kernel's fragment image
            std::vector<__bf16> img((size_t)((c.cout + 127) / 128) * c.ksz * (c.cin / 64) * 2 * 8 * 64 * 8);
            host_parallel_for((c.cout + 127) / 128, [&](int ct) { enc_taps_pack(w, scale.data(), c.cout, c.cin, c.ksz, ct, img.data()); });
            if ((rc = upload(&c.w_taps, img))) return rc;
        }
        if (c.dil == 1 && ((c.cin == 16 && c.ksz == 25) || (c.cin == 32 && c.ksz == 15)) && (c.cout == c.cin || c.cout == 2 * c.cin)) {
            // blocks 1 / 2 of the default encoder: the fused kernel's fragments, one 16-row tile after the other
            std::vector<float> wf((size_t)c.cout * K);
            for (int co = 0; co < c.cout; ++co)
                for (int k = 0; k < K; ++k) wf[(size_t)co * K + k] = w[(size_t)co * K + k] * scale[co];
            const size_t per_tile = (size_t)enc_block1_ks(c.cin, c.ksz) * 64 * 8;
            std::vector<__bf16> fr((size_t)(c.cout / 16) * per_tile);
            for (int m = 0; m < c.cout / 16; ++m) enc_block1_pack(wf.data(), 16 * m, c.cin, c.ksz, fr.data() + (size_t)m * per_tile);
            if ((rc = upload(&c.w_frag16, fr))) return rc;
        }
        if ((rc = upload(&c.stab, st))) return rc;
    }
    if ((rc = upload(&c.wpk, wp))) return rc;
    if ((rc = upload(&c.wpk16, wp16))) return rc;
    if ((rc = upload(&c.shift, sh))) return rc;
    if ((rc = upload(&c.ktab, kt))) return rc;
    c.loaded = true;
    return MST_OK;
}

extern "C" int mst_global_avgpool(const float *x, float *y, long rows, int L, void *stream) {
    if (!x || !y || rows < 1 || L < 1) return fail(MST_ERR_ARG, "mst_global_avgpool: bad argument");
    MST_LAUNCH(enc_avgpool_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), stream, x, y, rows, L);
    MST_CHECK_LAUNCH("enc_avgpool_kernel");
    return MST_OK;
}

extern "C" int mst_enc_zero_stuff(const float *x, float *y, long rows, long L, int stride, long pad_left, long Lu, void *stream) {
    if (!x || !y || rows < 1 || L < 1 || stride < 1 || pad_left < 0 || Lu < pad_left + (L - 1) * stride + 1)
        return fail(MST_ERR_ARG, "mst_enc_zero_stuff: bad argument");
    const long total = rows * Lu;
    if ((total + 255) / 256 > 0x7fffffffL) return fail(MST_ERR_ARG, "mst_enc_zero_stuff: too large");
    MST_LAUNCH(enc_zero_stuff_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), stream, x, y, rows, L, stride, pad_left, Lu);
    MST_CHECK_LAUNCH("enc_zero_stuff_kernel");
    return MST_OK;
}

extern "C" int mst_enc_set_schedule(MstEnc *e, int flags) {
    if (!e || flags < 0 || flags > 63) return fail(MST_ERR_ARG, "mst_enc_set_schedule: flags 0..63");
    e->schedule = flags;
    return MST_OK;
}

extern "C" int mst_enc_set_tuning(MstEnc *e, long rows_min_tiles) {
    if (!e) return fail(MST_ERR_ARG, "mst_enc_set_tuning: null handle");
    e->rows_min_tiles = rows_min_tiles;
    return MST_OK;
}

namespace {
int conv_out_length(const MstEncConv &c, int L) {      // reflection-padded length, then the strided "valid" conv
    const int span = (c.ksz - 1) * c.dil;
    const int Lp = L + c.pad_l + c.pad_r;
    return Lp > span ? (Lp - span - 1) / c.stride + 1 : 0;
}
}  // namespace

extern "C" int mst_enc_block_length(const MstEnc *e, int block, int L) {
    if (!e || block < 0 || block >= e->d.nblocks) return -1;
    for (int i = 0; i <= block; ++i) L = conv_out_length(e->conv[2 * i + 1], conv_out_length(e->conv[2 * i], L));
    return L;
}

extern "C" int mst_enc_conv_length(const MstEnc *e, int block, int which, int L) {
    if (!e || block < 0 || block >= e->d.nblocks || which < 0 || which > 1) return -1;
    return conv_out_length(e->conv[2 * block + which], L);
}

namespace {

size_t enc_buf_floats(const MstEnc *e, int B, int L) {
    size_t mx = 0;
    int len = L;
    for (int i = 0; i < e->d.nblocks; ++i) {
        mx = std::max(mx, (size_t)B * e->d.channels[i] * len);
        len = (len - 1) / e->d.strides[i] + 1;
        mx = std::max(mx, (size_t)B * e->d.channels[i + 1] * len);
    }
    return mx;
}

int enc_splitk_f32(long tiles, int nchunks) {        // slices of the fp32 NCL kernel: aim at >= 1024 workgroups, >= 8 k-chunks per slice
    if (tiles >= 512) return 1;
    int S = (int)((1024 + tiles - 1) / tiles);
    if (S > 16) S = 16;
    if (S > nchunks / 8) S = nchunks / 8;
    return S < 1 ? 1 : S;
}

int enc_launch(const MstEncConv &c, const float *x, float *y, int B, int Lin, int Lout, int residual, int precision,
               void *stream, float *scratch = nullptr, int schedule = 0) {
    if (Lin <= c.pad_l || Lin <= c.pad_r)
        return fail(MST_ERR_ARG, "mst_enc_forward: reflection padding needs the input to be longer than the padding");
    EncConvArgs a;
    a.x = x;
    a.y = y;
    a.wpk = c.wpk;
    a.shift = c.shift;
    a.ktab = c.ktab;
    a.wpk16 = c.wpk16;
    a.nchunks32 = c.nchunks32;
    a.B = B;
    a.Cin = c.cin;
    a.Lin = Lin;
    a.Cout = c.cout;
    a.Lout = Lout;
    a.stride = c.stride;
    a.nchunks = c.nchunks;
    a.residual = residual;
    a.Ntot = (long)B * Lout;
    a.pad_zero = 0;
    a.epi = 0;
    a.film = nullptr;
    a.res = nullptr;
    a.film_rows = 1;
    a.res_div = 1;
    a.slope = c.slope;
    a.buf32 = (schedule & 4) ? 0 : conv_buf32(c.mw, B, c.cin, Lin, Lout);          // bit 2: the 64-bit gather path (a test hook)
    const int MT = 32 * c.mw, NT = 128 * (4 / c.mw);
    dim3 grid((unsigned)((a.Ntot + NT - 1) / NT), (unsigned)((c.cout + MT - 1) / MT));
    int S = 1;
    if (precision != MST_PREC_BF16 && scratch) {
        S = enc_splitk_f32((long)grid.x * grid.y, c.nchunks);
        if (S > 1) {
            a.part = scratch;
            grid.z = (unsigned)S;
        }
    }
    if (precision == MST_PREC_BF16) {
        switch (c.mw) {
            case 1: MST_LAUNCH((enc_conv_bf16_kernel<1>), grid, dim3(256), stream, a); break;
            case 2: MST_LAUNCH((enc_conv_bf16_kernel<2>), grid, dim3(256), stream, a); break;
            default: MST_LAUNCH((enc_conv_bf16_kernel<4>), grid, dim3(256), stream, a); break;
        }
    } else {
        switch (c.mw) {
            case 1: MST_LAUNCH((enc_conv_kernel<1>), grid, dim3(256), stream, a); break;
            case 2: MST_LAUNCH((enc_conv_kernel<2>), grid, dim3(256), stream, a); break;
            default: MST_LAUNCH((enc_conv_kernel<4>), grid, dim3(256), stream, a); break;
        }
    }
    MST_CHECK_LAUNCH("enc_conv_kernel");
    if (S > 1) {
        const long total = a.Ntot * c.cout;
        MST_LAUNCH(enc_splitk_finalize_ncl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), stream, (const float *)scratch, S, a.Ntot,
                   c.cout, Lout, (const float *)c.shift, residual ? x : (const float *)nullptr, y, c.slope);
        MST_CHECK_LAUNCH("enc_splitk_finalize_ncl_kernel");
    }
    return MST_OK;
}

// the channel-minor bf16 pipeline needs a stereo-like first block for the direct kernel and channel counts that are
// multiples of 8 afterwards (true for configs.yaml); otherwise bf16 mode uses the NCL gather kernel
bool enc_nlc_eligible(const MstEnc *e) {
    const MstEncDesc &d = e->d;
    if (d.channels[0] > 4 || d.channels[1] > 32 || d.channels[1] % 8 != 0 || d.kernels[0] > 64) return false;
    if (255 * d.strides[0] + (d.kernels[0] - 1) * d.dilations[0] + 1 > 256 * 8 + 64) return false;
    for (int i = 1; i <= d.nblocks; ++i)
        if (d.channels[i] % 8 != 0) return false;
    return true;
}

// the raw-rows kernel (enc_conv_taps_kernel: 256-column tiles, one workgroup per CU): which layers it serves and its k-slices (over 64-channel
// blocks): aim at one workgroup per CU of the chip - a second round of a few workgroups doubles the launch
bool enc_taps_fits(const MstEncConv &c, int Lout) {
    return c.mw == 4 && c.dil == 1 && c.cin % 64 == 0 && Lout % 32 == 0 && (c.ksz == 5 || c.ksz == 10) && (c.stride == 1 || c.stride == 2) && c.w_taps;
}
int enc_splitk_taps(long tiles256, int nblk) {
    const long cus = mst_num_cus();
    int S = (int)((cus + tiles256 / 2) / std::max(1L, tiles256));
    if (S > 8) S = 8;
    if (S > nblk) S = nblk;
    return S < 1 ? 1 : S;
}
int enc_splitk(long tiles, int nchunks) {
    if (tiles >= 512) return 1;
    int S = (int)((768 + tiles - 1) / tiles);
    if (S > 8) S = 8;
    if (S > nchunks / 4) S = nchunks / 4;
    return S < 1 ? 1 : S;
}

size_t enc_scratch_floats(const MstEnc *e, int B, int L) {
    size_t mx = 0;
    int len = L;
    for (int i = 0; i < e->d.nblocks; ++i) {
        const int lout = (len - 1) / e->d.strides[i] + 1;
        for (int which = 0; which < 2; ++which) {
            const MstEncConv &c = e->conv[2 * i + which];
            const long ntot = (long)B * (which ? lout : len);
            const int MT = 32 * c.mw, NT = 128 * (4 / c.mw);
            const long tiles = ((ntot + NT - 1) / NT) * ((c.cout + MT - 1) / MT);
            const int nch = (c.cin * c.ksz + 63) / 64;
            const int S = enc_splitk(tiles, nch);
            if (S > 1) mx = std::max(mx, (size_t)S * ntot * c.cout);
            if (enc_taps_fits(c, which ? lout : len)) {
                const int St = enc_splitk_taps(((ntot + 255) / 256) * ((c.cout + MT - 1) / MT), c.cin / 64);
                if (St > 1) mx = std::max(mx, (size_t)St * ntot * c.cout);
            }
            const int Sf = enc_splitk_f32(tiles, c.nchunks);          // exact-fp32 mode slices
            if (Sf > 1) mx = std::max(mx, (size_t)Sf * ntot * c.cout);
        }
        len = lout;
    }
    return mx;
}

int enc_launch_direct(const MstEncConv &c, const float *x, void *y, bool out_nlc, int B, int Lin, int Lout, int residual,
                      void *stream, void *ylo = nullptr) {
    if (Lin <= c.pad_l || Lin <= c.pad_r)
        return fail(MST_ERR_ARG, "mst_enc_forward: reflection padding needs the input to be longer than the padding");
    EncDirectArgs a;
    a.x = x;
    a.y = y;
    a.ylo = ylo;
    a.w = c.w_direct;
    a.shift = c.shift;
    a.slope = c.slope;
    a.B = B;
    a.Cin = c.cin;
    a.Lin = Lin;
    a.Cout = c.cout;
    a.Lout = Lout;
    a.ksz = c.ksz;
    a.stride = c.stride;
    a.dil = c.dil;
    a.pad_l = c.pad_l;
    a.residual = residual;
    const dim3 grid((unsigned)(B * ((Lout + 255) / 256)));
    // accumulator capacity of the instantiation: next power of two >= Cout (NLC output packs 8 channels per store)
    const int cm = c.cout <= 2 ? 2 : c.cout <= 4 ? 4 : c.cout <= 8 ? 8 : c.cout <= 16 ? 16 : 32;
    if (out_nlc) {
        switch (cm) {
            case 8: MST_LAUNCH((enc_direct_kernel<true, 8>), grid, dim3(256), stream, a); break;
            case 16: MST_LAUNCH((enc_direct_kernel<true, 16>), grid, dim3(256), stream, a); break;
            default: MST_LAUNCH((enc_direct_kernel<true, 32>), grid, dim3(256), stream, a); break;
        }
    } else {
        switch (cm) {
            case 2: MST_LAUNCH((enc_direct_kernel<false, 2>), grid, dim3(256), stream, a); break;
            case 4: MST_LAUNCH((enc_direct_kernel<false, 4>), grid, dim3(256), stream, a); break;
            case 8: MST_LAUNCH((enc_direct_kernel<false, 8>), grid, dim3(256), stream, a); break;
            case 16: MST_LAUNCH((enc_direct_kernel<false, 16>), grid, dim3(256), stream, a); break;
            default: MST_LAUNCH((enc_direct_kernel<false, 32>), grid, dim3(256), stream, a); break;
        }
    }
    MST_CHECK_LAUNCH("enc_direct_kernel");
    return MST_OK;
}

// the default encoder's stereo block (2 -> 2, k = 25 with skip; 2 -> 16, k = 25, stride 4) as one launch
bool enc_stereo_block_fits(const MstEncConv &c0, const MstEncConv &c1, int L) {
    auto same_pad = [](const MstEncConv &c) { return c.ksz == ENC_STEREO_K && c.dil == 1 && c.pad_l == 12 && c.pad_r == 12 && c.cin == 2 && c.w_frag; };
    return same_pad(c0) && same_pad(c1) && c0.cout == 2 && c0.stride == 1 && c1.cout == 16 && c1.stride == 4 && L > 12 && L < (1 << 29);
}
int enc_launch_stereo_block(const MstEncConv &c0, const MstEncConv &c1, const float *x, void *y, void *ylo, int B, int L, int Lout, void *stream) {
    EncStereoArgs a;
    a.x = x;
    a.y = y;
    a.ylo = ylo;
    a.a0 = c0.w_frag;
    a.shift0 = c0.shift;
    a.a1 = c1.w_frag;
    a.shift1 = c1.shift;
    a.B = B;
    a.L = L;
    a.Lout = Lout;
    a.tiles = (Lout + ENC_STEREO_TO - 1) / ENC_STEREO_TO;
    a.slope0 = c0.slope;
    a.slope1 = c1.slope;
    if ((long)B * a.tiles > 0x7fffffffL) return fail(MST_ERR_ARG, "enc_stereo_block_kernel: grid too large");
    MST_LAUNCH(enc_stereo_block_kernel, dim3((unsigned)(B * a.tiles)), dim3(256), stream, a);
    MST_CHECK_LAUNCH("enc_stereo_block_kernel");
    return MST_OK;
}

// blocks 1 / 2 of the default encoder (C -> C, k with skip; C -> 2 C, k, stride S for (C, k, S) = (16, 25, 4), (32, 15, 2)), bf16 mode, as one launch each
int enc_block1_form(const MstEncConv &c0, const MstEncConv &c1, int L) {          // 1 / 2: which instantiation fits, 0: none
    auto same = [&](const MstEncConv &c, int cin, int ksz) {
        return c.cin == cin && c.ksz == ksz && c.dil == 1 && c.pad_l == (ksz - 1) / 2 && c.pad_r == (ksz - 1) / 2 && c.w_frag16;
    };
    if (L <= c0.pad_l || L >= (1 << 25)) return 0;
    if (same(c0, 16, 25) && same(c1, 16, 25) && c0.cout == 16 && c0.stride == 1 && c1.cout == 32 && c1.stride == 4) return 1;
    if (same(c0, 32, 15) && same(c1, 32, 15) && c0.cout == 32 && c0.stride == 1 && c1.cout == 64 && c1.stride == 2) return 2;
    return 0;
}
int enc_launch_block1(int form, const MstEncConv &c0, const MstEncConv &c1, const __bf16 *x, __bf16 *y, int B, int L, int Lout, const void *zeros, void *stream) {
    EncBlock1Args a;
    a.x = x;
    a.y = y;
    a.a0 = c0.w_frag16;
    a.a1 = c1.w_frag16;
    a.shift0 = c0.shift;
    a.shift1 = c1.shift;
    a.B = B;
    a.L = L;
    a.Lout = Lout;
    a.tiles = (Lout + ENC_B1_TO - 1) / ENC_B1_TO;
    a.slope0 = c0.slope;
    a.slope1 = c1.slope;
    a.zeros = zeros;
    if ((long)B * a.tiles > 0x7fffffffL) return fail(MST_ERR_ARG, "enc_block1_fused_kernel: grid too large");
    if (form == 1) MST_LAUNCH((enc_block1_fused_kernel<16, 25, 4, 66>), dim3((unsigned)(B * a.tiles)), dim3(256), stream, a);
    else MST_LAUNCH((enc_block1_fused_kernel<32, 15, 2, 33>), dim3((unsigned)(B * a.tiles)), dim3(256), stream, a);
    MST_CHECK_LAUNCH("enc_block1_fused_kernel");
    return MST_OK;
}

// x3: split mode - x / y point at the high parts' planes, the low parts' planes follow at B * L * C elements
int enc_launch_nlc(const MstEncConv &c, const __bf16 *x, __bf16 *y, float *scratch, int B, int Lin, int Lout, int residual,
                   long rows_min_tiles, void *stream, bool x3 = false, int schedule = 0, const void *zeros = nullptr) {
    if (Lin <= c.pad_l || Lin <= c.pad_r)
        return fail(MST_ERR_ARG, "mst_enc_forward: reflection padding needs the input to be longer than the padding");
    EncNlcArgs a;
    a.x = x;
    a.y = y;
    a.xlo = x3 ? x + (size_t)B * Lin * c.cin : nullptr;
    a.ylo = x3 ? y + (size_t)B * Lout * c.cout : nullptr;
    a.wpk = c.wpk_nlc;
    a.wpk_lo = x3 ? c.wpk_nlc_lo : c.wpk_nlc;
    a.shift = c.shift;
    a.stab = c.stab;
    a.B = B;
    a.Cin = c.cin;
    a.Lin = Lin;
    a.Cout = c.cout;
    a.Lout = Lout;
    a.stride = c.stride;
    a.nchunks = c.nchunks64;
    a.residual = residual;
    a.Ntot = (long)B * Lout;
    const int MT = 32 * c.mw, NT = 128 * (4 / c.mw);
    const long ntiles = (a.Ntot + NT - 1) / NT, cotiles = (c.cout + MT - 1) / MT;
    a.ksz = c.ksz;
    a.pad_l = c.pad_l;
    a.wmajor = 0;
    a.slope = c.slope;
    a.zeros = zeros;
    {   // 32-bit offsets of the im2col kernel's buffer loads: a tile's rows lie within NT batch items of one descriptor; 24-bit row multiply
        const long NTl = 128 * (4 / c.mw), span_items = std::min<long>(B, NTl / std::max(1, Lout) + 2);      // items a tile of NT columns can touch
        if (Lin >= (1 << 24) || c.cin >= (1 << 24) || (double)span_items * Lin * c.cin * 2.0 >= 2147483647.0 ||
            (double)c.nchunks64 * 4.0 * c.mw * 64.0 * 16.0 >= 2147483647.0)
            return fail(MST_ERR_UNSUPPORTED, "mst_enc_forward: activation too long for the channel-minor pipeline (use MST_PREC_F32)");
    }
    if (!zeros) return fail(MST_ERR_ARG, "enc_launch_nlc: no zero page");
    {
        // long early layers: the tile's input rows resident in LDS instead of an im2col slice per k-chunk
        const long tiles_item = (Lout + NT - 1) / NT;
        const long R = (long)(NT - 1) * c.stride + c.ksz, rpp = (R + c.stride - 1) / c.stride;
        const long lds = (long)c.stride * rpp * (c.cin * 2 + 16);
        const bool fits = x3 ? 2 * lds <= 80 * 1024 : lds <= 64 * 1024;
        // measured (round 3, same box, after the im2col kernel's loads were fixed and the rows kernel got its four-step A ring): bf16 mode -
        // rows 52.6 / 23.2 / 22.9 / 32.4 us vs im2col 92.9 / 31.4 / 28.6 / 37.6 us for 16 / 32 / 32 / 64 input channels, equal at 64 -> 128
        // (32.3 / 32.6), im2col ahead at 128 channels (35.7 vs 38.9); split mode - rows ahead wherever it fits (54.3 / 52.7 / 63.5 vs
        // 67.6 / 60.8 / 71.1 us).  rows_min_tiles = 0 forces the rows form wherever it qualifies (tests).
        const bool narrow = rows_min_tiles == 0 || x3 || c.cin <= 64;
        if (rows_min_tiles >= 0 && narrow && c.dil == 1 && c.cin % 16 == 0 && Lout >= NT && fits && (long)B * tiles_item * cotiles >= rows_min_tiles) {
            a.S = 1;
            a.part = nullptr;
            const dim3 grid((unsigned)(B * tiles_item), (unsigned)cotiles);
            if (x3) {
                switch (c.mw) {
                    case 1: MST_LAUNCH((enc_conv_rows_kernel<1, true>), grid, dim3(256), stream, a); break;
                    case 2: MST_LAUNCH((enc_conv_rows_kernel<2, true>), grid, dim3(256), stream, a); break;
                    default: MST_LAUNCH((enc_conv_rows_kernel<4, true>), grid, dim3(256), stream, a); break;
                }
            } else
            switch (c.mw) {
                case 1: MST_LAUNCH((enc_conv_rows_kernel<1>), grid, dim3(256), stream, a); break;
                case 2: MST_LAUNCH((enc_conv_rows_kernel<2>), grid, dim3(256), stream, a); break;
                default: MST_LAUNCH((enc_conv_rows_kernel<4>), grid, dim3(256), stream, a); break;
            }
            MST_CHECK_LAUNCH("enc_conv_rows_kernel");
            return MST_OK;
        }
    }
    if (!x3 && !(schedule & 2) && !(schedule & 32) && enc_taps_fits(c, Lout) && a.Ntot < 0x7fffff00L && (!residual || (Lin == Lout && c.cin == c.cout))) {          // the 128-channel layers on raw input rows with loader waves
        EncTapsArgs t;
        t.x = x;
        t.y = y;
        t.wpk = c.w_taps;
        t.shift = c.shift;
        t.B = B;
        t.Cin = c.cin;
        t.Lin = Lin;
        t.Cout = c.cout;
        t.Lout = Lout;
        t.stride = c.stride;
        t.ksz = c.ksz;
        t.pad_l = c.pad_l;
        t.nchunks = c.nchunks64;
        t.residual = residual;
        t.Ntot = a.Ntot;
        t.slope = c.slope;
        t.zeros = zeros;
        const long nt2 = (a.Ntot + 255) / 256;
        t.S = enc_splitk_taps(nt2 * cotiles, c.cin / 64);
        t.part = t.S > 1 ? scratch : nullptr;
        const dim3 g2((unsigned)nt2, (unsigned)cotiles, (unsigned)t.S);
        if (c.ksz == 5 && c.stride == 1) MST_LAUNCH((enc_conv_taps_kernel<5, 1>), g2, dim3(512), stream, t);
        else if (c.ksz == 5) MST_LAUNCH((enc_conv_taps_kernel<5, 2>), g2, dim3(512), stream, t);
        else if (c.stride == 1) MST_LAUNCH((enc_conv_taps_kernel<10, 1>), g2, dim3(512), stream, t);
        else MST_LAUNCH((enc_conv_taps_kernel<10, 2>), g2, dim3(512), stream, t);
        MST_CHECK_LAUNCH("enc_conv_taps_kernel");
        if (t.S > 1) {
            const long total = a.Ntot * (c.cout / 4);
            MST_LAUNCH(enc_splitk_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), stream, (const float *)scratch,
                       t.S, a.Ntot, c.cout, (const float *)c.shift, residual ? x : (const __bf16 *)nullptr, y,
                       (const __bf16 *)nullptr, (__bf16 *)nullptr, c.slope);
            MST_CHECK_LAUNCH("enc_splitk_finalize_kernel");
        }
        return MST_OK;
    }
    a.S = enc_splitk(ntiles * cotiles, c.nchunks64);
    a.part = a.S > 1 ? scratch : nullptr;
    dim3 grid((unsigned)ntiles, (unsigned)cotiles, (unsigned)a.S);
    // weight-heavy layers (more weight bytes than activation bytes, at least 8 weight slices): weight-major workgroup order (see the kernel)
    if ((schedule & 1) && (long)c.cout * c.cin * c.ksz > a.Ntot * c.cin && cotiles * a.S >= 8) {
        a.wmajor = (int)cotiles;
        grid = dim3((unsigned)(ntiles * cotiles * a.S));
    }
    const bool w22 = c.mw == 4 && (schedule & 2);          // the 128 x 128 tile with its waves 2 x 2 (two MFMAs per LDS read)
    if (x3) {
        switch (c.mw) {
            case 1: MST_LAUNCH((enc_conv_nlc_kernel<1, true>), grid, dim3(256), stream, a); break;
            case 2: MST_LAUNCH((enc_conv_nlc_kernel<2, true>), grid, dim3(256), stream, a); break;
            default:
                if (w22) MST_LAUNCH((enc_conv_nlc22_kernel<true>), grid, dim3(256), stream, a);
                else MST_LAUNCH((enc_conv_nlc_kernel<4, true>), grid, dim3(256), stream, a);
                break;
        }
    } else
    switch (c.mw) {
        case 1: MST_LAUNCH((enc_conv_nlc_kernel<1>), grid, dim3(256), stream, a); break;
        case 2: MST_LAUNCH((enc_conv_nlc_kernel<2>), grid, dim3(256), stream, a); break;
        default:
            if (w22) MST_LAUNCH((enc_conv_nlc22_kernel<false>), grid, dim3(256), stream, a);
            else MST_LAUNCH((enc_conv_nlc_kernel<4>), grid, dim3(256), stream, a);
            break;
    }
    MST_CHECK_LAUNCH("enc_conv_nlc_kernel");
    if (a.S > 1) {
        const long total = a.Ntot * (c.cout / 4);
        MST_LAUNCH(enc_splitk_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), stream, (const float *)scratch,
                   a.S, a.Ntot, c.cout, (const float *)c.shift, residual ? x : (const __bf16 *)nullptr, y,
                   residual ? a.xlo : (const __bf16 *)nullptr, a.ylo, c.slope);
        MST_CHECK_LAUNCH("enc_splitk_finalize_kernel");
    }
    return MST_OK;
}

int enc_run_nlc(MstEnc *e, const float *x, float *emb, float *blk_out, int B, int L, int n_run, void *ws, void *stream, bool x3 = false) {
    const size_t nb = align_up(enc_buf_floats(e, B, L) * sizeof(float), 256);
    unsigned char *base = (unsigned char *)ws;
    void *t1 = base;
    void *o[2] = {base + nb, base + 2 * nb};
    float *scratch = (float *)(base + 3 * nb);
    int len = L, rc, pp = 0;
    const void *cur = x;
    for (int i = 0; i < n_run; ++i) {
        const int lout = (len - 1) / e->d.strides[i] + 1;
        if (i == 0) {
            void *lo_plane = x3 ? (void *)((__bf16 *)o[pp] + (size_t)B * lout * e->conv[1].cout) : nullptr;
            if (!(e->schedule & 8) && enc_stereo_block_fits(e->conv[0], e->conv[1], len) && lout == conv_out_length(e->conv[1], len)) {
                if ((rc = enc_launch_stereo_block(e->conv[0], e->conv[1], (const float *)cur, o[pp], lo_plane, B, len, lout, stream))) return rc;
            } else {
                if ((rc = enc_launch_direct(e->conv[0], (const float *)cur, t1, false, B, len, len, 1, stream))) return rc;
                if ((rc = enc_launch_direct(e->conv[1], (const float *)t1, o[pp], true, B, len, lout, 0, stream, lo_plane))) return rc;
            }
        } else if (!x3 && !(e->schedule & 16) && enc_block1_form(e->conv[2 * i], e->conv[2 * i + 1], len) && lout == conv_out_length(e->conv[2 * i + 1], len)) {
            if ((rc = enc_launch_block1(enc_block1_form(e->conv[2 * i], e->conv[2 * i + 1], len), e->conv[2 * i], e->conv[2 * i + 1], (const __bf16 *)cur,
                                        (__bf16 *)o[pp], B, len, lout, e->zeros, stream))) return rc;
        } else {
            if ((rc = enc_launch_nlc(e->conv[2 * i], (const __bf16 *)cur, (__bf16 *)t1, scratch, B, len, len, 1, e->rows_min_tiles, stream, x3, e->schedule, e->zeros))) return rc;
            if ((rc = enc_launch_nlc(e->conv[2 * i + 1], (const __bf16 *)t1, (__bf16 *)o[pp], scratch, B, len, lout, 0, e->rows_min_tiles, stream, x3, e->schedule, e->zeros))) return rc;
        }
        cur = o[pp];
        pp ^= 1;
        len = lout;
    }
    const int C = e->d.channels[n_run];
    if (blk_out) {
        const size_t total = (size_t)B * len * C;
        MST_LAUNCH(enc_unpack_nlc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), stream, (const __bf16 *)cur,
                   x3 ? (const __bf16 *)cur + total : (const __bf16 *)nullptr, blk_out, B, len, C);
        MST_CHECK_LAUNCH("enc_unpack_nlc_kernel");
    }
    if (emb) {
        MST_LAUNCH(enc_avgpool_nlc_kernel, dim3((unsigned)(((long)B * C + 255) / 256)), dim3(256), stream, (const __bf16 *)cur,
                   x3 ? (const __bf16 *)cur + (size_t)B * len * C : (const __bf16 *)nullptr, emb, B, len, C);
        MST_CHECK_LAUNCH("enc_avgpool_nlc_kernel");
    }
    return MST_OK;
}

int enc_run(MstEnc *e, const float *x, float *emb, float *blk_out, int B, int L, int precision, int n_run, void *ws,
            size_t ws_bytes, void *stream) {
    if (!e || !x || B < 1 || L < 1) return fail(MST_ERR_ARG, "mst_enc_forward: bad argument");
    if (e && e->d.valid_padding)
        return fail(MST_ERR_UNSUPPORTED, "mst_enc_forward: a Res_ConvBlock needs 'SAME' padding (conv1(x) + x); VALID layers run through mst_enc_forward_conv");
    // bf16x3: the channel-minor pipeline in split mode (two bf16 planes per activation, three MFMAs per product); configurations the
    // pipeline does not cover run the exact-fp32 path
    if (precision == MST_PREC_BF16X3 && !enc_nlc_eligible(e)) precision = MST_PREC_F32;
    if (precision != MST_PREC_F32 && precision != MST_PREC_BF16 && precision != MST_PREC_BF16X3) return fail(MST_ERR_ARG, "mst_enc_forward: bad precision");
    for (auto &c : e->conv)
        if (!c.loaded) return fail(MST_ERR_STATE, "mst_enc_forward: conv weights not loaded");
    if (!ws || ws_bytes < mst_enc_workspace_bytes(e, B, L)) return fail(MST_ERR_WORKSPACE, "mst_enc_forward: workspace too small");
    if (precision == MST_PREC_BF16 && enc_nlc_eligible(e)) return enc_run_nlc(e, x, emb, blk_out, B, L, n_run, ws, stream);
    if (precision == MST_PREC_BF16X3) return enc_run_nlc(e, x, emb, blk_out, B, L, n_run, ws, stream, true);
    const size_t nb = align_up(enc_buf_floats(e, B, L) * sizeof(float), 256);
    float *t1 = (float *)ws;
    float *o[2] = {(float *)((unsigned char *)ws + nb), (float *)((unsigned char *)ws + 2 * nb)};
    float *scratch = (float *)((unsigned char *)ws + 3 * nb);
    const float *cur = x;
    int len = L, rc, pp = 0;
    for (int i = 0; i < n_run; ++i) {
        const int lout = (len - 1) / e->d.strides[i] + 1;
        if ((rc = enc_launch(e->conv[2 * i], cur, t1, B, len, len, 1, precision, stream, scratch, e->schedule))) return rc;
        float *dst = (blk_out && i == n_run - 1) ? blk_out : o[pp];
        if ((rc = enc_launch(e->conv[2 * i + 1], t1, dst, B, len, lout, 0, precision, stream, scratch, e->schedule))) return rc;
        cur = dst;
        pp ^= 1;
        len = lout;
    }
    if (emb) {
        const long rows = (long)B * e->d.channels[e->d.nblocks];
        MST_LAUNCH(enc_avgpool_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), stream, cur, emb, rows, len);
        MST_CHECK_LAUNCH("enc_avgpool_kernel");
    }
    return MST_OK;
}

}  // namespace

extern "C" size_t mst_enc_workspace_bytes(const MstEnc *e, int B, int L) {
    if (!e || B < 1 || L < 1) return 0;
    return 3 * align_up(enc_buf_floats(e, B, L) * sizeof(float), 256) + align_up(enc_scratch_floats(e, B, L) * sizeof(float), 256);
}

extern "C" int mst_enc_forward(MstEnc *e, const float *x, float *emb, int B, int L, int precision, void *ws,
                               size_t ws_bytes, void *stream) {
    if (!emb) return fail(MST_ERR_ARG, "mst_enc_forward: null output");
    return enc_run(e, x, emb, nullptr, B, L, precision, e ? e->d.nblocks : 0, ws, ws_bytes, stream);
}

extern "C" int mst_enc_forward_blocks(MstEnc *e, const float *x, float *out, int B, int L, int precision, int n_run,
                                      void *ws, size_t ws_bytes, void *stream) {
    if (!e || !out || n_run < 1 || n_run > e->d.nblocks) return fail(MST_ERR_ARG, "mst_enc_forward_blocks: bad argument");
    return enc_run(e, x, nullptr, out, B, L, precision, n_run, ws, ws_bytes, stream);
}

extern "C" int mst_enc_forward_conv(MstEnc *e, int block, int which, const float *x, float *y, int B, int L, void *stream) {
    if (!e || !x || !y || B < 1 || L < 1 || block < 0 || block >= e->d.nblocks || which < 0 || which > 1)
        return fail(MST_ERR_ARG, "mst_enc_forward_conv: bad argument");
    const MstEncConv &c = e->conv[2 * block + which];
    if (!c.loaded) return fail(MST_ERR_STATE, "mst_enc_forward_conv: conv weights not loaded");
    const int lout = conv_out_length(c, L);
    if (lout < 1) return fail(MST_ERR_ARG, "mst_enc_forward_conv: input shorter than the kernel");
    return enc_launch(c, x, y, B, L, lout, 0, MST_PREC_F32, stream, nullptr, e->schedule);
}

extern "C" int mst_embedding_mean(const float *emb, int n_rows, int dim, float *out, void *stream) {
    if (!emb || !out || n_rows < 1 || dim < 1) return fail(MST_ERR_ARG, "mst_embedding_mean: bad argument");
    MST_LAUNCH(embedding_mean_kernel, dim3((dim + 255) / 256), dim3(256), stream, emb, n_rows, dim, out);
    MST_CHECK_LAUNCH("embedding_mean_kernel");
    return MST_OK;
}

