// libmst_hip.so - C ABI implementation (host side): opaque handles, BatchNorm folding + weight packing into
// MFMA fragment order, tile geometry and kernel launches.  See include/mst_hip.h for the contract.
#include "../../include/mst_hip.h"

#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "enc_kernels.h"
#include "fft_kernels.h"
#include "fx_kernels.h"
#include "tcn_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define MST_HIP_TRY(expr)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(MST_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define MST_CHECK_LAUNCH(name)                                                                     \
    do {                                                                                           \
        hipError_t e_ = hipGetLastError();                                                         \
        if (e_ != hipSuccess) return fail(MST_ERR_HIP, std::string(name) + ": " + hipGetErrorString(e_)); \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <typename T> int upload(T **dev, const std::vector<T> &host) {
    if (*dev == nullptr) MST_HIP_TRY(hipMalloc((void **)dev, host.size() * sizeof(T)));
    MST_HIP_TRY(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return MST_OK;
}

// eval-mode BatchNorm as y = x*scale + shift
// host-side weight packing of the large encoder layers (81 M parameters, five images): the independent tiles of an image on a few threads
template <typename F> void host_parallel_for(int n, F fn) {
    const int nt = std::max(1, std::min({n, 16, (int)std::thread::hardware_concurrency()}));
    if (nt == 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([=]() {
            for (int i = t; i < n; i += nt) fn(i);
        });
    for (auto &t : th) t.join();
}

void bn_fold(const float *w, const float *b, const float *mean, const float *var, float eps, int c,
             std::vector<float> &scale, std::vector<float> &shift) {
    scale.resize(c);
    shift.resize(c);
    for (int i = 0; i < c; ++i) {
        scale[i] = w[i] / std::sqrt(var[i] + eps);
        shift[i] = b[i] - mean[i] * scale[i];
    }
}

}  // namespace

// one convolution layer as packed for enc_conv_kernel / enc_conv_bf16_kernel / the NLC pipeline
struct MstEncConv {
    float *wpk = nullptr, *shift = nullptr;
    __bf16 *wpk16 = nullptr;
    int *ktab = nullptr;
    float *w_direct = nullptr;   // [Cout][Cin][ksz] folded fp32 (layers with Cin < 8: direct kernel)
    __bf16 *w_taps = nullptr;    // 128-channel layers (Cin a multiple of 64, k = 5 / 10): enc_conv_taps_kernel's A fragments (enc_taps_pack)
    __bf16 *w_frag16 = nullptr;  // blocks 1 / 2 of the default encoder (Cin = 16, k = 25 / Cin = 32, k = 15): the fused kernel's bf16 A fragments, one 16-row tile after the other (enc_block1_pack)
    float *w_frag = nullptr;     // stereo block (Cin = 2, k = 25): the fused kernel's fp32 MFMA A fragments (enc_stereo_pack_a0 / _a1)
    __bf16 *wpk_nlc = nullptr;   // NLC pipeline A fragments, k = j*Cin + ci, K-chunk 64
    float slope = 0.0f;             // activation slope for negative values: 0 ReLU, 0.01 LeakyReLU, 1 none (MstEncDesc.act_slope)
    __bf16 *wpk_nlc_lo = nullptr;   // split mode: bf16(W' - bf16(W')) in the same fragment order
    int *stab = nullptr;         // NLC pipeline slot table
    int nchunks64 = 0;
    int cin = 0, cout = 0, ksz = 0, stride = 1, dil = 1, pad_l = 0, pad_r = 0, nchunks = 0, nchunks32 = 0, mw = 4;
    bool loaded = false;
};


namespace {
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
}  // namespace

// =================================================================================================
// TCN
// =================================================================================================
struct MstTcnBlock {
    void *w_bf16 = nullptr;   // blocks >= 1: [60][2][4][64][8] bf16 (A fragments of v_mfma_f32_16x16x32_bf16)
    void *w_x3 = nullptr;     // blocks >= 1: [hi | lo][120][4][64][8] bf16 (bf16x3 mode: W' = W'_hi + W'_lo)
    float *w_f32 = nullptr;   // blocks >= 1: [15][4][4][4][64][4] fp32 ; block 0: [2][15][128]
    float *shift = nullptr;   // [128]
    float *res = nullptr;     // [128]
    bool loaded = false;
};

struct MstTcn {
    MstTcnDesc d;
    bool generic = false;              // configuration outside the specialised 128-channel / k=15 kernels
    std::vector<MstEncConv> gconv;     // generic path: one packed conv per block + the output head (fp32 implicit GEMM)
    std::vector<MstTcnBlock> blk;
    float *film_w = nullptr;  // [nblocks][2C][D]
    float *film_b = nullptr;  // [nblocks][2C]
    float *film = nullptr;    // [nblocks][rows][2C]
    int film_rows = 0, film_cap = 0;
    float *out_w = nullptr, *out_b = nullptr;
    bool out_loaded = false;
    void *zero_row = nullptr;     // 1 KB of zeros: what the block kernels stage for time steps outside the segment
    int x3_small_tiles = 1;       // bf16x3 mode: 128-time tiles of <= 2 phases, two workgroups per CU (mst_tcn_set_tuning; measured 5.13 vs 5.45 ms)
    int x3_half_cm = 1;           // bf16x3 mode: class-major loop in the eight-phase half-tile kernel (mst_tcn_set_tuning bit 6; round 5: GPU-tested,
                                  // 566 -> 572 segments/s at 32 x 131072, profiles/r05_x3_ab_bit6_53_117.jsonl: on)
    int bf16_fuse0 = 1;           // bf16 mode: block 0 computed by the loader waves of block 1's duo kernel (mst_tcn_set_tuning bit 5; measured -0.2 ms
                                  // per forward, bit-identical to the separate kernel; default since round 5 - tests/test_gpu_parity.py form 53)
    int last_fused0 = 0;          // whether the last forward of this handle really ran block 0 inside block 1's launch (mst_tcn_get_tuning)
    int bf16_reuse = 1;           // bf16 mode, duo kernel: the class-major main loop (mst_tcn_set_tuning bit 4; measured 1.40 vs 1.46 ms per launch)
    int bf16_form = 2;            // bf16 mode, form of the block kernel (mst_tcn_set_tuning bits 1-2): 0 one tile per workgroup, 2 duo (default)
    std::vector<hipEvent_t> ev;   // timing hook: (nblocks + 2) events per recorded forward
    int ev_max = 0, ev_used = 0;
};

extern "C" int mst_version(void) { return 100; }
extern "C" const char *mst_last_error(void) { return g_err.c_str(); }

extern "C" int mst_tcn_create(const MstTcnDesc *desc, MstTcn **out) {
    if (!desc || !out) return fail(MST_ERR_ARG, "mst_tcn_create: null argument");
    const MstTcnDesc &d = *desc;
    if (d.nblocks < 1 || d.nblocks > MST_MAX_BLOCKS) return fail(MST_ERR_ARG, "mst_tcn_create: nblocks out of range");
    if (d.channels < 1 || d.kernel_size < 1 || d.ninputs < 1 || d.noutputs < 1 || d.cond_dim < 1)
        return fail(MST_ERR_ARG, "mst_tcn_create: bad layer description");
    if (d.channels % d.ninputs != 0)
        return fail(MST_ERR_UNSUPPORTED, "mst_tcn_create: channel_width must be a multiple of ninputs (grouped 1x1 residual)");
    const bool fast = d.channels == 128 && d.kernel_size == 15 && d.ninputs == 2 && d.noutputs <= 2 && d.dilations[0] == 1 && !d.causal;
    for (int n = 0; n < d.nblocks; ++n)
        if (d.dilations[n] < 1) return fail(MST_ERR_ARG, "mst_tcn_create: dilation < 1");
    MstTcn *t = new MstTcn();
    t->d = d;
    t->generic = !fast;
    t->blk.resize(d.nblocks);
    if (t->generic) {
        t->gconv.resize(d.nblocks + 1);
        for (int n = 0; n < d.nblocks; ++n) {
            const int span = (d.kernel_size - 1) * d.dilations[n];          // architectures.py:199: span/2 each side, or all of it on the
            const int pad_l = d.causal ? span : span / 2;                   // left for a causal block (pad both sides, drop the tail)
            conv_geometry(t->gconv[n], n == 0 ? d.ninputs : d.channels, d.channels, d.kernel_size, 1, d.dilations[n], pad_l, span - pad_l);
        }
        conv_geometry(t->gconv[d.nblocks], d.channels, d.noutputs, 1, 1, 1, 0, 0);
    }
    const size_t fw = (size_t)d.nblocks * 2 * d.channels * d.cond_dim;
    if (hipMalloc((void **)&t->film_w, fw * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&t->film_b, (size_t)d.nblocks * 2 * d.channels * sizeof(float)) != hipSuccess ||
        hipMalloc(&t->zero_row, 1024) != hipSuccess || hipMemset(t->zero_row, 0, 1024) != hipSuccess) {
        (void)hipFree(t->film_w);
        (void)hipFree(t->film_b);
        (void)hipFree(t->zero_row);
        delete t;
        return fail(MST_ERR_HIP, "mst_tcn_create: hipMalloc failed");
    }
    *out = t;
    return MST_OK;
}

extern "C" int mst_tcn_destroy(MstTcn *t) {
    if (!t) return MST_OK;
    for (auto &b : t->blk) {
        (void)hipFree(b.w_bf16);
        (void)hipFree(b.w_x3);
        (void)hipFree(b.w_f32);
        (void)hipFree(b.shift);
        (void)hipFree(b.res);
    }
    (void)hipFree(t->film_w);
    (void)hipFree(t->film_b);
    (void)hipFree(t->film);
    (void)hipFree(t->out_w);
    (void)hipFree(t->out_b);
    (void)hipFree(t->zero_row);
    for (auto &c : t->gconv) {
        (void)hipFree(c.wpk);
        (void)hipFree(c.ktab);
        (void)hipFree(c.shift);
    }
    for (auto e : t->ev) (void)hipEventDestroy(e);
    delete t;
    return MST_OK;
}

extern "C" int mst_tcn_load_block(MstTcn *t, int n, const float *conv_w, const float *bn_weight, const float *bn_bias,
                                  const float *bn_mean, const float *bn_var, float bn_eps, const float *film_w,
                                  const float *film_b, const float *res_w, void *) {
    if (!t || !conv_w || !bn_weight || !bn_bias || !bn_mean || !bn_var || !film_w || !film_b || !res_w)
        return fail(MST_ERR_ARG, "mst_tcn_load_block: null argument");
    if (n < 0 || n >= t->d.nblocks) return fail(MST_ERR_ARG, "mst_tcn_load_block: block index out of range");
    const int C = t->d.channels, K = t->d.kernel_size;
    const int cin = n == 0 ? t->d.ninputs : C;
    std::vector<float> scale, shift;
    bn_fold(bn_weight, bn_bias, bn_mean, bn_var, bn_eps, C, scale, shift);
    MstTcnBlock &b = t->blk[n];
    if (t->generic) {
        int rc;
        MstEncConv &c = t->gconv[n];
        if ((rc = pack_conv_f32(c, conv_w, scale))) return rc;
        std::vector<float> sh((size_t)((C + 32 * c.mw - 1) / (32 * c.mw)) * 32 * c.mw, 0.0f);
        for (int co = 0; co < C; ++co) sh[co] = shift[co];
        if ((rc = upload(&c.shift, sh))) return rc;
        std::vector<float> res(res_w, res_w + C);
        if ((rc = upload(&b.res, res))) return rc;
        const size_t fwn = (size_t)2 * C * t->d.cond_dim;
        MST_HIP_TRY(hipMemcpy(t->film_w + (size_t)n * fwn, film_w, fwn * sizeof(float), hipMemcpyHostToDevice));
        MST_HIP_TRY(hipMemcpy(t->film_b + (size_t)n * 2 * C, film_b, 2 * C * sizeof(float), hipMemcpyHostToDevice));
        c.loaded = b.loaded = true;
        return MST_OK;
    }
    auto W = [&](int co, int ci, int j) { return conv_w[((size_t)co * cin + ci) * K + j] * scale[co]; };
    int rc;
    if (n == 0) {
        std::vector<float> w0((size_t)cin * K * C);
        for (int ci = 0; ci < cin; ++ci)
            for (int j = 0; j < K; ++j)
                for (int co = 0; co < C; ++co) w0[((size_t)ci * K + j) * C + co] = W(co, ci, j);
        if ((rc = upload(&b.w_f32, w0))) return rc;
        if (cin == 2 && K == 15) {        // bf16 A fragments of the matrix-core block-0 kernel: [s][wave][lane][e], k = ci * 15 + j
            std::vector<__bf16> wb((size_t)2 * 4 * 64 * 8);
            for (int sI = 0; sI < 2; ++sI)
                for (int w = 0; w < 4; ++w)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const int k = 16 * sI + 8 * (l >> 5) + e;
                            wb[(((size_t)sI * 4 + w) * 64 + l) * 8 + e] = k < 30 ? (__bf16)W(32 * w + (l & 31), k / 15, k % 15) : (__bf16)0.0f;
                        }
            if ((rc = upload((__bf16 **)&b.w_bf16, wb))) return rc;
        }
    } else {
        // bf16 A fragments of v_mfma_f32_16x16x32_bf16: [ks = j*4 + kk][row tile m][wave][lane][e]
        std::vector<__bf16> wb((size_t)120 * 4 * 64 * 8);
        for (int j = 0; j < K; ++j)
            for (int kk = 0; kk < 4; ++kk)
                for (int m = 0; m < 2; ++m)
                    for (int w = 0; w < 4; ++w)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e)
                                wb[(((((size_t)(j * 4 + kk) * 2 + m) * 4 + w) * 64 + l) * 8) + e] =
                                    (__bf16)W(32 * w + 16 * m + (l & 15), 32 * kk + 8 * (l >> 4) + e, j);
        if ((rc = upload((__bf16 **)&b.w_bf16, wb))) return rc;
        // bf16x3 mode: the same fragment image twice, W'_hi = bf16(W') and W'_lo = bf16(W' - W'_hi)
        std::vector<__bf16> wx((size_t)2 * 120 * 4 * 64 * 8);
        for (int j = 0; j < K; ++j)
            for (int kk = 0; kk < 4; ++kk)
                for (int m = 0; m < 2; ++m)
                    for (int w = 0; w < 4; ++w)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                const float v = W(32 * w + 16 * m + (l & 15), 32 * kk + 8 * (l >> 4) + e, j);
                                const __bf16 hi = (__bf16)v;
                                const size_t idx = (((((size_t)(j * 4 + kk) * 2 + m) * 4 + w) * 64 + l) * 8) + e;
                                wx[idx] = hi;
                                wx[(size_t)120 * 4 * 64 * 8 + idx] = (__bf16)(v - (float)hi);
                            }
        if ((rc = upload((__bf16 **)&b.w_x3, wx))) return rc;
        // fp32 A fragments of v_mfma_f32_32x32x2_f32: [j][chunk c][ksg][wave][lane][i]
        std::vector<float> wf((size_t)K * 4 * 4 * 4 * 64 * 4);
        for (int j = 0; j < K; ++j)
            for (int c = 0; c < 4; ++c)
                for (int ksg = 0; ksg < 4; ++ksg)
                    for (int w = 0; w < 4; ++w)
                        for (int l = 0; l < 64; ++l)
                            for (int i = 0; i < 4; ++i)
                                wf[(((((size_t)(j * 4 + c) * 4 + ksg) * 4 + w) * 64 + l) * 4) + i] =
                                    W(32 * w + (l & 31), 32 * c + 2 * (4 * ksg + i) + (l >> 5), j);
        if ((rc = upload(&b.w_f32, wf))) return rc;
    }
    if ((rc = upload(&b.shift, shift))) return rc;
    std::vector<float> res(res_w, res_w + C);
    if ((rc = upload(&b.res, res))) return rc;
    const size_t fwn = (size_t)2 * C * t->d.cond_dim;
    MST_HIP_TRY(hipMemcpy(t->film_w + (size_t)n * fwn, film_w, fwn * sizeof(float), hipMemcpyHostToDevice));
    MST_HIP_TRY(hipMemcpy(t->film_b + (size_t)n * 2 * C, film_b, 2 * C * sizeof(float), hipMemcpyHostToDevice));
    b.loaded = true;
    return MST_OK;
}

extern "C" int mst_tcn_load_output(MstTcn *t, const float *w, const float *b, void *) {
    if (!t || !w || !b) return fail(MST_ERR_ARG, "mst_tcn_load_output: null argument");
    int rc;
    if (t->generic) {
        MstEncConv &c = t->gconv[t->d.nblocks];
        std::vector<float> ones(t->d.noutputs, 1.0f);
        if ((rc = pack_conv_f32(c, w, ones))) return rc;
        std::vector<float> sh((size_t)32 * c.mw * ((t->d.noutputs + 32 * c.mw - 1) / (32 * c.mw)), 0.0f);
        for (int o = 0; o < t->d.noutputs; ++o) sh[o] = b[o];
        if ((rc = upload(&c.shift, sh))) return rc;
        c.loaded = t->out_loaded = true;
        return MST_OK;
    }
    std::vector<float> wv(w, w + (size_t)t->d.noutputs * 128), bv(b, b + t->d.noutputs);
    if ((rc = upload(&t->out_w, wv))) return rc;
    if ((rc = upload(&t->out_b, bv))) return rc;
    t->out_loaded = true;
    return MST_OK;
}

extern "C" int mst_tcn_set_cond(MstTcn *t, const float *cond_dev, int n_rows, long block_stride, void *stream) {
    if (!t || !cond_dev || n_rows < 1 || block_stride < 0) return fail(MST_ERR_ARG, "mst_tcn_set_cond: bad argument");
    for (auto &b : t->blk)
        if (!b.loaded) return fail(MST_ERR_STATE, "mst_tcn_set_cond: block weights not loaded");
    if (n_rows > t->film_cap) {
        (void)hipFree(t->film);
        t->film = nullptr;
        t->film_cap = 0;
        MST_HIP_TRY(hipMalloc((void **)&t->film, (size_t)t->d.nblocks * n_rows * 2 * t->d.channels * sizeof(float)));
        t->film_cap = n_rows;
    }
    FilmArgs a;
    a.fw = t->film_w;
    a.fb = t->film_b;
    a.cond = cond_dev;
    a.film = t->film;
    a.nblocks = t->d.nblocks;
    a.two_c = 2 * t->d.channels;
    a.D = t->d.cond_dim;
    a.rows = n_rows;
    a.block_stride = block_stride;
    const int outs = t->d.nblocks * 2 * t->d.channels;
    MST_LAUNCH(tcn_film_kernel, dim3((outs + 3) / 4), dim3(256), stream, a);
    MST_CHECK_LAUNCH("tcn_film_kernel");
    t->film_rows = n_rows;
    return MST_OK;
}

namespace {

size_t tcn_elem(int precision) { return precision == MST_PREC_BF16 ? 2 : 4; }      // bf16x3 keeps fp32 activations in HBM

// phases per tile: P | d.  P = 4 with 256-time tiles (78 KB of LDS, 2 workgroups per CU) whenever a tile's 64 steps
// fit the segment; for larger dilations P = 8 with 128-time tiles (16 steps per tile, 61 KB, still 2 per CU); P = 16
// (256-time tiles, 16 steps per tile) only for segments with fewer than 16 steps per phase.
int choose_phases(int d, int L, int precision) {
    int P = (d % 4 == 0) ? 4 : (d % 2 == 0 ? 2 : 1);
    const long nsteps = ((long)L + d - 1) / d;
    if (precision == MST_PREC_BF16X3) {    // two LDS tiles (hi, lo): 256-time tiles up to P = 4, 128-time tiles of 8 phases for large dilations
        if (P == 4 && d % 8 == 0 && 256 / P > nsteps) P = 8;
        return P;
    }
    if (precision == MST_PREC_BF16X3 + 100) {   // bf16x3 with small tiles: 2 phases wherever 64 steps fit the segment
        int Q = (d % 2 == 0) ? 2 : 1;
        if (128 / Q <= nsteps) return Q;
        return choose_phases(d, L, MST_PREC_BF16X3);
    }
    if (precision != MST_PREC_BF16) {      // fp32 kernel: 256-time tiles only (its LDS tile is a 32-channel chunk)
        while (P < 16 && d % (2 * P) == 0 && 256 / P > nsteps) P *= 2;
        return P;
    }
    while (P < 8 && d % (2 * P) == 0 && 256 / P > nsteps) P *= 2;
    if (P == 8 && d % 16 == 0 && nsteps < 16) P = 16;     // very short segments: 16-step tiles of 16 phases
    return P;
}

// the persistent double-tile bf16 kernel: one workgroup per CU
template <int P, int NQ> int launch_block_duo(TcnBlockArgs a, void *stream, int reuse = 0) {
    if (a.x0 && !(P == 2 && NQ == 8 && reuse)) return fail(MST_ERR_STATE, "tcn_block_bf16_duo_kernel: block 0 can only be fused into two-phase class-major tiles");
    const long nsteps = ((long)a.L + a.d - 1) / a.d;
    a.tiles_step = (int)((nsteps + (32 * NQ) / P - 1) / ((32 * NQ) / P));
    const long ntiles = (long)a.B * a.tiles_phase * a.tiles_step;
    if (ntiles > 0x7fffffffL) return fail(MST_ERR_ARG, "tcn_block_bf16_duo_kernel: more than 2^31 tiles");
    long grid = mst_num_cus();
    if (grid > ntiles) grid = ntiles;
    a.xcd_tiles = 0;
    if (grid >= 8) {
        grid -= grid % 8;
        a.xcd_tiles = (int)((ntiles + 7) / 8);
    }
    if constexpr (P == 2 && NQ == 8) {
        if (reuse && a.x0) {          // block 0 computed by the loader waves (mst_tcn_set_tuning bit 5)
            MST_LAUNCH((tcn_block_bf16_duo_kernel<P, false, NQ, true, true>), dim3((unsigned)grid), dim3(512), stream, a);
            MST_CHECK_LAUNCH("tcn_block_bf16_duo_kernel");
            return MST_OK;
        }
    }
    if constexpr ((P == 4 || P == 2) && NQ == 8) {
        if (reuse) {          // the class-major main loop (B fragments reused across the taps of a class)
            MST_LAUNCH((tcn_block_bf16_duo_kernel<P, false, NQ, true>), dim3((unsigned)grid), dim3(512), stream, a);
            MST_CHECK_LAUNCH("tcn_block_bf16_duo_kernel");
            return MST_OK;
        }
    }
    MST_LAUNCH((tcn_block_bf16_duo_kernel<P, false, NQ>), dim3((unsigned)grid), dim3(512), stream, a);
    MST_CHECK_LAUNCH("tcn_block_bf16_duo_kernel");
    return MST_OK;
}

template <int P> int launch_block(int precision, const TcnBlockArgs &a0, int grid, void *stream, int x3_small = 0, int bf16_form = 0,
                                  int bf16_small4 = 0, int bf16_reuse = 0, int x3_half_cm = 0) {
    TcnBlockArgs a = a0;
    if constexpr (P == 4) {
        // (the same 128-time form for EVERY block - three workgroups per CU instead of the duo kernel - measured 1.53-1.58 ms per launch
        //  against 1.48-1.53: it only wins where the eight-phase tiles' halo is the alternative)
        if (precision == MST_PREC_BF16 && bf16_small4) {          // 128-time tiles of 4 phases (one-tile kernel, three workgroups per CU)
            const long nsteps = ((long)a.L + a.d - 1) / a.d;
            a.tiles_step = (int)((nsteps + 128 / P - 1) / (128 / P));
            const long g2 = (long)a.B * a.tiles_phase * a.tiles_step;
            if (g2 % 8 == 0) a.xcd_tiles = (int)(g2 / 8);
            if (a.y_out)
                MST_LAUNCH((tcn_block_bf16_kernel<P, true, 4>), dim3((unsigned)g2), dim3(256), stream, a);
            else
                MST_LAUNCH((tcn_block_bf16_kernel<P, false, 4>), dim3((unsigned)g2), dim3(256), stream, a);
            MST_CHECK_LAUNCH("tcn_block_bf16_kernel");
            return MST_OK;
        }
    }
    if (precision == MST_PREC_BF16 && bf16_form == 2) {
        // 256-time tiles only: at P = 8 (128-time tiles: half the work per tile for the same two barriers) the duo form measured
        // 1.62-1.82 ms against 1.50 ms, those blocks run the one-tile-per-workgroup kernel
        // (the last block - fused output head, 32 more live registers - spills in the duo form and runs the one-tile kernel too)
        if constexpr (P <= 4) {
            if (!a.y_out) return launch_block_duo<P, 8>(a, stream, bf16_reuse);
        }
    }
    if (precision == MST_PREC_BF16X3) {
        if constexpr (P <= 2) {
            if (x3_small) {          // 128-time tiles: 2 x 39 KB of LDS, two workgroups (8 waves) per CU
                const long nsteps = ((long)a.L + a.d - 1) / a.d;
                a.tiles_step = (int)((nsteps + 128 / P - 1) / (128 / P));
                const long g2 = (long)a.B * a.tiles_phase * a.tiles_step;
                if (g2 % 8 == 0) a.xcd_tiles = (int)(g2 / 8);
                MST_LAUNCH((tcn_block_bf16x3_kernel<P, 4>), dim3((unsigned)g2), dim3(256), stream, a);
                MST_CHECK_LAUNCH("tcn_block_bf16x3_kernel");
                return MST_OK;
            }
        }
        if constexpr (P <= 8) {
            constexpr int NQ = P == 8 ? 4 : 8;
            const long nsteps = ((long)a.L + a.d - 1) / a.d;
            a.tiles_step = (int)((nsteps + (32 * NQ) / P - 1) / ((32 * NQ) / P));
            const long g2 = (long)a.B * a.tiles_phase * a.tiles_step;
            if (g2 % 8 == 0) a.xcd_tiles = (int)(g2 / 8);
            if constexpr (P == 8) {    // 8-phase tiles: the input staged in two halves of 64 channels (60 KB of LDS, two workgroups per CU)
                if (x3_half_cm)
                    MST_LAUNCH((tcn_block_bf16x3_half_kernel<P, NQ, true>), dim3((unsigned)g2), dim3(256), stream, a);
                else
                    MST_LAUNCH((tcn_block_bf16x3_half_kernel<P, NQ>), dim3((unsigned)g2), dim3(256), stream, a);
            } else
            MST_LAUNCH((tcn_block_bf16x3_kernel<P, NQ>), dim3((unsigned)g2), dim3(256), stream, a);
            MST_CHECK_LAUNCH("tcn_block_bf16x3_kernel");
            return MST_OK;
        } else {
            return fail(MST_ERR_UNSUPPORTED, "tcn_block_bf16x3_kernel: no 16-phase form");
        }
    }
    if (precision == MST_PREC_BF16) {
        // XCD-aware tile order: measured read traffic 1.38 -> 1.20 GB per launch at P = 4 (1.07 algorithmic)
        constexpr int xcd_on = 1;
        if constexpr (P == 8) {
            // P = 8 tiles of 256 times need 94 KB of LDS (one workgroup per CU); 128-time tiles (61 KB) keep two resident:
            // measured 1.98 -> 1.70 ms for the d = 4096 block at L = 131072
            const long nsteps = ((long)a.L + a.d - 1) / a.d;
            a.tiles_step = (int)((nsteps + 128 / P - 1) / (128 / P));
            const long g2 = (long)a.B * a.tiles_phase * a.tiles_step;
            if (xcd_on && g2 % 8 == 0) a.xcd_tiles = (int)(g2 / 8);
            if (a.y_out)
                MST_LAUNCH((tcn_block_bf16_kernel<P, true, 4>), dim3((unsigned)g2), dim3(256), stream, a);
            else
                MST_LAUNCH((tcn_block_bf16_kernel<P, false, 4>), dim3((unsigned)g2), dim3(256), stream, a);
        } else {
            if (xcd_on && grid % 8 == 0) a.xcd_tiles = grid / 8;
            if (a.y_out)
                MST_LAUNCH((tcn_block_bf16_kernel<P, true, 8>), dim3(grid), dim3(256), stream, a);
            else
                MST_LAUNCH((tcn_block_bf16_kernel<P, false, 8>), dim3(grid), dim3(256), stream, a);
        }
    } else {
        if (grid % 8 == 0) a.xcd_tiles = grid / 8;
        MST_LAUNCH((tcn_block_f32_kernel<P>), dim3(grid), dim3(256), stream, a);
    }
    MST_CHECK_LAUNCH("tcn_block_kernel");
    return MST_OK;
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

int tcn_run_generic(MstTcn *t, const float *x, float *y, float *act_out, int B, int L, int n_run, void *ws, void *stream) {
    const int C = t->d.channels;
    const size_t buf_bytes = align_up((size_t)B * L * C * sizeof(float), 256);
    float *buf[2] = {(float *)ws, (float *)((unsigned char *)ws + buf_bytes)};
    const float *cur = x;
    int rc, pp = 0;
    for (int n = 0; n < n_run; ++n) {
        float *dst = (act_out && n == n_run - 1) ? act_out : buf[pp];
        const int cin = n == 0 ? t->d.ninputs : C;
        if ((rc = tcn_launch_generic(t->gconv[n], cur, dst, B, L, 1, t->film + (size_t)n * t->film_rows * 2 * C, t->film_rows,
                                     t->blk[n].res, C / cin, stream)))
            return rc;
        cur = dst;
        pp ^= 1;
    }
    if (act_out) return MST_OK;
    return tcn_launch_generic(t->gconv[t->d.nblocks], cur, y, B, L, 2, nullptr, 1, nullptr, 1, stream);
}

int tcn_run(MstTcn *t, const float *x, float *y, float *act_out, int B, int L, int precision, int n_run, void *ws,
            size_t ws_bytes, void *stream) {
    if (!t || !x || B < 1 || L < 1) return fail(MST_ERR_ARG, "mst_tcn_forward: bad argument");
    if (precision != MST_PREC_F32 && precision != MST_PREC_BF16 && precision != MST_PREC_BF16X3)
        return fail(MST_ERR_ARG, "mst_tcn_forward: bad precision");
    for (auto &b : t->blk)
        if (!b.loaded) return fail(MST_ERR_STATE, "mst_tcn_forward: block weights not loaded");
    if (!t->out_loaded) return fail(MST_ERR_STATE, "mst_tcn_forward: output conv not loaded");
    if (t->film_rows == 0) return fail(MST_ERR_STATE, "mst_tcn_forward: mst_tcn_set_cond has not been called");
    if (t->film_rows != 1 && t->film_rows != B)
        return fail(MST_ERR_ARG, "mst_tcn_forward: condition rows must be 1 or equal the batch size");
    const size_t need = mst_tcn_workspace_bytes(t, B, L, precision);
    if (!ws || ws_bytes < need) return fail(MST_ERR_WORKSPACE, "mst_tcn_forward: workspace too small");
    if (t->generic) return tcn_run_generic(t, x, y, act_out, B, L, n_run, ws, stream);
    const size_t es = tcn_elem(precision);
    const size_t buf_bytes = align_up((size_t)B * L * 128 * es, 256);
    unsigned char *buf[2] = {(unsigned char *)ws, (unsigned char *)ws + buf_bytes};
    const int Lp = L;
    hipEvent_t *ev = nullptr;
    if (!act_out && t->ev_used < t->ev_max) {
        ev = t->ev.data() + (size_t)t->ev_used * (t->d.nblocks + 2);
        t->ev_used++;
        MST_HIP_TRY(hipEventRecord(ev[0], (hipStream_t)stream));
    }

    // block 0 inside block 1's launch (bf16, tuning bit 5): block 1 must be the d = 2 block on the duo kernel's two-phase class-major tiles
    // and not the last block; the probes of block 0 itself (n_run == 1) always run the separate kernel
    const bool fuse0 = precision == MST_PREC_BF16 && t->bf16_fuse0 && t->bf16_reuse && t->bf16_form == 2 && t->blk[0].w_bf16 && n_run >= 2 &&
                       t->d.nblocks > 2 && t->d.dilations[0] == 1 && t->d.dilations[1] == 2 && choose_phases(2, L, precision) == 2;
    t->last_fused0 = fuse0 ? 1 : 0;
    if (fuse0) {
        if (ev) MST_HIP_TRY(hipEventRecord(ev[1], (hipStream_t)stream));
    } else {
        TcnBlock0Args a;
        a.x = x;
        a.y = buf[0];
        a.w = t->blk[0].w_f32;
        a.shift = t->blk[0].shift;
        a.film = t->film;
        a.res = t->blk[0].res;
        a.film_rows = t->film_rows;
        a.B = B;
        a.L = L;
        a.Lp = Lp;
        const int grid = B * ((L + 511) / 512);      // 8 tiles of 64 steps per workgroup
        a.wpk16 = t->blk[0].w_bf16;
        if (precision == MST_PREC_BF16 && a.wpk16)
            MST_LAUNCH(tcn_block0_mfma_kernel, dim3(B * ((L + 255) / 256)), dim3(256), stream, a);
        else if (precision == MST_PREC_BF16)
            MST_LAUNCH((tcn_block0_kernel<__bf16>), dim3(grid), dim3(256), stream, a);
        else
            MST_LAUNCH((tcn_block0_kernel<float>), dim3(grid), dim3(256), stream, a);
        MST_CHECK_LAUNCH("tcn_block0_kernel");
        if (ev) MST_HIP_TRY(hipEventRecord(ev[1], (hipStream_t)stream));
    }
    int cur = 0;
    bool fused_head = false;
    for (int n = 1; n < n_run; ++n) {
        const int d = t->d.dilations[n];
        int P = choose_phases(d, L, (precision == MST_PREC_BF16X3 && t->x3_small_tiles) ? MST_PREC_BF16X3 + 100 : precision);
        const int x3_small = (precision == MST_PREC_BF16X3 && t->x3_small_tiles && P <= 2) ? 1 : 0;
        // bf16, 17 ... 32 steps per phase (d = 4096 at L = 131072): 128-time tiles of FOUR phases x 32 steps (184 rows staged per 128
        // outputs, three workgroups per CU) instead of eight phases x 16 steps (240 rows, two workgroups per CU)
        int bf16_small4 = 0;
        if (precision == MST_PREC_BF16 && P == 8) {
            const long ns = ((long)L + d - 1) / d;
            if (ns > 16 && ns <= 32) {
                P = 4;
                bf16_small4 = 1;
            }
        }
        TcnBlockArgs a;
        a.x = buf[cur];
        a.y = buf[cur ^ 1];
        a.wpk = precision == MST_PREC_BF16 ? t->blk[n].w_bf16 : (precision == MST_PREC_BF16X3 ? t->blk[n].w_x3 : (void *)t->blk[n].w_f32);
        a.shift = t->blk[n].shift;
        a.film = t->film + (size_t)n * t->film_rows * 256;
        a.res = t->blk[n].res;
        a.film_rows = t->film_rows;
        a.B = B;
        a.L = L;
        a.Lp = Lp;
        a.d = d;
        a.tiles_phase = d / P;
        const long nsteps = ((long)L + d - 1) / d;
        a.tiles_step = (int)((nsteps + 256 / P - 1) / (256 / P));
        const long grid = (long)B * a.tiles_phase * a.tiles_step;
        // bf16 / bf16x3 modes: the last block applies the output head in its epilogue (no separate output kernel; the split mode's
        // kernels exist for up to 8 phases - every dilation of a 2^19-sample segment - otherwise the separate head runs)
        const bool fuse_out = (precision == MST_PREC_BF16 || (precision == MST_PREC_BF16X3 && P <= 8 && t->d.noutputs <= 2)) && !act_out &&
                              n == t->d.nblocks - 1;
        fused_head = fused_head || fuse_out;
        a.out_w = t->out_w;
        a.out_b = t->out_b;
        a.y_out = fuse_out ? y : nullptr;
        a.nout = t->d.noutputs;
        a.xcd_tiles = 0;
        a.zeros = t->zero_row;
        if (fuse0 && n == 1) {
            a.x0 = x;
            a.w0pk = t->blk[0].w_bf16;
            a.shift0 = t->blk[0].shift;
            a.film0 = t->film;
            a.res0 = t->blk[0].res;
        }
        if (grid > 0x7fffffffL) return fail(MST_ERR_ARG, "mst_tcn_forward: grid too large");
        int rc;
        switch (P) {
            case 1: rc = launch_block<1>(precision, a, (int)grid, stream, x3_small, t->bf16_form, bf16_small4, t->bf16_reuse, t->x3_half_cm); break;
            case 2: rc = launch_block<2>(precision, a, (int)grid, stream, x3_small, t->bf16_form, bf16_small4, t->bf16_reuse, t->x3_half_cm); break;
            case 4: rc = launch_block<4>(precision, a, (int)grid, stream, x3_small, t->bf16_form, bf16_small4, t->bf16_reuse, t->x3_half_cm); break;
            case 8: rc = launch_block<8>(precision, a, (int)grid, stream, x3_small, t->bf16_form, bf16_small4, t->bf16_reuse, t->x3_half_cm); break;
            default: rc = launch_block<16>(precision, a, (int)grid, stream, 0, t->bf16_form); break;
        }
        if (rc) return rc;
        if (ev) MST_HIP_TRY(hipEventRecord(ev[n + 1], (hipStream_t)stream));
        cur ^= 1;
    }
    if (act_out) {
        const size_t total = (size_t)B * L * 128;
        const unsigned grid = (unsigned)((total + 255) / 256);
        if (precision == MST_PREC_BF16)
            MST_LAUNCH((tcn_unpack_kernel<__bf16>), dim3(grid), dim3(256), stream, (const void *)buf[cur], act_out, B, L, Lp);
        else
            MST_LAUNCH((tcn_unpack_kernel<float>), dim3(grid), dim3(256), stream, (const void *)buf[cur], act_out, B, L, Lp);
        MST_CHECK_LAUNCH("tcn_unpack_kernel");
        return MST_OK;
    }
    if (fused_head && n_run == t->d.nblocks && t->d.nblocks > 1) {
        if (ev) {      // the output head ran inside the last block kernel
            MST_HIP_TRY(hipEventRecord(ev[t->d.nblocks + 1], (hipStream_t)stream));
        }
        return MST_OK;
    }
    TcnOutArgs o;
    o.x = buf[cur];
    o.y = y;
    o.w = t->out_w;
    o.bias = t->out_b;
    o.nout = t->d.noutputs;
    o.B = B;
    o.L = L;
    o.Lp = Lp;
    const int grid = B * ((L + 63) / 64);
    if (precision == MST_PREC_BF16)
        MST_LAUNCH((tcn_output_kernel<__bf16>), dim3(grid), dim3(256), stream, o);
    else
        MST_LAUNCH((tcn_output_kernel<float>), dim3(grid), dim3(256), stream, o);
    MST_CHECK_LAUNCH("tcn_output_kernel");
    if (ev) MST_HIP_TRY(hipEventRecord(ev[t->d.nblocks + 1], (hipStream_t)stream));
    return MST_OK;
}

}  // namespace

extern "C" int mst_tcn_set_tuning(MstTcn *t, int flags) {
    if (!t) return fail(MST_ERR_ARG, "mst_tcn_set_tuning: null handle");
    if (flags < 0 || flags > 127 || (((flags >> 1) & 3) != 0 && ((flags >> 1) & 3) != 2) || ((flags >> 3) & 1))
        return fail(MST_ERR_ARG, "mst_tcn_set_tuning: unknown flag bits (form 1 - the stream kernel - and bit 3 - the split-bf16 duo kernel - left the library in round 5)");
    t->x3_small_tiles = flags & 1;
    t->bf16_form = (flags >> 1) & 3;
    t->bf16_reuse = (flags >> 4) & 1;
    t->bf16_fuse0 = (flags >> 5) & 1;
    t->x3_half_cm = (flags >> 6) & 1;
    return MST_OK;
}

extern "C" int mst_tcn_get_tuning(const MstTcn *t, int *flags, int *last_forward_fused_block0) {
    if (!t) return fail(MST_ERR_ARG, "mst_tcn_get_tuning: null handle");
    if (flags)
        *flags = t->x3_small_tiles | t->bf16_form << 1 | t->bf16_reuse << 4 | t->bf16_fuse0 << 5 | t->x3_half_cm << 6;
    if (last_forward_fused_block0) *last_forward_fused_block0 = t->last_fused0;
    return MST_OK;
}

extern "C" int mst_tcn_timing_begin(MstTcn *t, int max_forwards) {
    if (!t || max_forwards < 1) return fail(MST_ERR_ARG, "mst_tcn_timing_begin: bad argument");
    for (auto e : t->ev) (void)hipEventDestroy(e);
    t->ev.assign((size_t)max_forwards * (t->d.nblocks + 2), nullptr);
    for (auto &e : t->ev) MST_HIP_TRY(hipEventCreate(&e));
    t->ev_max = max_forwards;
    t->ev_used = 0;
    return MST_OK;
}

extern "C" int mst_tcn_timing_end(MstTcn *t, float *ms_out, int *n_forwards) {
    if (!t || !ms_out || !n_forwards) return fail(MST_ERR_ARG, "mst_tcn_timing_end: bad argument");
    const int per = t->d.nblocks + 2;
    for (int k = 0; k <= t->d.nblocks; ++k) ms_out[k] = 0.0f;
    for (int f = 0; f < t->ev_used; ++f) {
        MST_HIP_TRY(hipEventSynchronize(t->ev[(size_t)f * per + per - 1]));
        for (int k = 0; k <= t->d.nblocks; ++k) {
            float ms = 0.0f;
            MST_HIP_TRY(hipEventElapsedTime(&ms, t->ev[(size_t)f * per + k], t->ev[(size_t)f * per + k + 1]));
            ms_out[k] += ms;
        }
    }
    if (t->ev_used > 0)
        for (int k = 0; k <= t->d.nblocks; ++k) ms_out[k] /= (float)t->ev_used;
    *n_forwards = t->ev_used;
    for (auto e : t->ev) (void)hipEventDestroy(e);
    t->ev.clear();
    t->ev_max = t->ev_used = 0;
    return MST_OK;
}

extern "C" int mst_calib_mainloop(int launches, float *ms_per_launch, float *sclk_mhz, void *stream) {
    if (launches < 2 || !ms_per_launch || !sclk_mhz) return fail(MST_ERR_ARG, "mst_calib_mainloop: bad argument");
    constexpr int WG = 512, REP = 32;                      // 512 x 32 tiles of 256 times = 32 x 131072 output steps
    const size_t wbytes = (size_t)120 * 256 * 16;          // 60 k-steps x 2 row tiles x 4 waves x 64 lanes x 16 B
    void *w = nullptr;
    float *out = nullptr;
    long long *clk = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = MST_OK;
    auto cleanup = [&]() {
        if (w) (void)hipFree(w);
        if (out) (void)hipFree(out);
        if (clk) (void)hipFree(clk);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
#define MST_CALIB_TRY(x)                                                        \
    if ((x) != hipSuccess) {                                                    \
        rc = fail(MST_ERR_HIP, "mst_calib_mainloop: HIP call failed");          \
        cleanup();                                                              \
        return rc;                                                              \
    }
    MST_CALIB_TRY(hipMalloc(&w, wbytes));
    MST_CALIB_TRY(hipMalloc((void **)&out, (size_t)WG * 256 * sizeof(float)));
    MST_CALIB_TRY(hipMalloc((void **)&clk, 2 * sizeof(long long)));
    MST_CALIB_TRY(hipEventCreate(&e0));
    MST_CALIB_TRY(hipEventCreate(&e1));
    MST_LAUNCH(tcn_calib_fill_kernel, dim3((unsigned)(wbytes / 4 + 255) / 256), dim3(256), stream, (unsigned *)w, (int)(wbytes / 4));
    const int warm = launches / 2, timed = launches - warm;
    for (int i = 0; i < warm; ++i) MST_LAUNCH(tcn_calib_mainloop_kernel, dim3(WG), dim3(256), stream, (const void *)w, out, clk, REP);
    MST_CALIB_TRY(hipEventRecord(e0, (hipStream_t)stream));
    for (int i = 0; i < timed; ++i) MST_LAUNCH(tcn_calib_mainloop_kernel, dim3(WG), dim3(256), stream, (const void *)w, out, clk, REP);
    MST_CALIB_TRY(hipEventRecord(e1, (hipStream_t)stream));
    MST_CALIB_TRY(hipEventSynchronize(e1));
    float ms = 0.0f;
    MST_CALIB_TRY(hipEventElapsedTime(&ms, e0, e1));
    long long c[2] = {0, 0};
    MST_CALIB_TRY(hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost));
#undef MST_CALIB_TRY
    *ms_per_launch = ms / (float)timed;
    *sclk_mhz = c[1] > 0 ? (float)((double)c[0] / ((double)c[1] / 100.0)) : 0.0f;      // shader clocks per microsecond
    cleanup();
    return MST_OK;
}

extern "C" size_t mst_tcn_workspace_bytes(const MstTcn *t, int B, int L, int precision) {
    if (B < 1 || L < 1) return 0;
    if (t && t->generic) return 2 * align_up((size_t)B * L * t->d.channels * sizeof(float), 256);
    return 2 * align_up((size_t)B * L * 128 * tcn_elem(precision), 256);
}

extern "C" int mst_tcn_forward(MstTcn *t, const float *x, float *y, int B, int L, int precision, void *ws,
                               size_t ws_bytes, void *stream) {
    if (!y) return fail(MST_ERR_ARG, "mst_tcn_forward: null output");
    return tcn_run(t, x, y, nullptr, B, L, precision, t ? t->d.nblocks : 0, ws, ws_bytes, stream);
}

extern "C" int mst_tcn_forward_blocks(MstTcn *t, const float *x, float *act, int B, int L, int precision, int n_run,
                                      void *ws, size_t ws_bytes, void *stream) {
    if (!t || !act || n_run < 1 || n_run > t->d.nblocks) return fail(MST_ERR_ARG, "mst_tcn_forward_blocks: bad argument");
    return tcn_run(t, x, nullptr, act, B, L, precision, n_run, ws, ws_bytes, stream);
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

extern "C" int mst_film_forward(const float *w, const float *b, const float *cond, int rows, int cond_dim, int C, const float *x,
                                float *y, int B, long L, float *table, void *stream) {
    if (!w || !b || !cond || !x || !y || !table || rows < 1 || cond_dim < 1 || C < 1 || B < 1 || L < 1)
        return fail(MST_ERR_ARG, "mst_film_forward: bad argument");
    if (rows != 1 && rows != B) return fail(MST_ERR_ARG, "mst_film_forward: condition rows must be 1 or equal the batch size");
    FilmArgs a;
    a.fw = w;
    a.fb = b;
    a.cond = cond;
    a.film = table;
    a.nblocks = 1;
    a.two_c = 2 * C;
    a.D = cond_dim;
    a.rows = rows;
    a.block_stride = 0;
    MST_LAUNCH(tcn_film_kernel, dim3((2 * C + 3) / 4), dim3(256), stream, a);
    MST_CHECK_LAUNCH("tcn_film_kernel");
    const long total = (long)B * C * L;
    MST_LAUNCH(film_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), stream, x, y, (const float *)table, rows, C, L, total);
    MST_CHECK_LAUNCH("film_apply_kernel");
    return MST_OK;
}

extern "C" int mst_embedding_mean(const float *emb, int n_rows, int dim, float *out, void *stream) {
    if (!emb || !out || n_rows < 1 || dim < 1) return fail(MST_ERR_ARG, "mst_embedding_mean: bad argument");
    MST_LAUNCH(embedding_mean_kernel, dim3((dim + 255) / 256), dim3(256), stream, emb, n_rows, dim, out);
    MST_CHECK_LAUNCH("embedding_mean_kernel");
    return MST_OK;
}

// =================================================================================================
// FX processors
// =================================================================================================
namespace {
int g_fx_eq_valu_ends = 0;      // mst_fx_set_tuning bit 5: stereo equaliser state pass on float64 VALU dot products with the table in LDS (the reference form of the MFMA kernel)
int g_fx_eq_lane_apply = 0;     // mst_fx_set_tuning bit 4: stereo equaliser apply pass one lane per chunk straight from global memory (the reference form of the slab kernel)
// Steps per chunk of the time-parallel biquad cascade: the number of 511-chunk scan blocks that minimises
//     16 us per scan block + two chunk passes at 0.19 us per step of a chunk
// (measured on an MI355X; a pass is one lane per chunk and its time falls with the chunk length until every SIMD holds a wave = 65536
// lanes, after which it is the total work that counts: 144-step chunks at 116 k lanes take the same 105 us as 272-step chunks at 62 k).
// A 131072-sample segment x 128 sequences: 1 block, 272 steps, 482 chunks; a 3-minute stem x 2: ~16 blocks, ~976 steps.
int biquad_chunk(long L, long n_seq) {
    auto up16 = [](long v) { return (int)((v + 15) / 16 * 16); };
    int best_m = 64;
    double best = 1e300;
    for (int B = 1; B <= 64; ++B) {
        int m = up16((L + 511L * B - 1) / (511L * B));
        if (m < 64) m = 64;
        const long nchunks = (L + m - 1) / m;
        const double lanes = (double)n_seq * (double)nchunks;
        const double cost = 16.0 * (double)((nchunks + 510) / 511) + 0.39 * m * (lanes > 65536.0 ? lanes / 65536.0 : 1.0);
        if (cost < best) {
            best = cost;
            best_m = m;
        }
        if (m == 64) break;
    }
    return best_m;
}
void biquad_coefs(const double *coef, int n_bands, double (*out)[5]) {
    for (int k = 0; k < MST_MAX_BANDS; ++k)
        for (int i = 0; i < 5; ++i) out[k][i] = 0.0;
    for (int k = 0; k < n_bands; ++k) {
        const double a0 = coef[6 * k + 3];
        out[k][0] = coef[6 * k + 0] / a0;
        out[k][1] = coef[6 * k + 1] / a0;
        out[k][2] = coef[6 * k + 2] / a0;
        out[k][3] = coef[6 * k + 4] / a0;
        out[k][4] = coef[6 * k + 5] / a0;
    }
}
}  // namespace

namespace {
// Impulse-state table of a biquad cascade for fx_biquad_ends_kernel: h_m = the cascade's state m steps after a unit impulse, m = 0 .. M - 1,
// [M][2 * n_bands] float64.  Built on the host (the kernels' own recursion) and kept on the device per (device, coefficients, M): a chain calls
// its equaliser with the same settings again and again.  An entry owns its host copy (the asynchronous upload reads it) and its device
// buffer; the 16 most recent entries per device are kept, evicting one waits for the device (rare: randomised parameter sweeps).
struct BiquadTab {
    int dev = -1, n_bands = 0, M = 0;
    double coef[MST_MAX_BANDS][5];
    std::vector<double> host;
    double *devp = nullptr;
    unsigned long stamp = 0;
};
// (the same entry carries the scan's matrix powers (A^M)^(2^l), l = 0 .. MST_BIQUAD_LEVELS - 1, behind the table: [M][S] | [levels][S][S];
//  round 4 squared them up on the device with a one-workgroup launch per call - 5-8 us on the chain's critical path)
const double *biquad_impulse_table(const double (*coef)[5], int n_bands, int M, void *stream) {
    static std::mutex mu;
    static std::vector<BiquadTab *> tabs;
    static unsigned long clock_ = 0;
    const int dev = mst_current_device();
    std::lock_guard<std::mutex> lock(mu);
    BiquadTab *oldest = nullptr;
    int n_dev = 0;
    for (BiquadTab *t : tabs) {
        if (t->dev != dev) continue;
        ++n_dev;
        if (t->n_bands == n_bands && t->M == M && std::memcmp(t->coef, coef, sizeof(double) * 5 * n_bands) == 0) {
            t->stamp = ++clock_;
            return t->devp;
        }
        if (!oldest || t->stamp < oldest->stamp) oldest = t;
    }
    const int S = 2 * n_bands;
    BiquadTab *t = nullptr;
    if (n_dev >= 16) {
        t = oldest;
        if (hipDeviceSynchronize() != hipSuccess) return nullptr;      // nobody reads the evicted table any more
    } else {
        t = new BiquadTab;
        tabs.push_back(t);
    }
    t->dev = dev;
    t->n_bands = n_bands;
    t->M = M;
    std::memset(t->coef, 0, sizeof(t->coef));
    std::memcpy(t->coef, coef, sizeof(double) * 5 * n_bands);
    t->stamp = ++clock_;
    // [M][S] table | [levels][S][S] powers | [M / 16][4][64] the table as A fragments of v_mfma_f64_16x16x4_f64 (state rows, 16 samples per slab)
    const size_t n_tab = (size_t)M * S, n_pow = (size_t)MST_BIQUAD_LEVELS * S * S, n_frag = (size_t)((M + 15) / 16) * 4 * 64, n_all = n_tab + n_pow + n_frag;
    if (t->host.size() < n_all) {
        if (t->devp) (void)hipFree(t->devp);
        t->devp = nullptr;
        t->host.assign(n_all, 0.0);
    }
    std::vector<double> z(S, 0.0);
    for (int m = 0; m < M; ++m) {
        double v = m == 0 ? 1.0 : 0.0;
        for (int b = 0; b < n_bands; ++b) v = fx_biquad_band(v, z[2 * b], z[2 * b + 1], coef[b]);
        for (int j = 0; j < S; ++j) t->host[(size_t)m * S + j] = z[j];
    }
    {   // A^M column by column (the cascade run M steps on zero input from each unit state), then squared up level by level
        double *pm = t->host.data() + n_tab;
        for (int col = 0; col < S; ++col) {
            std::vector<double> u(S, 0.0);
            u[col] = 1.0;
            for (int n = 0; n < M; ++n) {
                double v = 0.0;
                for (int b = 0; b < n_bands; ++b) v = fx_biquad_band(v, u[2 * b], u[2 * b + 1], coef[b]);
            }
            for (int row = 0; row < S; ++row) pm[(size_t)row * S + col] = u[row];
        }
        for (int l = 1; l < MST_BIQUAD_LEVELS; ++l) {
            const double *cur = pm + (size_t)(l - 1) * S * S;
            double *nxt = pm + (size_t)l * S * S;
            for (int r = 0; r < S; ++r)
                for (int c = 0; c < S; ++c) {
                    double acc = 0.0;
                    for (int j = 0; j < S; ++j) acc += cur[r * S + j] * cur[j * S + c];
                    nxt[r * S + c] = acc;
                }
        }
    }
    {   // fragment (slab sb, k-step kk), lane (state j = l & 15, kq = l >> 4): the weight of sample 16 sb + 4 kk + kq in the end state, h_(M - 1 - sample)[j]
        double *fr = t->host.data() + n_tab + n_pow;
        for (size_t i = 0; i < n_frag; ++i) {
            const int l = (int)(i % 64), kk = (int)(i / 64 % 4), sb = (int)(i / 256), j = l & 15, smp = 16 * sb + 4 * kk + (l >> 4);
            fr[i] = (j < S && smp < M) ? t->host[(size_t)(M - 1 - smp) * S + j] : 0.0;
        }
    }
    if (!t->devp && hipMalloc((void **)&t->devp, t->host.size() * sizeof(double)) != hipSuccess) {
        t->dev = -1;
        t->devp = nullptr;
        return nullptr;
    }
    if (hipMemcpyAsync(t->devp, t->host.data(), n_all * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
        t->dev = -1;
        return nullptr;
    }
    return t->devp;
}
}  // namespace

extern "C" size_t mst_fx_biquad_scratch_bytes(int n_items, long L, int C, int n_bands) {
    if (n_items < 1 || L < 1 || C < 1 || n_bands < 1) return 0;
    const int M = biquad_chunk(L, (long)n_items * C);
    const long nchunks = (L + M - 1) / M;
    const size_t states = (size_t)n_items * C * nchunks * 2 * MST_MAX_BANDS;
    return (2 * states + (size_t)MST_BIQUAD_LEVELS * 4 * MST_MAX_BANDS * MST_MAX_BANDS) * sizeof(double);      // ends | starts | (A^M)^(2^l)
}

extern "C" int mst_fx_biquad_cascade(const float *x, float *y, int n_items, long L, int C, const double *coef, int n_bands,
                                     double *scratch, size_t scratch_bytes, const MstFxFuse *fuse, void *stream) {
    if (!x || !y || !coef || n_items < 1 || L < 1 || C < 1) return fail(MST_ERR_ARG, "mst_fx_biquad_cascade: bad argument");
    if (fuse && fuse->post_rms) return fail(MST_ERR_UNSUPPORTED, "mst_fx_biquad_cascade: tail folding (post_rms) is the imager's");
    if (fuse && (fuse->out_ms_dev || fuse->in_ms_dev)) return fail(MST_ERR_UNSUPPORTED, "mst_fx_biquad_cascade: the mid / side energies travel from the compressor to the imager");
    if (n_bands < 0 || n_bands > MST_MAX_BANDS) return fail(MST_ERR_UNSUPPORTED, "mst_fx_biquad_cascade: at most 8 bands");
    const int M = biquad_chunk(L, (long)n_items * C);
    const long nchunks = (L + M - 1) / M;
    if (scratch && nchunks > 1 && n_bands > 0) {
        if (scratch_bytes < mst_fx_biquad_scratch_bytes(n_items, L, C, n_bands))
            return fail(MST_ERR_WORKSPACE, "mst_fx_biquad_cascade: scratch too small");
        BiquadChunkArgs a;
        a.x = x;
        a.y = y;
        a.n_seq = n_items * C;
        a.C = C;
        a.nchunks = (int)nchunks;
        a.M = M;
        a.L = L;
        a.n_bands = n_bands;
        a.in_scale = fuse ? fuse->in_scale_dev : nullptr;
        a.out_sumsq = fuse ? fuse->out_sumsq_dev : nullptr;
        a.out_in_sumsq = fuse ? fuse->out_in_sumsq_dev : nullptr;
        biquad_coefs(coef, n_bands, a.coef);
        const size_t states = (size_t)a.n_seq * nchunks * 2 * MST_MAX_BANDS;
        double *ends = scratch, *starts = scratch + states;
        a.ends = ends;
        a.starts = starts;
        const int S = 2 * n_bands;
        const long lanes = (long)a.n_seq * nchunks;
        const dim3 cg((unsigned)((lanes + 63) / 64));
        auto launch_chunks = [&](auto APPLY) {         // the band count is a template parameter: no per-band branches in the recursion
            constexpr bool ap = decltype(APPLY)::value;
            switch (n_bands) {
                case 1: MST_LAUNCH((fx_biquad_chunk_kernel<ap, 1>), cg, dim3(64), stream, a); break;
                case 2: MST_LAUNCH((fx_biquad_chunk_kernel<ap, 2>), cg, dim3(64), stream, a); break;
                case 3: MST_LAUNCH((fx_biquad_chunk_kernel<ap, 3>), cg, dim3(64), stream, a); break;
                case 4: MST_LAUNCH((fx_biquad_chunk_kernel<ap, 4>), cg, dim3(64), stream, a); break;
                case 5: MST_LAUNCH((fx_biquad_chunk_kernel<ap, 5>), cg, dim3(64), stream, a); break;
                case 6: MST_LAUNCH((fx_biquad_chunk_kernel<ap, 6>), cg, dim3(64), stream, a); break;
                case 7: MST_LAUNCH((fx_biquad_chunk_kernel<ap, 7>), cg, dim3(64), stream, a); break;
                default: MST_LAUNCH((fx_biquad_chunk_kernel<ap, 8>), cg, dim3(64), stream, a); break;
            }
        };
        // pass 1: zero-state end states as dot products with the cascade's impulse-state table (fx_biquad_ends_kernel)
        const double *htab = biquad_impulse_table(a.coef, n_bands, M, stream);
        if (!htab) return fail(MST_ERR_HIP, "mst_fx_biquad_cascade: impulse-state table");
        const long npairs = (long)n_items * nchunks;                       // stereo: (item, chunk) pairs - 32 per wave, slabs through LDS
        if (C == 2 && !g_fx_eq_valu_ends && M % 16 == 0) {          // stereo: the end states as a matrix product on the float64 matrix cores
            const dim3 eg((unsigned)((npairs + 127) / 128));
            MST_LAUNCH(fx_biquad_stereo_ends_mfma_kernel, eg, dim3(256), stream, a, htab + (size_t)M * S + (size_t)MST_BIQUAD_LEVELS * S * S);
        } else if (C == 2) {
            const dim3 eg((unsigned)((npairs + 127) / 128));
            switch (n_bands) {
                case 1: MST_LAUNCH(fx_biquad_stereo_ends_kernel<1>, eg, dim3(256), stream, a, htab); break;
                case 2: MST_LAUNCH(fx_biquad_stereo_ends_kernel<2>, eg, dim3(256), stream, a, htab); break;
                case 3: MST_LAUNCH(fx_biquad_stereo_ends_kernel<3>, eg, dim3(256), stream, a, htab); break;
                case 4: MST_LAUNCH(fx_biquad_stereo_ends_kernel<4>, eg, dim3(256), stream, a, htab); break;
                case 5: MST_LAUNCH(fx_biquad_stereo_ends_kernel<5>, eg, dim3(256), stream, a, htab); break;
                case 6: MST_LAUNCH(fx_biquad_stereo_ends_kernel<6>, eg, dim3(256), stream, a, htab); break;
                case 7: MST_LAUNCH(fx_biquad_stereo_ends_kernel<7>, eg, dim3(256), stream, a, htab); break;
                default: MST_LAUNCH(fx_biquad_stereo_ends_kernel<8>, eg, dim3(256), stream, a, htab); break;
            }
        } else {
            const dim3 eg((unsigned)((4 * lanes + 255) / 256));          // four lanes per chunk
            switch (n_bands) {
                case 1: MST_LAUNCH(fx_biquad_ends_kernel<1>, eg, dim3(256), stream, a, htab); break;
                case 2: MST_LAUNCH(fx_biquad_ends_kernel<2>, eg, dim3(256), stream, a, htab); break;
                case 3: MST_LAUNCH(fx_biquad_ends_kernel<3>, eg, dim3(256), stream, a, htab); break;
                case 4: MST_LAUNCH(fx_biquad_ends_kernel<4>, eg, dim3(256), stream, a, htab); break;
                case 5: MST_LAUNCH(fx_biquad_ends_kernel<5>, eg, dim3(256), stream, a, htab); break;
                case 6: MST_LAUNCH(fx_biquad_ends_kernel<6>, eg, dim3(256), stream, a, htab); break;
                case 7: MST_LAUNCH(fx_biquad_ends_kernel<7>, eg, dim3(256), stream, a, htab); break;
                default: MST_LAUNCH(fx_biquad_ends_kernel<8>, eg, dim3(256), stream, a, htab); break;
            }
        }
        MST_CHECK_LAUNCH("fx_biquad_ends_kernel");
        const dim3 sg((unsigned)a.n_seq);
        const double *pmat = htab + (size_t)M * S;          // (A^M)^(2^l), l = 0 .. 8: behind the impulse-state table (cached per coefficient set)
        auto launch_scan = [&](auto NBv) {
            constexpr int nb = decltype(NBv)::value;
            const double *e = ends, *pmc = pmat;
            switch (n_bands) {
                case 1: MST_LAUNCH((fx_biquad_scan_kernel<1, nb>), sg, dim3(nb), stream, e, starts, pmc, a.n_seq, (int)nchunks); break;
                case 2: MST_LAUNCH((fx_biquad_scan_kernel<2, nb>), sg, dim3(nb), stream, e, starts, pmc, a.n_seq, (int)nchunks); break;
                case 3: MST_LAUNCH((fx_biquad_scan_kernel<3, nb>), sg, dim3(nb), stream, e, starts, pmc, a.n_seq, (int)nchunks); break;
                case 4: MST_LAUNCH((fx_biquad_scan_kernel<4, nb>), sg, dim3(nb), stream, e, starts, pmc, a.n_seq, (int)nchunks); break;
                case 5: MST_LAUNCH((fx_biquad_scan_kernel<5, nb>), sg, dim3(nb), stream, e, starts, pmc, a.n_seq, (int)nchunks); break;
                case 6: MST_LAUNCH((fx_biquad_scan_kernel<6, nb>), sg, dim3(nb), stream, e, starts, pmc, a.n_seq, (int)nchunks); break;
                case 7: MST_LAUNCH((fx_biquad_scan_kernel<7, nb>), sg, dim3(nb), stream, e, starts, pmc, a.n_seq, (int)nchunks); break;
                default: MST_LAUNCH((fx_biquad_scan_kernel<8, nb>), sg, dim3(nb), stream, e, starts, pmc, a.n_seq, (int)nchunks); break;
            }
        };
        if (nchunks > 255) launch_scan(std::integral_constant<int, 512>{});
        else launch_scan(std::integral_constant<int, 256>{});
        MST_CHECK_LAUNCH("fx_biquad_scan_kernel");
        if (C == 2 && !g_fx_eq_lane_apply) {          // stereo: the chunks travel in 16-frame slabs through LDS, in and out
            const dim3 ag((unsigned)((npairs + 127) / 128));
            switch (n_bands) {
                case 1: MST_LAUNCH(fx_biquad_stereo_apply_kernel<1>, ag, dim3(256), stream, a); break;
                case 2: MST_LAUNCH(fx_biquad_stereo_apply_kernel<2>, ag, dim3(256), stream, a); break;
                case 3: MST_LAUNCH(fx_biquad_stereo_apply_kernel<3>, ag, dim3(256), stream, a); break;
                case 4: MST_LAUNCH(fx_biquad_stereo_apply_kernel<4>, ag, dim3(256), stream, a); break;
                case 5: MST_LAUNCH(fx_biquad_stereo_apply_kernel<5>, ag, dim3(256), stream, a); break;
                case 6: MST_LAUNCH(fx_biquad_stereo_apply_kernel<6>, ag, dim3(256), stream, a); break;
                case 7: MST_LAUNCH(fx_biquad_stereo_apply_kernel<7>, ag, dim3(256), stream, a); break;
                default: MST_LAUNCH(fx_biquad_stereo_apply_kernel<8>, ag, dim3(256), stream, a); break;
            }
        } else {
            launch_chunks(std::true_type{});
        }
        MST_CHECK_LAUNCH("fx_biquad_chunk_kernel<apply>");
        return MST_OK;
    }
    if (fuse && (fuse->in_scale_dev || fuse->out_sumsq_dev || fuse->out_in_sumsq_dev))
        return fail(MST_ERR_UNSUPPORTED, "mst_fx_biquad_cascade: chain fusion needs the time-parallel path (scratch, more than one chunk, >= 1 band)");
    BiquadArgs a;
    a.x = x;
    a.y = y;
    a.n_seq = n_items * C;
    a.C = C;
    a.L = L;
    a.n_bands = n_bands;
    biquad_coefs(coef, n_bands, a.coef);
    MST_LAUNCH(fx_biquad_kernel, dim3((a.n_seq + 63) / 64), dim3(64), stream, a);
    MST_CHECK_LAUNCH("fx_biquad_kernel");
    return MST_OK;
}

namespace {
// scratch = level differences [L][n_seq] (serial fallback only) | chunk maps [n_seq][nchunks][NP + 1] | chunk start values
// [nchunks][n_seq] | log10 table [256]
struct CompScratch { size_t xl, maps, ystart, tab, carry, total; long nchunks; };
CompScratch comp_scratch(int n_items, long L, int C) {
    CompScratch c;
    const size_t n_seq = (size_t)n_items * C;
    c.nchunks = (L + MST_COMP_T - 1) / MST_COMP_T;
    c.xl = c.nchunks < 4 ? n_seq * (size_t)L * sizeof(double) : 0;      // only the serial form of very short signals stores them
    c.maps = n_seq * (size_t)c.nchunks * MST_COMP_REC * sizeof(double);
    c.ystart = n_seq * (size_t)c.nchunks * sizeof(double);
    c.tab = 256 * sizeof(double);
    c.carry = n_seq * sizeof(double);                                      // the smoother's value between two time slices of the chain
    c.total = c.xl + c.maps + c.ystart + c.tab + c.carry;
    return c;
}
}  // namespace

extern "C" size_t mst_fx_compressor_scratch_bytes(int n_items, long L, int C) {
    if (n_items < 1 || L < 1 || C < 1) return 0;
    return comp_scratch(n_items, L, C).total;
}

namespace {
// fx_log10_table_kernel's 256 doubles, one copy per device, made by the first compressor call there (kept for the life of the process)
const double *log10_table(void *stream) {
    static std::mutex mu;
    static double *tabs[64] = {};
    const int dev = mst_current_device();
    if (dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!tabs[dev]) {
        double *t = nullptr;
        if (hipMalloc((void **)&t, 256 * sizeof(double)) != hipSuccess) return nullptr;
        MST_LAUNCH(fx_log10_table_kernel, dim3(1), dim3(128), stream, t);
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
            (void)hipFree(t);
            return nullptr;
        }
        tabs[dev] = t;
    }
    return tabs[dev];
}

// the side stream of the time-parallel FX kernels (compressor_run): one per device, non-blocking, lowest priority, with the events of one
// fork / join; kept for the life of the process
struct FxSide {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, map_done[8] = {}, chain_done[8] = {};
    std::mutex mu;
};
int g_fx_pipeline = 1;          // mst_fx_set_tuning bit 0
int g_fx_pipeline_any_size = 0; // mst_fx_set_tuning bit 1 (test / A-B hook: slices whatever the size of the batch)
int g_fx_slices = 3;            // mst_fx_set_tuning bits 2-3: 0 -> 3 slices (default: measured 0.523-0.539 ms per chain against 0.540-0.550 with 4), 1 -> 2, 2 -> 4, 3 -> 8
FxSide *fx_side() {
    static std::mutex mu;
    static FxSide *sides[64] = {};
    const int dev = mst_current_device();
    if (dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!sides[dev]) {
        FxSide *f = new FxSide;
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);          // lo = the numerically greatest = lowest priority
        bool ok = hipStreamCreateWithPriority(&f->stream, hipStreamNonBlocking, lo) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&f->fork, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&f->join, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 8 && ok; ++i)
            ok = hipEventCreateWithFlags(&f->map_done[i], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&f->chain_done[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            delete f;
            return nullptr;
        }
        sides[dev] = f;
    }
    return sides[dev];
}

int compressor_run(CompArgs a, int n_items, long L, int C, double *scratch, size_t scratch_bytes, void *stream) {
    if (scratch) {
        if (scratch_bytes < mst_fx_compressor_scratch_bytes(n_items, L, C))
            return fail(MST_ERR_WORKSPACE, "mst_fx_compressor: scratch too small");
        const dim3 tiles((unsigned)((L + 63) / 64), (unsigned)((a.n_seq + 63) / 64));      // 64 x 64 (time x sequence) tiles
        const CompScratch cs = comp_scratch(n_items, L, C);
        if (cs.nchunks < 4) {
            if (a.out_sumsq) MST_HIP_TRY(hipMemsetAsync(a.out_sumsq, 0, (size_t)n_items * MST_SUMSQ_SLOTS * sizeof(double), (hipStream_t)stream));
            if (a.out_ms) MST_HIP_TRY(hipMemsetAsync(a.out_ms, 0, (size_t)n_items * MST_SUMSQ_SLOTS * 2 * sizeof(double), (hipStream_t)stream));
            MST_LAUNCH(fx_comp_gain_kernel, tiles, dim3(256), stream, a, scratch);
            MST_CHECK_LAUNCH("fx_comp_gain_kernel");
            MST_LAUNCH(fx_comp_smooth_kernel, dim3((a.n_seq + 63) / 64), dim3(64), stream, a, scratch);
            MST_CHECK_LAUNCH("fx_comp_smooth_kernel");
        } else {       // the smoother parallel in time: chunk maps (convex piecewise-linear), a chain over chunks, the rest in one pass
            CompMapArgs m;
            const double *tab = log10_table(stream);          // a constant of the device: built on first use
            if (!tab) return fail(MST_ERR_HIP, "mst_fx_compressor: log10 table");
            m.log_tab = tab;
            m.maps = (double *)((unsigned char *)scratch + cs.xl);
            m.ystart = (double *)((unsigned char *)scratch + cs.xl + cs.maps);
            m.n_seq = a.n_seq;
            m.nchunks = (int)cs.nchunks;
            m.L = L;
            m.aA = a.alpha_att;
            m.aR = a.alpha_rel;
            m.use_min = a.alpha_att > a.alpha_rel ? 1 : 0;
            // the piece at sorted position p of a chunk of n steps has been through n - p attack and p release steps
            const int n_last = (int)(L - (cs.nchunks - 1) * MST_COMP_T);
            for (int which = 0; which < 2; ++which) {
                const int n = which ? n_last : MST_COMP_T;
                for (int p = 0; p < MST_COMP_NP; ++p) {
                    m.slope[which][p] = p <= n ? std::pow(m.aA, n - p) * std::pow(m.aR, p) : 0.0;
                    m.inv_slope[which][p] = p <= n ? 1.0 / m.slope[which][p] : 0.0;
                }
            }
            m.ycarry = (double *)((unsigned char *)scratch + cs.xl + cs.maps + cs.ystart + cs.tab);
            // Time slices.  The chain is ONE dependent walk per sequence (n_seq workgroups, latency-bound: most of the chip idles beside it) while
            // the map and apply kernels are throughput work.  The signal is cut into NS (three) slices of whole chain batches; the caller's stream runs
            // the chain of slice 0, 1, ... back to back, a side stream (lower priority) the maps of slice 1, 2, ... and the applies of slice
            // 0 .. NS - 2 beside it (events order map_i -> chain_i -> apply_i); the last apply follows the last chain on the caller's stream, which
            // then waits for the side stream.  Same arithmetic, same results (the smoother's value crosses a slice boundary as a float64 in
            // ycarry); without concurrency (a profiler serialising the queues) the launches simply run one after the other.
            const int nbatch = (int)((cs.nchunks + MST_CHAIN_CB - 1) / MST_CHAIN_CB);
            const int gy = (a.n_seq + 63) / 64;
            int ns = 1;
            if (g_fx_pipeline && ((nbatch >= 32 && (double)a.n_seq * (double)L >= 4.0e6) || (g_fx_pipeline_any_size && nbatch >= 8))) ns = g_fx_slices;
            FxSide *side = ns > 1 ? fx_side() : nullptr;
            if (!side) ns = 1;
            auto launch_map = [&](int b0, int b1, void *st) -> int {
                CompMapArgs mm = m;
                mm.chunk0 = b0 * MST_CHAIN_CB;
                mm.clear_sumsq = b0 == 0 ? 1 : 0;
                const long c1 = std::min<long>((long)b1 * MST_CHAIN_CB, cs.nchunks);
                const dim3 cg((unsigned)(c1 - mm.chunk0), (unsigned)gy);
                if (m.use_min) MST_LAUNCH(fx_comp_map_kernel<true>, cg, dim3(64), st, mm, a);
                else MST_LAUNCH(fx_comp_map_kernel<false>, cg, dim3(64), st, mm, a);
                MST_CHECK_LAUNCH("fx_comp_map_kernel");
                return MST_OK;
            };
            auto launch_chain = [&](int b0, int b1, void *st) -> int {
                CompMapArgs mm = m;
                mm.batch0 = b0;
                mm.batch1 = b1;
                MST_LAUNCH(fx_comp_chain_kernel, dim3(a.n_seq), dim3(MST_CHAIN_THREADS), st, mm);
                MST_CHECK_LAUNCH("fx_comp_chain_kernel");
                return MST_OK;
            };
            auto launch_apply = [&](int b0, int b1, void *st) -> int {      // a batch is 32 chunks = 16 time tiles of 64 samples
                const long t0 = (long)b0 * (MST_CHAIN_CB / 2), t1 = std::min<long>((long)b1 * (MST_CHAIN_CB / 2), (long)tiles.x);
                MST_LAUNCH((fx_comp_apply_kernel<true>), dim3((unsigned)(t1 - t0), tiles.y), dim3(256), st, a, tab, (const double *)m.ystart, m.nchunks, (int)t0);
                MST_CHECK_LAUNCH("fx_comp_apply_kernel");
                return MST_OK;
            };
            static_assert(MST_COMP_T == 32 && MST_CHAIN_CB % 2 == 0, "two chunks per 64-sample apply tile");
            int rc;
            if (ns == 1) {
                if ((rc = launch_map(0, nbatch, stream)) || (rc = launch_chain(0, nbatch, stream)) || (rc = launch_apply(0, nbatch, stream))) return rc;
                return MST_OK;
            }
            std::lock_guard<std::mutex> lock(side->mu);          // one fork / join at a time per device: the events are reused
            hipStream_t main_s = (hipStream_t)stream, side_s = side->stream;
            auto bound = [&](int i) { return (int)((long)nbatch * i / ns); };
            if ((rc = launch_map(0, bound(1), stream))) return rc;          // slice 0's map: nothing to overlap it with
            MST_HIP_TRY(hipEventRecord(side->fork, main_s));               // the side stream sees the input (and slice 0's cleared energy slots)
            MST_HIP_TRY(hipStreamWaitEvent(side_s, side->fork, 0));
            for (int i = 1; i < ns; ++i) {
                if ((rc = launch_map(bound(i), bound(i + 1), side_s))) return rc;
                MST_HIP_TRY(hipEventRecord(side->map_done[i], side_s));
            }
            for (int i = 0; i < ns; ++i) {
                if (i > 0) MST_HIP_TRY(hipStreamWaitEvent(main_s, side->map_done[i], 0));
                if ((rc = launch_chain(bound(i), bound(i + 1), stream))) return rc;
                if (i + 1 < ns) {
                    MST_HIP_TRY(hipEventRecord(side->chain_done[i], main_s));
                    MST_HIP_TRY(hipStreamWaitEvent(side_s, side->chain_done[i], 0));
                    if ((rc = launch_apply(bound(i), bound(i + 1), side_s))) return rc;
                }
            }
            if ((rc = launch_apply(bound(ns - 1), nbatch, stream))) return rc;
            MST_HIP_TRY(hipEventRecord(side->join, side_s));
            MST_HIP_TRY(hipStreamWaitEvent(main_s, side->join, 0));
            return MST_OK;
        }
        MST_LAUNCH((fx_comp_apply_kernel<false>), tiles, dim3(256), stream, a, (const double *)scratch, (const double *)nullptr, 0, 0);
        MST_CHECK_LAUNCH("fx_comp_apply_kernel");
        return MST_OK;
    }
    MST_LAUNCH(fx_compressor_kernel, dim3((a.n_seq + 3) / 4), dim3(256), stream, a);
    MST_CHECK_LAUNCH("fx_compressor_kernel");
    return MST_OK;
}
}  // namespace

extern "C" int mst_fx_set_tuning(int flags) {
    if (flags < 0 || flags > 63) return fail(MST_ERR_ARG, "mst_fx_set_tuning: unknown flag bits");
    g_fx_eq_lane_apply = (flags >> 4) & 1;
    g_fx_eq_valu_ends = (flags >> 5) & 1;
    g_fx_pipeline = flags & 1;
    g_fx_pipeline_any_size = (flags >> 1) & 1;
    static const int slices[4] = {3, 2, 4, 8};
    g_fx_slices = slices[(flags >> 2) & 3];
    return MST_OK;
}

extern "C" int mst_fx_compressor(const float *x, float *y, int n_items, long L, int C, double threshold_db,
                                 double attack_ms, double release_ms, double ratio, double sample_rate, double *scratch,
                                 size_t scratch_bytes, const MstFxFuse *fuse, void *stream) {
    if (!x || !y || n_items < 1 || L < 1 || C < 1 || attack_ms <= 0 || release_ms <= 0 || ratio <= 0 || sample_rate <= 0)
        return fail(MST_ERR_ARG, "mst_fx_compressor: bad argument");
    const bool fused = fuse && (fuse->in_scale_dev || fuse->out_sumsq_dev);
    if (fuse && (fuse->post_rms || fuse->out_in_sumsq_dev))
        return fail(MST_ERR_UNSUPPORTED, "mst_fx_compressor: tail folding (post_rms) is the imager's, out_in_sumsq_dev the equaliser's");
    if (fused && (!scratch || (threshold_db == 0.0 && ratio == 1.0)))
        return fail(MST_ERR_UNSUPPORTED, "mst_fx_compressor: chain fusion needs the scratch buffer and an active compressor");
    if (threshold_db == 0.0 && ratio == 1.0) {   // bypass (common_audioeffects.py:637)
        if (x != y) MST_HIP_TRY(hipMemcpyAsync(y, x, (size_t)n_items * L * C * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return MST_OK;
    }
    CompArgs a;
    a.x = x;
    a.y = y;
    a.n_seq = n_items * C;
    a.C = C;
    a.L = L;
    a.threshold = threshold_db;
    a.ratio = ratio;
    a.alpha_att = std::exp(-1.0 / (0.001 * sample_rate * attack_ms));
    a.alpha_rel = std::exp(-1.0 / (0.001 * sample_rate * release_ms));
    a.makeup = 0.0;
    a.in_scale = fuse ? fuse->in_scale_dev : nullptr;
    a.out_sumsq = fuse ? fuse->out_sumsq_dev : nullptr;
    if (fuse && fuse->in_ms_dev) return fail(MST_ERR_UNSUPPORTED, "mst_fx_compressor: in_ms_dev is the imager's");
    if (fuse && fuse->out_ms_dev) {
        if (C != 2 || !fuse->out_sumsq_dev) return fail(MST_ERR_UNSUPPORTED, "mst_fx_compressor: out_ms_dev needs stereo audio and out_sumsq_dev");
        a.out_ms = fuse->out_ms_dev;
    }
    return compressor_run(a, n_items, L, C, scratch, scratch_bytes, stream);
}

extern "C" int mst_fx_compressor_grid(const float *x, float *y, int n_items, long L, int C, const double *threshold_db_dev,
                                      const double *ratio_dev, double attack_ms, double release_ms, double sample_rate,
                                      double *scratch, size_t scratch_bytes, double *peak_dev, void *stream) {
    if (!x || !y || !threshold_db_dev || !ratio_dev || !scratch || n_items < 1 || L < 1 || C < 1 || attack_ms <= 0 ||
        release_ms <= 0 || sample_rate <= 0)
        return fail(MST_ERR_ARG, "mst_fx_compressor_grid: bad argument");
    CompArgs a;
    a.x = x;
    a.y = y;
    a.n_seq = n_items * C;
    a.C = C;
    a.L = L;
    a.threshold = 0.0;
    a.ratio = 1.0;
    a.thr_items = threshold_db_dev;
    a.ratio_items = ratio_dev;
    a.shared_x = 1;
    a.alpha_att = std::exp(-1.0 / (0.001 * sample_rate * attack_ms));
    a.alpha_rel = std::exp(-1.0 / (0.001 * sample_rate * release_ms));
    a.makeup = 0.0;
    int rc;
    if ((rc = compressor_run(a, n_items, L, C, scratch, scratch_bytes, stream))) return rc;
    if (peak_dev) {          // `compress` clips a candidate whose peak reaches 1 (utils_data_normalization.py:352-353)
        MST_LAUNCH(fx_item_peak_kernel, dim3(64, n_items), dim3(256), stream, (const float *)y, L * C, peak_dev);
        MST_CHECK_LAUNCH("fx_item_peak_kernel");
        MST_LAUNCH(fx_clip_if_kernel, dim3((unsigned)((L * C + 255) / 256), n_items), dim3(256), stream, y, L * C, (const double *)peak_dev);
        MST_CHECK_LAUNCH("fx_clip_if_kernel");
    }
    return MST_OK;
}

extern "C" int mst_fx_range_reduce(const float *x, long L, int C, int channel, const int *item_dev, const long *lo_dev,
                                   const long *hi_dev, int n_ranges, int mode, double *out_dev, void *stream) {
    if (!x || !item_dev || !lo_dev || !hi_dev || !out_dev || L < 1 || C < 1 || channel < 0 || channel >= C || n_ranges < 1 ||
        (mode != 0 && mode != 1))
        return fail(MST_ERR_ARG, "mst_fx_range_reduce: bad argument");
    MST_LAUNCH(fx_range_reduce_kernel, dim3(n_ranges), dim3(256), stream, x, L, C, channel, item_dev, lo_dev, hi_dev, mode, out_dev);
    MST_CHECK_LAUNCH("fx_range_reduce_kernel");
    return MST_OK;
}

extern "C" int mst_fx_onset_hfc(const float *x, int n_items, long L, int C, int channel, int win, float *out_dev, void *stream) {
    if (!x || !out_dev || n_items < 1 || L < 1 || C < 1 || channel < 0 || channel >= C)
        return fail(MST_ERR_ARG, "mst_fx_onset_hfc: bad argument");
    if (win != 256 && win != 512 && win != 1024 && win != 2048)
        return fail(MST_ERR_UNSUPPORTED, "mst_fx_onset_hfc: window must be 256, 512, 1024 or 2048 samples");
    const long n_frames = L / win;          // whole frames only (librosa.util.frame)
    if (n_frames < 1) return MST_OK;
    const dim3 grid((unsigned)(n_frames * n_items));
    switch (win) {
        case 256: MST_LAUNCH((fx_onset_hfc_kernel<256>), grid, dim3(256), stream, x, L, C, channel, n_frames, (float2 *)out_dev); break;
        case 512: MST_LAUNCH((fx_onset_hfc_kernel<512>), grid, dim3(256), stream, x, L, C, channel, n_frames, (float2 *)out_dev); break;
        case 1024: MST_LAUNCH((fx_onset_hfc_kernel<1024>), grid, dim3(256), stream, x, L, C, channel, n_frames, (float2 *)out_dev); break;
        default: MST_LAUNCH((fx_onset_hfc_kernel<2048>), grid, dim3(256), stream, x, L, C, channel, n_frames, (float2 *)out_dev); break;
    }
    MST_CHECK_LAUNCH("fx_onset_hfc_kernel");
    return MST_OK;
}

namespace {
int energy(const float *x, double *acc, int n_items, long per_item, int mode, void *stream) {
    MST_HIP_TRY(hipMemsetAsync(acc, 0, (size_t)n_items * 2 * sizeof(double), (hipStream_t)stream));
    const long frames = mode == 1 ? per_item / 2 : per_item;
    int chunks = (int)std::min<long>(64, (frames + 8191) / 8192);
    if (chunks < 1) chunks = 1;
    MST_LAUNCH(fx_energy_kernel, dim3(n_items * chunks), dim3(256), stream, x, acc, per_item, mode, chunks);
    MST_CHECK_LAUNCH("fx_energy_kernel");
    return MST_OK;
}
}  // namespace

extern "C" int mst_fx_midside_imager(const float *x, float *y, int n_items, long L, double bal, double *scratch, const MstFxFuse *fuse,
                                     void *stream) {
    if (!x || !y || !scratch || n_items < 1 || L < 1) return fail(MST_ERR_ARG, "mst_fx_midside_imager: bad argument");
    const bool fold = fuse && fuse->post_rms;
    if (fuse && fuse->out_in_sumsq_dev) return fail(MST_ERR_UNSUPPORTED, "mst_fx_midside_imager: out_in_sumsq_dev is the equaliser's");
    if (fold && !fuse->in_sumsq_dev) return fail(MST_ERR_ARG, "mst_fx_midside_imager: post_rms needs in_sumsq_dev");
    if (fuse && fuse->out_ms_dev) return fail(MST_ERR_UNSUPPORTED, "mst_fx_midside_imager: out_ms_dev is the compressor's");
    int chunks = (int)std::min<long>(MST_SUMSQ_SLOTS, (L + 8191) / 8192);
    if (chunks < 1) chunks = 1;
    const double *parts = scratch;
    if (fuse && fuse->in_ms_dev) {          // the producer of x (the compressor's apply pass) left the mid / side energies behind: no energy pass
        parts = fuse->in_ms_dev;
        chunks = MST_SUMSQ_SLOTS;
    } else {
        MST_LAUNCH(fx_energy_parts_kernel, dim3(n_items * chunks), dim3(256), stream, x, scratch, L, chunks);
        MST_CHECK_LAUNCH("fx_energy_parts_kernel");
    }
    const double bal_r = std::round(bal * 1000.0) / 1000.0;   // round(bal, 3) (:980)
    MST_LAUNCH(fx_imager_apply_kernel, dim3((unsigned)((L + MST_IMAGER_FRAMES - 1) / MST_IMAGER_FRAMES), n_items), dim3(256), stream, x, y,
               parts, chunks, L, bal_r, fuse ? fuse->in_scale_dev : (const double *)nullptr,
               fuse ? fuse->out_sumsq_dev : (double *)nullptr, fold ? fuse->in_sumsq_dev : (const double *)nullptr,
               fold ? fuse->post_gain : 1.0f);
    MST_CHECK_LAUNCH("fx_imager_apply_kernel");
    return MST_OK;
}

extern "C" int mst_fx_gain(const float *x, float *y, int n_items, long L, int C, double gain_db, int invert, const MstFxFuse *fuse,
                           void *stream) {
    if (!x || !y || n_items < 1 || L < 1 || C < 1) return fail(MST_ERR_ARG, "mst_fx_gain: bad argument");
    if (fuse && (fuse->post_rms || fuse->out_in_sumsq_dev || fuse->out_ms_dev || fuse->in_ms_dev))
        return fail(MST_ERR_UNSUPPORTED, "mst_fx_gain: tail folding (post_rms) is the imager's, out_in_sumsq_dev the equaliser's, the mid / side energies the compressor's / imager's");
    double g = std::pow(10.0, gain_db / 20.0);
    if (invert) g = -g;
    const long per = L * C;
    MST_LAUNCH(fx_scale_kernel, dim3((unsigned)((per + 255) / 256), n_items), dim3(256), stream, x, y, per, (float)g,
               (const double *)nullptr, (const double *)nullptr, 0, per, fuse ? fuse->in_scale_dev : (const double *)nullptr);
    MST_CHECK_LAUNCH("fx_scale_kernel");
    return MST_OK;
}

extern "C" int mst_fx_haas(const float *x, float *y, int n_items, long L, int c_in, long delay, double feedback,
                           int wet_channel, void *stream) {
    if (!x || !y || n_items < 1 || L < 1) return fail(MST_ERR_ARG, "mst_fx_haas: bad argument");
    if (c_in != 1 && c_in != 2) return fail(MST_ERR_ARG, "mst_fx_haas: Haas effect only works with monaural or stereo audio");
    if (wet_channel != 0 && wet_channel != 1) return fail(MST_ERR_ARG, "mst_fx_haas: wet_channel must be 0 (left) or 1 (right)");
    if (x == y) return fail(MST_ERR_ARG, "mst_fx_haas: in-place operation is not supported (circular read)");
    long shift = delay % L;
    if (shift < 0) shift += L;
    MST_LAUNCH(fx_haas_kernel, dim3((unsigned)((L + 255) / 256), n_items), dim3(256), stream, x, y, L, c_in, shift,
               (float)feedback, wet_channel);
    MST_CHECK_LAUNCH("fx_haas_kernel");
    return MST_OK;
}

extern "C" int mst_fx_panner(const float *x, float *y, int n_items, long L, int c_in, float g0, float g1, void *stream) {
    if (!x || !y || n_items < 1 || L < 1) return fail(MST_ERR_ARG, "mst_fx_panner: bad argument");
    if (c_in != 1 && c_in != 2) return fail(MST_ERR_ARG, "mst_fx_panner: Panner only works with monaural or stereo audio");
    if (x == y && c_in != 2) return fail(MST_ERR_ARG, "mst_fx_panner: in-place needs a stereo input");
    MST_LAUNCH(fx_panner_kernel, dim3((unsigned)((L + 255) / 256), n_items), dim3(256), stream, x, y, L, c_in, g0, g1);
    MST_CHECK_LAUNCH("fx_panner_kernel");
    return MST_OK;
}

// ---- power-of-two real FFTs (csrc/fft_kernels.h): plans for the FFT convolution and the STFT ---------------------------------------
struct MstFftPlan {
    long n = 0, m = 0;              // transform length, m = n / 2 complex points
    int log2m = 0;
    int passes = 0;                 // Stockham passes: log2(m) / 2 of radix 4, one of radix 2 in front when log2(m) is odd
    float2 *tw_m = nullptr;         // exp(-2 pi i j / m), j < m / 2
    float2 *tw_n = nullptr;         // exp(-2 pi i k / n), k <= m / 2
};

namespace {
void fft_plan_destroy(MstFftPlan *p) {
    if (!p) return;
    (void)hipFree(p->tw_m);
    (void)hipFree(p->tw_n);
    delete p;
}
// n: a power of two >= 4.  The twiddle tables are written on the null stream and waited for: plans are made once.
constexpr long MST_FFT_MAX_N = 1L << 21;          // the four-step kernels' largest transform (fft_kernels.h: LOGMAX 10 columns x 10 rows of complex points)
int fft_plan_create(MstFftPlan **out, long n) {
    if (n < 4 || (n & (n - 1)) || n > MST_FFT_MAX_N) return fail(MST_ERR_UNSUPPORTED, "FFT length must be a power of two in 4 ... 2^21");
    auto *p = new MstFftPlan;
    p->n = n;
    p->m = n / 2;
    for (long v = p->m; v > 1; v >>= 1) p->log2m++;
    p->passes = p->log2m >= 8 ? 2 : p->log2m / 2 + p->log2m % 2;          // four-step form (two kernels) from 256 complex points on
    const long cm = std::max<long>(1, p->m / 2), cn = p->m / 2 + 1;
    if (hipMalloc((void **)&p->tw_m, (size_t)cm * sizeof(float2)) != hipSuccess || hipMalloc((void **)&p->tw_n, (size_t)cn * sizeof(float2)) != hipSuccess) {
        fft_plan_destroy(p);
        return fail(MST_ERR_HIP, "FFT plan: hipMalloc failed");
    }
    MST_LAUNCH(fft_twiddle_kernel, dim3((unsigned)((cm + 255) / 256)), dim3(256), nullptr, p->tw_m, p->m, cm);
    MST_LAUNCH(fft_twiddle_kernel, dim3((unsigned)((cn + 255) / 256)), dim3(256), nullptr, p->tw_n, p->n, cn);
    if (hipStreamSynchronize(nullptr) != hipSuccess) {
        fft_plan_destroy(p);
        return fail(MST_ERR_HIP, "FFT plan: twiddle kernels failed");
    }
    *out = p;
    return MST_OK;
}
// the size-m complex FFT of nb sequences, ping-pong between a (stride sa) and b (stride sb), starting in `a`; returns where the result is
int fft_passes(const MstFftPlan *p, float2 *a, long sa, float2 *b, long sb, int nb, int inverse, void *stream, float2 **res, long *sres) {
    float2 *src = a, *dst = b;
    long ss = sa, sd = sb;
    if (p->log2m >= 8) {          // four-step: columns (a -> b), rows (b -> a)
        const int l1 = (p->log2m + 1) / 2, l2 = p->log2m - l1;
        const dim3 ga((unsigned)((1L << l2) / 16), (unsigned)nb), gb((unsigned)((1L << l1) / 16), (unsigned)nb);
#define MST_FFT_STEP(KERN, GRID, LG)                                                                                                         \
    if ((LG) <= 6) MST_LAUNCH((KERN<6>), GRID, dim3(256), stream, (const float2 *)src, dst, (const float2 *)p->tw_m, p->m, l1, l2, ss, sd, inverse); \
    else if ((LG) <= 8) MST_LAUNCH((KERN<8>), GRID, dim3(256), stream, (const float2 *)src, dst, (const float2 *)p->tw_m, p->m, l1, l2, ss, sd, inverse); \
    else if ((LG) <= 9) MST_LAUNCH((KERN<9>), GRID, dim3(256), stream, (const float2 *)src, dst, (const float2 *)p->tw_m, p->m, l1, l2, ss, sd, inverse); \
    else MST_LAUNCH((KERN<10>), GRID, dim3(256), stream, (const float2 *)src, dst, (const float2 *)p->tw_m, p->m, l1, l2, ss, sd, inverse);
        MST_FFT_STEP(fft_cols_kernel, ga, l1)
        MST_CHECK_LAUNCH("fft_cols_kernel");
        std::swap(src, dst);
        std::swap(ss, sd);
        MST_FFT_STEP(fft_rows_kernel, gb, l2)
        MST_CHECK_LAUNCH("fft_rows_kernel");
#undef MST_FFT_STEP
        std::swap(src, dst);
        std::swap(ss, sd);
        *res = src;
        *sres = ss;
        return MST_OK;
    }
    long Ns = 1;
    if (p->log2m % 2) {
        MST_LAUNCH(fft_stockham2_kernel, dim3((unsigned)((p->m / 2 + 255) / 256), (unsigned)nb), dim3(256), stream, (const float2 *)src, dst,
                   (const float2 *)p->tw_m, p->m, Ns, ss, sd, inverse);
        MST_CHECK_LAUNCH("fft_stockham2_kernel");
        std::swap(src, dst);
        std::swap(ss, sd);
        Ns = 2;
    }
    for (; Ns < p->m; Ns <<= 2) {
        MST_LAUNCH(fft_stockham4_kernel, dim3((unsigned)((p->m / 4 + 255) / 256), (unsigned)nb), dim3(256), stream, (const float2 *)src, dst,
                   (const float2 *)p->tw_m, p->m, Ns, ss, sd, inverse);
        MST_CHECK_LAUNCH("fft_stockham4_kernel");
        std::swap(src, dst);
        std::swap(ss, sd);
    }
    *res = src;
    *sres = ss;
    return MST_OK;
}
// hipfftExecR2C's contract: in [nb][n] reals (DESTROYED: it is one of the two work buffers), out [nb][n / 2 + 1] bins, unnormalised
int fft_exec_r2c(const MstFftPlan *p, float *in, float2 *out, int nb, void *stream) {
    if (nb < 1) return MST_OK;
    if (nb > 65535) return fail(MST_ERR_UNSUPPORTED, "FFT: more than 65535 sequences per call");
    float2 *z;
    long sz;
    int rc = fft_passes(p, (float2 *)in, p->m, out, p->m + 1, nb, 0, stream, &z, &sz);
    if (rc) return rc;
    MST_LAUNCH(fft_r2c_post_kernel, dim3((unsigned)((p->m / 2 + 1 + 255) / 256), (unsigned)nb), dim3(256), stream, (const float2 *)z, out,
               (const float2 *)p->tw_n, p->m, sz, p->m + 1);
    MST_CHECK_LAUNCH("fft_r2c_post_kernel");
    return MST_OK;
}
// hipfftExecC2R's contract: in [nb][n / 2 + 1] bins (DESTROYED), out [nb][n] reals = n * irfft(in)
int fft_exec_c2r(const MstFftPlan *p, float2 *in, float *out, int nb, void *stream) {
    if (nb < 1) return MST_OK;
    if (nb > 65535) return fail(MST_ERR_UNSUPPORTED, "FFT: more than 65535 sequences per call");
    float2 *o = (float2 *)out;
    // the passes alternate between the two buffers and must end in `out`: an even number starts there, an odd number starts in `in`
    float2 *start = (p->passes % 2 == 0) ? o : in;
    const long sstart = (p->passes % 2 == 0) ? p->m : p->m + 1;
    MST_LAUNCH(fft_c2r_pre_kernel, dim3((unsigned)((p->m / 2 + 1 + 255) / 256), (unsigned)nb), dim3(256), stream, (const float2 *)in, start,
               (const float2 *)p->tw_n, p->m, p->m + 1, sstart);
    MST_CHECK_LAUNCH("fft_c2r_pre_kernel");
    float2 *other = (start == o) ? in : o;
    const long sother = (start == o) ? p->m + 1 : p->m;
    float2 *z;
    long sz;
    int rc = fft_passes(p, start, sstart, other, sother, nb, 1, stream, &z, &sz);
    if (rc) return rc;
    if (z != o) return fail(MST_ERR_STATE, "FFT: inverse passes ended in the wrong buffer");
    return MST_OK;
}
}  // namespace

// ---- FFT convolution (ConvolutionalReverb) ---------------------------------------------------------------------------
struct MstConvolver {
    long L = 0, Lh_max = 0, n_fft = 0;
    long step = 0, shift = 0;       // overlap-save: block b holds the samples b * step - shift + i; one block: step = n_fft, shift = 0
    int nb = 1;                     // blocks per (item, channel)
    int n_items = 0, C = 0;
    MstFftPlan *plan = nullptr;     // one plan serves the signal blocks, the impulse response and the inverse
};

extern "C" int mst_fx_convolver_create(long L, long Lh_max, int n_items, int C, MstConvolver **out) {
    if (!out || L < 1 || Lh_max < 1 || n_items < 1 || C < 1) return fail(MST_ERR_ARG, "mst_fx_convolver_create: bad argument");
    long n = 4;
    while (n < L + Lh_max - 1) n <<= 1;
    long step = n, shift = 0;
    int nb = 1;
    if (n > (1L << 18)) {
        // a long signal: overlap-save blocks of max(2^16, 4 x the response rounded up to a power of two) samples - one plan for every
        // signal length, float32 rounding of a 2^16..2^19-point transform
        long nh = 1;
        while (nh < Lh_max) nh <<= 1;
        long nblk = 4 * nh > (1L << 16) ? 4 * nh : (1L << 16);
        if (nblk > MST_FFT_MAX_N) nblk = 2 * nh;          // a response longer than 2^19 samples (11.9 s at 44.1 kHz): blocks of twice its length
        if (nblk < n) {
            n = nblk;
            shift = Lh_max - 1;
            step = n - shift;
            nb = (int)((L + Lh_max - 1 + step - 1) / step);
        }
    }
    if (n > MST_FFT_MAX_N)
        return fail(MST_ERR_UNSUPPORTED, "mst_fx_convolver_create: impulse responses longer than 2^20 samples (23.8 s at 44.1 kHz) need a transform beyond 2^21 points");
    if ((long)n_items * C * nb > 65535) return fail(MST_ERR_UNSUPPORTED, "mst_fx_convolver_create: more than 65535 transform blocks (split the batch)");
    auto *cv = new MstConvolver;
    cv->L = L; cv->Lh_max = Lh_max; cv->n_fft = n; cv->n_items = n_items; cv->C = C;
    cv->step = step; cv->shift = shift; cv->nb = nb;
    const int rc = fft_plan_create(&cv->plan, n);
    if (rc) {
        mst_fx_convolver_destroy(cv);
        return rc;
    }
    *out = cv;
    return MST_OK;
}

extern "C" void mst_fx_convolver_destroy(MstConvolver *cv) {
    if (!cv) return;
    fft_plan_destroy(cv->plan);
    delete cv;
}

extern "C" size_t mst_fx_convolver_workspace_bytes(const MstConvolver *cv) {
    if (!cv) return 0;
    const size_t nbin = (size_t)cv->n_fft / 2 + 1, seqs = (size_t)cv->n_items * cv->C * cv->nb + cv->C;
    return seqs * (size_t)cv->n_fft * sizeof(float) + seqs * nbin * sizeof(float2) + 256;
}

extern "C" int mst_fx_convolve(MstConvolver *cv, const float *x, const float *h, long Lh, float *y, long offset, double dry,
                               double wet, void *ws, size_t ws_bytes, void *stream) {
    if (!cv || !x || !h || !y || !ws) return fail(MST_ERR_ARG, "mst_fx_convolve: bad argument");
    if (Lh < 1 || Lh > cv->Lh_max) return fail(MST_ERR_ARG, "mst_fx_convolve: impulse response longer than the convolver was created for");
    if (offset < 0 || offset > Lh - 1) return fail(MST_ERR_ARG, "mst_fx_convolve: offset outside [0, Lh-1]");
    if (ws_bytes < mst_fx_convolver_workspace_bytes(cv)) return fail(MST_ERR_WORKSPACE, "mst_fx_convolve: workspace too small");
    const long n = cv->n_fft, nbin = n / 2 + 1;
    const int nseq = cv->n_items * cv->C, C = cv->C, nb = cv->nb, nblk = nseq * nb;
    int rc;
    float *rx = (float *)ws, *rh = rx + (size_t)nblk * n;
    float2 *cx = (float2 *)(((uintptr_t)(rh + (size_t)C * n) + 255) & ~(uintptr_t)255), *ch = cx + (size_t)nblk * nbin;
    const unsigned gb = (unsigned)((n + 255) / 256);
    MST_LAUNCH(fx_conv_pack_kernel, dim3(gb, nblk), dim3(256), stream, x, rx, cv->L, C, n, nb, cv->step, cv->shift);
    MST_CHECK_LAUNCH("fx_conv_pack_kernel");
    MST_LAUNCH(fx_conv_pack_kernel, dim3(gb, C), dim3(256), stream, h, rh, Lh, C, n, 1, n, 0L);       // the IR is one [Lh][C] "item"
    MST_CHECK_LAUNCH("fx_conv_pack_kernel");
    if ((rc = fft_exec_r2c(cv->plan, rx, cx, nblk, stream)) || (rc = fft_exec_r2c(cv->plan, rh, ch, C, stream))) return rc;
    MST_LAUNCH(fx_conv_mul_kernel, dim3((unsigned)((nbin + 255) / 256), nblk), dim3(256), stream, cx, (const float2 *)ch, nbin, C, nb,
               1.0f / (float)n);
    MST_CHECK_LAUNCH("fx_conv_mul_kernel");
    if ((rc = fft_exec_c2r(cv->plan, cx, rx, nblk, stream))) return rc;
    const long per = cv->L * C;
    MST_LAUNCH(fx_conv_mix_kernel, dim3((unsigned)((per + 255) / 256), cv->n_items), dim3(256), stream, x, (const float *)rx, y, cv->L,
               C, n, nb, cv->step, cv->shift, offset, (float)dry, (float)wet);
    MST_CHECK_LAUNCH("fx_conv_mix_kernel");
    return MST_OK;
}

extern "C" int mst_fx_rms_normalize(const float *x, float *y, int n_items, long per_x, long per_y, double *scratch, void *stream) {
    if (!x || !y || !scratch || n_items < 1 || per_x < 1 || per_y < 1) return fail(MST_ERR_ARG, "mst_fx_rms_normalize: bad argument");
    int rc;
    if ((rc = energy(x, scratch, n_items, per_x, 0, stream))) return rc;
    if ((rc = energy(y, scratch + 2 * n_items, n_items, per_y, 0, stream))) return rc;
    MST_LAUNCH(fx_scale_kernel, dim3((unsigned)((per_y + 255) / 256), n_items), dim3(256), stream, x, y, per_y, 1.0f,
               (const double *)scratch, (const double *)(scratch + 2 * n_items), 1, per_x, (const double *)nullptr);
    MST_CHECK_LAUNCH("fx_scale_kernel");
    return MST_OK;
}

// ---- chain fusion helpers -------------------------------------------------------------------------------------------------
extern "C" int mst_fx_sumsq(const float *x, int n_items, long per_item, double *out, void *stream) {
    if (!x || !out || n_items < 1 || per_item < 1) return fail(MST_ERR_ARG, "mst_fx_sumsq: bad argument");
    int chunks = (int)std::min<long>(MST_SUMSQ_SLOTS, (per_item + 8191) / 8192);
    if (chunks < 1) chunks = 1;
    MST_LAUNCH(fx_sumsq_kernel, dim3(n_items * chunks), dim3(256), stream, x, out, per_item, chunks);
    MST_CHECK_LAUNCH("fx_sumsq_kernel");
    return MST_OK;
}

extern "C" int mst_fx_rms_pending(const double *scale_x, const double *sumsq_x, long per_x, const double *sumsq_y, long per_y,
                                  double *scale_out, int n_items, void *stream) {
    if (!sumsq_x || !sumsq_y || !scale_out || n_items < 1 || per_x < 1 || per_y < 1) return fail(MST_ERR_ARG, "mst_fx_rms_pending: bad argument");
    MST_LAUNCH(fx_rms_pending_kernel, dim3(n_items), dim3(64), stream, scale_x, sumsq_x, per_x, sumsq_y, per_y, scale_out, n_items);
    MST_CHECK_LAUNCH("fx_rms_pending_kernel");
    return MST_OK;
}

extern "C" int mst_fx_scale_items(const float *x, float *y, int n_items, long per_item, const double *scale, void *stream) {
    if (!x || !y || !scale || n_items < 1 || per_item < 1) return fail(MST_ERR_ARG, "mst_fx_scale_items: bad argument");
    MST_LAUNCH(fx_scale_kernel, dim3((unsigned)((per_item + 255) / 256), n_items), dim3(256), stream, x, y, per_item, 1.0f,
               (const double *)nullptr, (const double *)nullptr, 0, per_item, scale);
    MST_CHECK_LAUNCH("fx_scale_kernel");
    return MST_OK;
}

// ---- STFT mean magnitude (EQ matching front end) ---------------------------------------------------------------------
struct MstStft {
    long n_fft = 0, hop = 0;
    int batch = 0;
    MstFftPlan *plan = nullptr;     // R2C of up to `batch` frames of n_fft
    float *win = nullptr;           // [n_fft] analysis window (device)
};

extern "C" int mst_fx_stft_create(long n_fft, long hop, const float *window_host, int max_batch, MstStft **out) {
    if (!out || !window_host || n_fft < 2 || hop < 1 || max_batch < 1) return fail(MST_ERR_ARG, "mst_fx_stft_create: bad argument");
    auto *st = new MstStft;
    st->n_fft = n_fft; st->hop = hop; st->batch = max_batch;
    const int rcp = fft_plan_create(&st->plan, n_fft);          // frame lengths are powers of two (the reference's FFT_SIZE is 65536)
    if (rcp) {
        delete st;
        return rcp;
    }
    if (hipMalloc((void **)&st->win, (size_t)n_fft * sizeof(float)) != hipSuccess ||
        hipMemcpy(st->win, window_host, (size_t)n_fft * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        fft_plan_destroy(st->plan);
        delete st;
        return fail(MST_ERR_HIP, "mst_fx_stft_create: hipMalloc failed");
    }
    *out = st;
    return MST_OK;
}

extern "C" void mst_fx_stft_destroy(MstStft *st) {
    if (!st) return;
    fft_plan_destroy(st->plan);
    (void)hipFree(st->win);
    delete st;
}

extern "C" size_t mst_fx_stft_workspace_bytes(const MstStft *st) {
    if (!st) return 0;
    return (size_t)st->batch * st->n_fft * sizeof(float) + (size_t)st->batch * (st->n_fft / 2 + 1) * sizeof(float2) + 512;
}

extern "C" int mst_fx_stft_mean_magnitude(MstStft *st, const float *x, long L, int C, int channel, float *mean_dev, void *ws,
                                          size_t ws_bytes, void *stream) {
    if (!st || !x || !mean_dev || !ws || C < 1 || channel < 0 || channel >= C) return fail(MST_ERR_ARG, "mst_fx_stft_mean_magnitude: bad argument");
    if (L < st->n_fft) return fail(MST_ERR_ARG, "mst_fx_stft_mean_magnitude: signal shorter than one frame");
    if (ws_bytes < mst_fx_stft_workspace_bytes(st)) return fail(MST_ERR_WORKSPACE, "mst_fx_stft_mean_magnitude: workspace too small");
    const long n = st->n_fft, nbin = n / 2 + 1;
    const long n_frames = 1 + (L - n) / st->hop;          // common_miscellaneous.py:64
    float *frames = (float *)ws;
    float2 *spec = (float2 *)(((uintptr_t)(frames + (size_t)st->batch * n) + 255) & ~(uintptr_t)255);
    MST_HIP_TRY(hipMemsetAsync(mean_dev, 0, (size_t)nbin * sizeof(float), (hipStream_t)stream));
    for (long f0 = 0; f0 < n_frames; f0 += st->batch) {
        const int nb = (int)std::min<long>(st->batch, n_frames - f0);
        MST_LAUNCH(fx_stft_frame_kernel, dim3((unsigned)((n + 255) / 256), nb), dim3(256), stream, x, frames, (const float *)st->win, L,
                   C, channel, n, st->hop, f0, n_frames);
        MST_CHECK_LAUNCH("fx_stft_frame_kernel");
        const int rcf = fft_exec_r2c(st->plan, frames, spec, nb, stream);          // only the frames that exist
        if (rcf) return rcf;
        MST_LAUNCH(fx_stft_mag_accum_kernel, dim3((unsigned)((nbin + 255) / 256)), dim3(256), stream, (const float2 *)spec, mean_dev, nbin, nb);
        MST_CHECK_LAUNCH("fx_stft_mag_accum_kernel");
    }
    MST_LAUNCH(fx_scale_inplace_kernel, dim3((unsigned)((nbin + 255) / 256)), dim3(256), stream, mean_dev, nbin, 1.0f / (float)n_frames);
    MST_CHECK_LAUNCH("fx_scale_inplace_kernel");
    return MST_OK;
}


extern "C" int mst_fx_stereo_moments(const float *x, int n_items, long L, double *out, void *stream) {
    if (!x || !out || n_items < 1 || L < 1) return fail(MST_ERR_ARG, "mst_fx_stereo_moments: bad argument");
    MST_HIP_TRY(hipMemsetAsync(out, 0, (size_t)n_items * 3 * sizeof(double), (hipStream_t)stream));
    const unsigned chunks = (unsigned)std::min<long>(256, (L + 4095) / 4096);
    MST_LAUNCH(fx_stereo_moments_kernel, dim3(chunks, n_items), dim3(256), stream, x, L, out);
    MST_CHECK_LAUNCH("fx_stereo_moments_kernel");
    return MST_OK;
}

extern "C" int mst_fx_stereo_mix(const float *x, float *y, int n_items, long L, float m00, float m01, float m10, float m11, void *stream) {
    if (!x || !y || n_items < 1 || L < 1) return fail(MST_ERR_ARG, "mst_fx_stereo_mix: bad argument");
    const long n = (long)n_items * L;
    MST_LAUNCH(fx_stereo_mix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), stream, x, y, n, m00, m01, m10, m11);
    MST_CHECK_LAUNCH("fx_stereo_mix_kernel");
    return MST_OK;
}

// ---- AlgorithmicReverb -------------------------------------------------------------------------------------------------
extern "C" size_t mst_fx_algorithmic_reverb_scratch_bytes(int n_items, long L, int n_combs) {
    if (n_items < 1 || L < 1 || n_combs < 1) return 0;
    return ((size_t)n_combs + 1) * n_items * 2 * (size_t)L * sizeof(double);
}

extern "C" int mst_fx_algorithmic_reverb(const float *x, float *y, int n_items, long L, int C, const int *comb_delays, int n_combs,
                                         const int *allpass_delays, int n_allpass, int stereo_spread, double damping, double room_size,
                                         double in_gain, double wet1, double wet2, double dry, double *scratch, size_t scratch_bytes,
                                         void *stream) {
    if (!x || !y || !comb_delays || !allpass_delays || !scratch || n_items < 1 || L < 1 || (C != 1 && C != 2) || n_combs < 1 ||
        n_combs > 8 || n_allpass < 1 || stereo_spread < 0)
        return fail(MST_ERR_ARG, "mst_fx_algorithmic_reverb: bad argument");
    if (scratch_bytes < mst_fx_algorithmic_reverb_scratch_bytes(n_items, L, n_combs))
        return fail(MST_ERR_WORKSPACE, "mst_fx_algorithmic_reverb: scratch too small");
    CombArgs a;
    a.x = x;
    a.y = scratch + (size_t)n_items * 2 * L;            // [n_combs][n_items * 2][L] behind the wet buffer
    a.L = L;
    a.C = C;
    a.n_items = n_items;
    a.n_combs = n_combs;
    for (int k = 0; k < n_combs; ++k) {
        if (comb_delays[k] < 1 || comb_delays[k] + stereo_spread > 2048)
            return fail(MST_ERR_UNSUPPORTED, "mst_fx_algorithmic_reverb: comb delays up to 2048 samples");
        a.delay[k][0] = comb_delays[k];
        a.delay[k][1] = comb_delays[k] + stereo_spread;
    }
    a.damp = damping;
    a.feedback = room_size;
    a.in_gain = in_gain;
    MST_LAUNCH(fx_comb_kernel, dim3(n_combs, n_items * 2), dim3(64), stream, a);
    MST_CHECK_LAUNCH("fx_comb_kernel");
    double *wet = scratch;                                // [n_items * 2][L]
    for (int k = 0; k < n_allpass; ++k) {
        const int dl = allpass_delays[2 * k], dr = allpass_delays[2 * k + 1];
        if (dl < 1 || dr < 1) return fail(MST_ERR_ARG, "mst_fx_algorithmic_reverb: all-pass delay < 1");
        const int threads = std::min(1024, std::max(64, ((std::max(dl, dr) + 63) / 64) * 64));
        MST_LAUNCH(fx_allpass_kernel, dim3(n_items * 2), dim3(threads), stream, wet, (const double *)a.y, k == 0 ? n_combs : 0, 0, L,
                   n_items * 2, dl, dr, room_size);
        MST_CHECK_LAUNCH("fx_allpass_kernel");
    }
    MST_LAUNCH(fx_reverb_mix_kernel, dim3((unsigned)((L + 255) / 256), n_items), dim3(256), stream, x, (const double *)wet, y, L, C, wet1,
               wet2, dry);
    MST_CHECK_LAUNCH("fx_reverb_mix_kernel");
    return MST_OK;
}
