// libmst_hip.so, MixFXcloner part of the C ABI (mst_tcn_*, mst_calib_mainloop, mst_film_forward): weight packing into MFMA fragment order,
// tile geometry and launches of csrc/tcn_kernels.h.  See include/mst_hip.h for the contract.
#include "mst_host.h"
#include "tcn_kernels.h"

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

constexpr int TCN_FILM_ROWS0 = 64;      // FiLM rows reserved at create time (14 blocks x 64 rows x 256 floats = 0.9 MB)
struct MstTcn {
    MstTcnDesc d;
    bool generic = false;              // configuration outside the specialised 128-channel / k=15 kernels
    std::vector<MstEncConv> gconv;     // generic path: one packed conv per block + the output head (fp32 implicit GEMM)
    std::vector<MstTcnBlock> blk;
    float *film_w = nullptr;  // [nblocks][2C][D]
    float *film_b = nullptr;  // [nblocks][2C]
    float *film = nullptr;    // [nblocks][rows][2C]
    int film_rows = 0, film_cap = 0;
    std::vector<float *> film_retired;   // FiLM tables outgrown by a larger set_cond: kept until destroy (a forward still in flight may read them; no hipFree - a device-wide wait - on the data path)
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
    int bf16_onetile = 1;         // bf16 mode: the two- / four-phase class-major blocks on the ONE-TILE kernel's 256-time tiles, two workgroups per CU, instead of the duo
                                  // kernel (mst_tcn_set_tuning bit 7, round 6: 1.31 against 1.40 ms per launch, the d = 2 block with block 0 inside 1.47 against 1.56; bit-identical)
    std::vector<hipEvent_t> ev;   // timing hook: (nblocks + 2) events per recorded forward
    int ev_max = 0, ev_used = 0;
};


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
        hipMalloc(&t->zero_row, 1024) != hipSuccess || hipMemset(t->zero_row, 0, 1024) != hipSuccess ||
        hipMalloc((void **)&t->film, (size_t)d.nblocks * TCN_FILM_ROWS0 * 2 * d.channels * sizeof(float)) != hipSuccess) {
        (void)hipFree(t->film_w);
        (void)hipFree(t->film_b);
        (void)hipFree(t->zero_row);
        (void)hipFree(t->film);
        delete t;
        return fail(MST_ERR_HIP, "mst_tcn_create: hipMalloc failed");
    }
    t->film_cap = TCN_FILM_ROWS0;
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
    for (float *f : t->film_retired) (void)hipFree(f);
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
        // the table is a high-water buffer: mst_tcn_create reserves TCN_FILM_ROWS0 rows (one pass of the whole-stem engine at 131072-sample
        // segments), so the per-pass set_cond of the interpolation loop never allocates.  More rows than ever before: a NEW table of at least
        // twice the size; the old one is retired, not freed (hipFree waits for the whole device and a forward in flight may still read it)
        const int cap = std::max(n_rows, 2 * t->film_cap);
        float *grown = nullptr;
        MST_HIP_TRY(hipMalloc((void **)&grown, (size_t)t->d.nblocks * cap * 2 * t->d.channels * sizeof(float)));
        t->film_retired.push_back(t->film);
        t->film = grown;
        t->film_cap = cap;
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

// bf16 mode, tile forms that tcn_run picks beside the phase count (launch_block's bf16_tile)
enum { TILE_DEFAULT = 0,             // 256-time tiles of P phases (128-time at P = 8)
       TILE_128_FOUR_PHASES = 1,     // 17 ... 32 steps per phase: 128-time tiles of four phases instead of eight, three workgroups per CU
       TILE_WHOLE_256 = 3 };         // a phase sequence is exactly one 256-time tile (64 / 32 / 16 steps at 4 / 8 / 16 phases): unrolled loop, trimmed halo
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
                                  int bf16_tile = TILE_DEFAULT, int bf16_reuse = 0, int x3_half_cm = 0, int bf16_onetile = 0) {
    TcnBlockArgs a = a0;
    if constexpr (P == 4) {
        // (the same 128-time form for EVERY block - three workgroups per CU instead of the duo kernel - measured 1.53-1.58 ms per launch
        //  against 1.48-1.53: it only wins where the eight-phase tiles' halo is the alternative)
        if (precision == MST_PREC_BF16 && bf16_tile == TILE_128_FOUR_PHASES) {          // 128-time tiles of 4 phases (one-tile kernel, three workgroups per CU)
            const long nsteps = ((long)a.L + a.d - 1) / a.d;
            a.tiles_step = (int)((nsteps + 128 / P - 1) / (128 / P));
            const long g2 = (long)a.B * a.tiles_phase * a.tiles_step;
            if (g2 % 8 == 0) a.xcd_tiles = (int)(g2 / 8);
            const bool whole = a.tiles_step == 1 && nsteps == 128 / P;          // every tile spans its whole phase sequence: unrolled class-major loop
            if (a.y_out && whole)
                MST_LAUNCH((tcn_block_bf16_kernel<P, true, 4, 1>), dim3((unsigned)g2), dim3(256), stream, a);
            else if (a.y_out)
                MST_LAUNCH((tcn_block_bf16_kernel<P, true, 4>), dim3((unsigned)g2), dim3(256), stream, a);
            else if (whole)
                MST_LAUNCH((tcn_block_bf16_kernel<P, false, 4, 1>), dim3((unsigned)g2), dim3(256), stream, a);
            else
                MST_LAUNCH((tcn_block_bf16_kernel<P, false, 4>), dim3((unsigned)g2), dim3(256), stream, a);
            MST_CHECK_LAUNCH("tcn_block_bf16_kernel");
            return MST_OK;
        }
    }
    if constexpr (P == 16 || P == 8 || P == 4) {
        if (precision == MST_PREC_BF16 && bf16_tile == TILE_WHOLE_256) {          // one 256-time tile = the whole phase sequence (tcn_run checked the shape)
            a.tiles_step = 1;
            grid = (int)((long)a.B * a.tiles_phase);
            if (grid % 8 == 0) a.xcd_tiles = grid / 8;
            if constexpr (P == 16) {
                if (a.y_out) MST_LAUNCH((tcn_block_bf16_kernel<P, true, 8, 1>), dim3(grid), dim3(256), stream, a);
                else MST_LAUNCH((tcn_block_bf16_kernel<P, false, 8, 1>), dim3(grid), dim3(256), stream, a);
            } else {
                if (a.y_out) return fail(MST_ERR_STATE, "tcn_block_bf16_kernel: the fused head exists for the sixteen-phase whole-sequence tile only");
                MST_LAUNCH((tcn_block_bf16_kernel<P, false, 8, 1>), dim3(grid), dim3(256), stream, a);
            }
            MST_CHECK_LAUNCH("tcn_block_bf16_kernel");
            return MST_OK;
        }
    }
    if (precision == MST_PREC_BF16 && bf16_form == 2) {
        // 256-time tiles only: at P = 8 (128-time tiles: half the work per tile for the same two barriers) the duo form measured
        // 1.62-1.82 ms against 1.50 ms, those blocks run the one-tile-per-workgroup kernel
        // (the last block - fused output head, 32 more live registers - spills in the duo form and runs the one-tile kernel too)
        if constexpr (P == 4) {
            // bit 7 (round 6): the class-major four-phase blocks on the one-tile kernel, 256-time tiles, two workgroups per CU
            if (bf16_onetile && bf16_reuse) {
                if (grid % 8 == 0) a.xcd_tiles = grid / 8;
                if (a.y_out) MST_LAUNCH((tcn_block_bf16_kernel<P, true, 8, 2>), dim3(grid), dim3(256), stream, a);          // the last block of a long segment: fused head
                else MST_LAUNCH((tcn_block_bf16_kernel<P, false, 8, 2>), dim3(grid), dim3(256), stream, a);
                MST_CHECK_LAUNCH("tcn_block_bf16_kernel");
                return MST_OK;
            }
        }
        if constexpr (P == 2) {
            // ... and the two-phase blocks (the d = 2 block; with bit 5 block 0 is computed in its staging)
            if (!a.y_out && bf16_onetile && bf16_reuse) {
                if (grid % 8 == 0) a.xcd_tiles = grid / 8;
                if (a.x0) MST_LAUNCH((tcn_block_bf16_kernel<P, false, 8, 2, true>), dim3(grid), dim3(256), stream, a);
                else MST_LAUNCH((tcn_block_bf16_kernel<P, false, 8, 2>), dim3(grid), dim3(256), stream, a);
                MST_CHECK_LAUNCH("tcn_block_bf16_kernel");
                return MST_OK;
            }
        }
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
            const bool whole = a.tiles_step == 1 && nsteps == 128 / P;
            if (a.y_out && whole)
                MST_LAUNCH((tcn_block_bf16_kernel<P, true, 4, 1>), dim3((unsigned)g2), dim3(256), stream, a);
            else if (a.y_out)
                MST_LAUNCH((tcn_block_bf16_kernel<P, true, 4>), dim3((unsigned)g2), dim3(256), stream, a);
            else if (whole)
                MST_LAUNCH((tcn_block_bf16_kernel<P, false, 4, 1>), dim3((unsigned)g2), dim3(256), stream, a);
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

    // block 0 inside block 1's launch (bf16, tuning bit 5): block 1 must be the d = 2 block on two-phase class-major tiles (the one-tile kernel
    // with bit 7, else the duo kernel) and not the last block; the probes of block 0 itself (n_run == 1) always run the separate kernel
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
        int bf16_tile = TILE_DEFAULT;
        if (precision == MST_PREC_BF16 && P == 8) {
            const long ns = ((long)L + d - 1) / d;
            if (ns > 16 && ns <= 32) {
                P = 4;
                bf16_tile = TILE_128_FOUR_PHASES;
            }
        }
        // bf16 (tuning bit 7), a block whose phase sequences are EXACTLY one 256-time tile - sixteen phases x 16 steps (d = 8192 at L = 131072, the last
        // block), eight x 32 (d = 4096), four x 64 (d = 2048): the unrolled class-major loop without the all-padding (column tile, tap) pairs, an LDS image
        // without the halo steps no live row window reaches (256 / 272 / 280 rows), two workgroups per CU
        if (precision == MST_PREC_BF16 && t->bf16_onetile && t->bf16_reuse && t->bf16_form == 2) {
            const long ns = ((long)L + d - 1) / d;
            const int Pw = ns == 16 ? 16 : (ns == 32 ? 8 : (ns == 64 ? 4 : 0));
            // (with the fused output head - the last block - only the sixteen-phase form fits 256 registers: 248; the other two would spill)
            const bool head = !act_out && n == t->d.nblocks - 1;
            if (Pw && d % Pw == 0 && (long)L == ns * d && (!head || Pw == 16)) {
                P = Pw;
                bf16_tile = TILE_WHOLE_256;
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
            case 1: rc = launch_block<1>(precision, a, (int)grid, stream, x3_small, t->bf16_form, bf16_tile, t->bf16_reuse, t->x3_half_cm, t->bf16_onetile); break;
            case 2: rc = launch_block<2>(precision, a, (int)grid, stream, x3_small, t->bf16_form, bf16_tile, t->bf16_reuse, t->x3_half_cm, t->bf16_onetile); break;
            case 4: rc = launch_block<4>(precision, a, (int)grid, stream, x3_small, t->bf16_form, bf16_tile, t->bf16_reuse, t->x3_half_cm, t->bf16_onetile); break;
            case 8: rc = launch_block<8>(precision, a, (int)grid, stream, x3_small, t->bf16_form, bf16_tile, t->bf16_reuse, t->x3_half_cm, t->bf16_onetile); break;
            default: rc = launch_block<16>(precision, a, (int)grid, stream, 0, t->bf16_form, bf16_tile); break;
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
    if (flags < 0 || flags > 255 || (((flags >> 1) & 3) != 0 && ((flags >> 1) & 3) != 2) || ((flags >> 3) & 1))
        return fail(MST_ERR_ARG, "mst_tcn_set_tuning: unknown flag bits (form 1 - the stream kernel - and bit 3 - the split-bf16 duo kernel - left the library in round 5)");
    t->x3_small_tiles = flags & 1;
    t->bf16_form = (flags >> 1) & 3;
    t->bf16_reuse = (flags >> 4) & 1;
    t->bf16_fuse0 = (flags >> 5) & 1;
    t->x3_half_cm = (flags >> 6) & 1;
    t->bf16_onetile = (flags >> 7) & 1;
    return MST_OK;
}

extern "C" int mst_tcn_get_tuning(const MstTcn *t, int *flags, int *last_forward_fused_block0) {
    if (!t) return fail(MST_ERR_ARG, "mst_tcn_get_tuning: null handle");
    if (flags)
        *flags = t->x3_small_tiles | t->bf16_form << 1 | t->bf16_reuse << 4 | t->bf16_fuse0 << 5 | t->x3_half_cm << 6 | t->bf16_onetile << 7;
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

