// libmst_hip.so, FX-processor part of the C ABI (mst_fx_*): equaliser, compressor, imager, gain, Haas / panner, FFT convolution, STFT,
// algorithmic reverb - launches of csrc/fx_kernels.h and csrc/fft_kernels.h.  See include/mst_hip.h for the contract.
#include "mst_host.h"
#include "fft_kernels.h"
#include "fx_kernels.h"

// =================================================================================================
// FX processors
// =================================================================================================
namespace {
// an MstFxFuse built against another layout of the struct would be read as garbage: its first member is its own size
int fuse_check(const MstFxFuse *fuse, const char *who) {
    if (fuse && fuse->struct_size != sizeof(MstFxFuse)) return fail(MST_ERR_ARG, std::string(who) + ": MstFxFuse.struct_size does not match this library's layout");
    if (fuse && (fuse->forms & ~(MST_FX_FORM_EQ_LANE_APPLY | MST_FX_FORM_EQ_VALU_ENDS | MST_FX_FORM_COMP_SLICE_SMALL))) return fail(MST_ERR_ARG, std::string(who) + ": unknown MstFxFuse.forms bits");
    return MST_OK;
}
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
// buffer; the 16 most recent entries per device are kept.  A miss (rare: a new equaliser setting) uploads and WAITS for the upload; evicting
// an entry also waits for the device (randomised parameter sweeps) - neither can happen inside a stream capture.
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
    // a cache HIT hands devp to whatever stream (or thread) asks next, with no ordering against this upload: the table is complete before
    // this function - and with it the mutex - lets anybody see the entry (a miss is rare: once per equaliser setting)
    if (hipMemcpyAsync(t->devp, t->host.data(), n_all * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
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
    if (int rc = fuse_check(fuse, "mst_fx_biquad_cascade")) return rc;
    // kernel forms of THIS call (MstFxFuse.forms; results do not depend on them): the reference forms of the stereo passes
    const bool eq_lane_apply = fuse && (fuse->forms & MST_FX_FORM_EQ_LANE_APPLY), eq_valu_ends = fuse && (fuse->forms & MST_FX_FORM_EQ_VALU_ENDS);
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
        if (C == 2 && !eq_valu_ends && M % 16 == 0) {          // stereo: the end states as a matrix product on the float64 matrix cores
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
        if (C == 2 && !eq_lane_apply) {          // stereo: the chunks travel in 16-frame slabs through LDS, in and out
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
// [nchunks][n_seq] | log10 table [256] | carry [n_seq] | energy partials of the apply pass [ceil(L / 64)][max(n_seq, 3 n_items)]
struct CompScratch { size_t xl, maps, ystart, tab, carry, tsums, total; long nchunks, ntiles; };
CompScratch comp_scratch(int n_items, long L, int C) {
    CompScratch c;
    const size_t n_seq = (size_t)n_items * C;
    c.nchunks = (L + MST_COMP_T - 1) / MST_COMP_T;
    c.xl = c.nchunks < 4 ? n_seq * (size_t)L * sizeof(double) : 0;      // only the serial form of very short signals stores them
    c.maps = n_seq * (size_t)c.nchunks * MST_COMP_REC * sizeof(double);
    c.ystart = n_seq * (size_t)c.nchunks * sizeof(double);
    c.tab = 256 * sizeof(double);
    c.carry = n_seq * sizeof(double);                                      // the smoother's value between two time slices of the chain
    c.ntiles = (L + 63) / 64;
    c.tsums = (size_t)c.ntiles * std::max(n_seq, (size_t)3 * n_items) * sizeof(double);      // the apply pass's energy partials per 64-sample tile
    c.total = c.xl + c.maps + c.ystart + c.tab + c.carry + c.tsums;
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
constexpr int FX_SLICES = 3;    // time slices of a large compressor call (measured: 0.523-0.539 ms per chain with 3, 0.540-0.550 with 4, 0.556-0.580 with 2, 0.67 with 8; profiles/r05_fx_slices_ab.txt)
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

int compressor_run(CompArgs a, int n_items, long L, int C, double *scratch, size_t scratch_bytes, void *stream, bool slice_small = false) {
    if (!scratch) {
        MST_LAUNCH(fx_compressor_kernel, dim3((a.n_seq + 3) / 4), dim3(256), stream, a);
        MST_CHECK_LAUNCH("fx_compressor_kernel");
        return MST_OK;
    }
    if (scratch_bytes < mst_fx_compressor_scratch_bytes(n_items, L, C))
        return fail(MST_ERR_WORKSPACE, "mst_fx_compressor: scratch too small");
    if (a.out_ms && C != 2) return fail(MST_ERR_ARG, "mst_fx_compressor: mid / side energies need stereo items");
    const CompScratch cs = comp_scratch(n_items, L, C);
    const dim3 tiles((unsigned)cs.ntiles, (unsigned)((a.n_seq + 63) / 64));      // 64 x 64 (time x sequence) tiles
    double *tsums = (double *)((unsigned char *)scratch + cs.xl + cs.maps + cs.ystart + cs.tab + cs.carry);
    // the energy sums of the output (chain fusion): per-tile partials from the apply pass, reduced in a fixed order
    auto finish = [&]() -> int {
        if (!a.out_sumsq) return MST_OK;
        MST_LAUNCH(fx_tile_sums_kernel, dim3((unsigned)n_items), dim3(MST_TILE_SUBS * MST_SUMSQ_SLOTS), stream, (const double *)tsums, cs.ntiles, n_items, C,
                   a.out_ms ? 1 : 0, a.out_sumsq, a.out_ms);
        MST_CHECK_LAUNCH("fx_tile_sums_kernel");
        return MST_OK;
    };
    if (cs.nchunks < 4) {                               // very short signals: level differences, serial smoother, gain application
        MST_LAUNCH(fx_comp_gain_kernel, tiles, dim3(256), stream, a, scratch);
        MST_CHECK_LAUNCH("fx_comp_gain_kernel");
        MST_LAUNCH(fx_comp_smooth_kernel, dim3((a.n_seq + 63) / 64), dim3(64), stream, a, scratch);
        MST_CHECK_LAUNCH("fx_comp_smooth_kernel");
        MST_LAUNCH((fx_comp_apply_kernel<false>), tiles, dim3(256), stream, a, (const double *)scratch, (const double *)nullptr, 0, 0, tsums);
        MST_CHECK_LAUNCH("fx_comp_apply_kernel");
        return finish();
    }
    // the smoother parallel in time: chunk maps (convex piecewise-linear), a chain over chunks, the rest in one pass
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
    // the map and apply kernels are throughput work.  The signal is cut into FX_SLICES (three) slices of whole chain batches; the caller's
    // stream runs the chain of slice 0, 1, ... back to back, a side stream (lower priority) the maps of slice 1, 2, ... and the applies of slice
    // 0 .. NS - 2 beside it (events order map_i -> chain_i -> apply_i); the last apply follows the last chain on the caller's stream, which
    // then waits for the side stream.  Same arithmetic, same results (the smoother's value crosses a slice boundary as a float64 in
    // ycarry); without concurrency (a profiler serialising the queues) the launches simply run one after the other.
    // (Round 6 measured two other ways of overlapping the three kernels - the tail inside the chain kernel, and ONE chain launch that polls
    //  the map kernel's progress counters with the apply kernel polling the chain's: 0.74 and 0.51 ms per chain against 0.49 with the slices;
    //  EXPERIMENTS.md section E.3, tools/proto/r06_fx_*.)
    const int nbatch = (int)((cs.nchunks + MST_CHAIN_CB - 1) / MST_CHAIN_CB);
    const int gy = (a.n_seq + 63) / 64;
    int ns = ((nbatch >= 32 && (double)a.n_seq * (double)L >= 4.0e6) || (slice_small && nbatch >= 8)) ? FX_SLICES : 1;
    FxSide *side = ns > 1 ? fx_side() : nullptr;
    if (!side) ns = 1;
    auto launch_map = [&](int b0, int b1, void *st) -> int {
        CompMapArgs mm = m;
        mm.chunk0 = b0 * MST_CHAIN_CB;
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
        MST_LAUNCH((fx_comp_apply_kernel<true>), dim3((unsigned)(t1 - t0), tiles.y), dim3(256), st, a, tab, (const double *)m.ystart, m.nchunks, (int)t0, tsums);
        MST_CHECK_LAUNCH("fx_comp_apply_kernel");
        return MST_OK;
    };
    static_assert(MST_COMP_T == 32 && MST_CHAIN_CB % 2 == 0, "two chunks per 64-sample apply tile");
    int rc;
    if (ns == 1) {
        if ((rc = launch_map(0, nbatch, stream)) || (rc = launch_chain(0, nbatch, stream)) || (rc = launch_apply(0, nbatch, stream))) return rc;
        return finish();
    }
    std::lock_guard<std::mutex> lock(side->mu);          // one fork / join at a time per device: the events are reused
    hipStream_t main_s = (hipStream_t)stream, side_s = side->stream;
    auto bound = [&](int i) { return (int)((long)nbatch * i / ns); };
    if ((rc = launch_map(0, bound(1), stream))) return rc;          // slice 0's map: nothing to overlap it with
    MST_HIP_TRY(hipEventRecord(side->fork, main_s));               // the side stream sees the input
    MST_HIP_TRY(hipStreamWaitEvent(side_s, side->fork, 0));
    // From here on the side stream may hold work on the caller's buffers.  Whatever goes wrong below, the caller's stream is made to wait
    // for it before this call returns (torch's allocator hands the buffers to the next user of the CALLER's stream).
    auto forked = [&]() -> int {
        int r;
        hipError_t e;
        for (int i = 1; i < ns; ++i) {
            if ((r = launch_map(bound(i), bound(i + 1), side_s))) return r;
            if ((e = hipEventRecord(side->map_done[i], side_s)) != hipSuccess) return fail(MST_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e));
        }
        for (int i = 0; i < ns; ++i) {
            if (i > 0 && (e = hipStreamWaitEvent(main_s, side->map_done[i], 0)) != hipSuccess) return fail(MST_ERR_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e));
            if ((r = launch_chain(bound(i), bound(i + 1), stream))) return r;
            if (i + 1 < ns) {
                if ((e = hipEventRecord(side->chain_done[i], main_s)) != hipSuccess || (e = hipStreamWaitEvent(side_s, side->chain_done[i], 0)) != hipSuccess)
                    return fail(MST_ERR_HIP, std::string("chain -> apply event: ") + hipGetErrorString(e));
                if ((r = launch_apply(bound(i), bound(i + 1), side_s))) return r;
            }
        }
        return launch_apply(bound(ns - 1), nbatch, stream);
    };
    rc = forked();
    hipError_t e = hipEventRecord(side->join, side_s);
    if (e == hipSuccess) e = hipStreamWaitEvent(main_s, side->join, 0);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(side_s);            // the join could not be expressed as an event: wait for the side stream here
        if (rc == MST_OK) rc = fail(MST_ERR_HIP, std::string("mst_fx_compressor: side stream join: ") + hipGetErrorString(e));
    }
    return rc ? rc : finish();
}
}  // namespace

extern "C" int mst_fx_compressor(const float *x, float *y, int n_items, long L, int C, double threshold_db,
                                 double attack_ms, double release_ms, double ratio, double sample_rate, double *scratch,
                                 size_t scratch_bytes, const MstFxFuse *fuse, void *stream) {
    if (!x || !y || n_items < 1 || L < 1 || C < 1 || attack_ms <= 0 || release_ms <= 0 || ratio <= 0 || sample_rate <= 0)
        return fail(MST_ERR_ARG, "mst_fx_compressor: bad argument");
    if (int rc = fuse_check(fuse, "mst_fx_compressor")) return rc;
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
    return compressor_run(a, n_items, L, C, scratch, scratch_bytes, stream, fuse && (fuse->forms & MST_FX_FORM_COMP_SLICE_SMALL));
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
    if (int rc = fuse_check(fuse, "mst_fx_midside_imager")) return rc;
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
    if (int rc = fuse_check(fuse, "mst_fx_gain")) return rc;
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
