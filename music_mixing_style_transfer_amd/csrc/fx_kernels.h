// gfx950 kernels of the FX-manipulator processors (reference mixing_manipulator/common_audioeffects.py).
// Audio layout as the reference's processors: x[item][n][c], time-major with interleaved channels, fp32;
// float64 internal arithmetic for the recursions like the reference.  These are latency / HBM bound:
// no matrix work here.
#pragma once
#include "mst_dev.h"

#define MST_MAX_BANDS 8

// ------------------------------------------------------------------------------------------------
// Equaliser.process (:500-525): cascade of biquads, each over the whole signal with zero initial state
// (transposed direct form II, the scipy.signal.lfilter recursion).  Running the bands sample-by-sample in
// one pass is the same arithmetic as the reference's band-by-band passes (band k's output sequence only
// depends on band k-1's output sequence).  One lane per (item, channel) sequence; samples are fetched 16 at
// a time so the global loads are off the recursion's dependency chain.
// ------------------------------------------------------------------------------------------------
// One sample through one band (transposed direct form II, scipy.signal.lfilter's association: y = b0 x + z1; z1' = (z2 + b1 x) - a1 y;
// z2' = b2 x - a2 y), as explicit fused multiply-adds: the two state updates hang on y by ONE operation each (the products with x do not
// wait for y) - the recursion's dependent chain per band is fma -> fma instead of fma -> mul -> fma -> add, and 5 instructions instead of 6.
// Every kernel that runs the recursion calls this: they produce the same bits.
__host__ __device__ __forceinline__ double fx_biquad_band(double v, double &z1, double &z2, const double (&cf)[5]) {
    const double p1 = fma(cf[1], v, z2), p2 = cf[2] * v;
    const double yn = fma(cf[0], v, z1);
    z1 = fma(-cf[3], yn, p1);
    z2 = fma(-cf[4], yn, p2);
    return yn;
}

struct BiquadArgs {
    const float *x;
    float *y;
    int n_seq;      // n_items * C
    int C;
    long L;
    int n_bands;
    double coef[MST_MAX_BANDS][5];   // b0 b1 b2 a1 a2
};

__global__ __launch_bounds__(64) void fx_biquad_kernel(BiquadArgs a) {
    const int seq = blockIdx.x * 64 + threadIdx.x;
    if (seq >= a.n_seq) return;
    const int item = seq / a.C, c = seq % a.C;
    const float *xp = a.x + (size_t)item * a.L * a.C + c;
    float *yp = a.y + (size_t)item * a.L * a.C + c;
    double z1[MST_MAX_BANDS], z2[MST_MAX_BANDS];
#pragma unroll
    for (int k = 0; k < MST_MAX_BANDS; ++k) z1[k] = z2[k] = 0.0;
    for (long n0 = 0; n0 < a.L; n0 += 16) {
        float xin[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) xin[i] = (n0 + i < a.L) ? xp[(n0 + i) * a.C] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double v = (double)xin[i];
#pragma unroll
            for (int k = 0; k < MST_MAX_BANDS; ++k) {
                if (k < a.n_bands) v = fx_biquad_band(v, z1[k], z2[k], a.coef[k]);
            }
            if (n0 + i < a.L) yp[(n0 + i) * a.C] = (float)v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same cascade, parallel in time.  A biquad cascade is a linear time-invariant system with state s (2 per
// band): s[n+1] = A s[n] + B x[n].  Cutting every sequence into chunks of M samples,
//   pass 1 (fx_biquad_chunk_state_kernel): each chunk runs the cascade from ZERO state -> its end state e_k;
//   scan   (fx_biquad_scan_kernel): true chunk-start states  s_k = A^M s_{k-1} + e_{k-1}  (A^M from the host);
//   pass 2 (fx_biquad_chunk_apply_kernel): each chunk re-runs the cascade from s_k and writes the output.
// Same float64 recursion per sample as the serial kernel; only the chunk-start states come through the scan
// (float64, rounding-level differences).  One lane per (sequence, chunk): 16 384 lanes instead of 128.
// ------------------------------------------------------------------------------------------------
// energy sums of the chain fusion are kept as MST_SUMSQ_SLOTS partial sums per item (producers spread their atomics over the slots:
// a 64-segment batch is 4096 tiles per kernel onto 64 items - one address per item serialises ~2000 atomics, measured +0.4 ms)
#define MST_SUMSQ_SLOTS 64
#define MST_IMAGER_FRAMES 2048    // frames per workgroup of fx_imager_apply_kernel

struct BiquadChunkArgs {
    const float *x;
    float *y;             // pass 2 only
    double *ends;         // [n_seq][nchunks][2*MST_MAX_BANDS]  zero-state end states (pass 1 out): one 128-byte record per chunk
    const double *starts; // [n_seq][nchunks][2*MST_MAX_BANDS]  true start states (pass 2 in)
    int n_seq, C, nchunks, M;
    long L;
    int n_bands;
    double coef[MST_MAX_BANDS][5];
    // chain fusion (AugmentationChain): the cascade reads x * (float)in_scale[item] - the pending rms-normalise factor of the previous
    // processor, applied in float32 exactly like the separate scale pass would - and the apply pass adds sum(y^2) of every item
    // to out_sumsq[item] (float64) so that the next rms-normalise needs no energy pass.  Both may be null.
    const double *in_scale = nullptr;
    double *out_sumsq = nullptr;
    double *out_in_sumsq = nullptr;      // the apply pass also leaves sum(x_raw^2) here (the first rms-normalise of a chain needs it: no energy pass over x)
};

template <bool APPLY, int NBANDS>
__global__ __launch_bounds__(64) void fx_biquad_chunk_kernel(BiquadChunkArgs a) {
    const long gid = (long)blockIdx.x * 64 + threadIdx.x;
    if constexpr (!APPLY) {          // the state pass clears the energy slots the apply pass adds to (no memset launch in front of the call)
        if (a.out_sumsq)
            for (long i = gid; i < (long)(a.n_seq / a.C) * MST_SUMSQ_SLOTS; i += (long)gridDim.x * 64) a.out_sumsq[i] = 0.0;
        if (a.out_in_sumsq)
            for (long i = gid; i < (long)(a.n_seq / a.C) * MST_SUMSQ_SLOTS; i += (long)gridDim.x * 64) a.out_in_sumsq[i] = 0.0;
    }
    if (gid >= (long)a.n_seq * a.nchunks) return;
    // lanes: channel fastest, then chunk, then item - neighbouring lanes read neighbouring samples of a frame
    const int c = (int)(gid % a.C);
    const int k = (int)((gid / a.C) % a.nchunks);
    const int item = (int)(gid / ((long)a.C * a.nchunks));
    const int seq = item * a.C + c;
    const long n_lo = (long)k * a.M, n_hi = (n_lo + a.M < a.L) ? n_lo + a.M : a.L;
    const float *xp = a.x + (size_t)item * a.L * a.C + c;
    float *yp = APPLY ? a.y + (size_t)item * a.L * a.C + c : nullptr;
    double z1[NBANDS], z2[NBANDS];
#pragma unroll
    for (int b = 0; b < NBANDS; ++b) {
        const double2 zz = APPLY ? *(const double2 *)(a.starts + ((size_t)seq * a.nchunks + k) * (2 * MST_MAX_BANDS) + 2 * b) : double2{0.0, 0.0};
        z1[b] = zz.x;
        z2[b] = zz.y;
    }
    const float sf = a.in_scale ? (float)a.in_scale[item] : 1.0f;
    double ss = 0.0, ssx = 0.0;
    auto step = [&](float xi) {
        if (APPLY) ssx += (double)(xi * xi);          // float32 square, float64 sum: the arithmetic of fx_sumsq_kernel
        double v = (double)(xi * sf);
#pragma unroll
        for (int b = 0; b < NBANDS; ++b) v = fx_biquad_band(v, z1[b], z2[b], a.coef[b]);
        const float out = (float)v;
        if (APPLY) ss += (double)out * (double)out;
        return out;
    };
    // full batches of 16 steps without per-element predicates (predicated loads make hipcc drain vmcnt(0) per element), the
    // next batch's loads in flight behind the current one's arithmetic; the ragged tail of the last chunk step by step
    constexpr int NB = 16;
    const long nfull = (n_hi - n_lo) / NB;
    float nx[NB];
    if (nfull > 0) {
#pragma unroll
        for (int i = 0; i < NB; ++i) nx[i] = xp[(n_lo + i) * a.C];
    }
    for (long bt = 0; bt < nfull; ++bt) {
        float xin[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) xin[i] = nx[i];
        const long nn = n_lo + ((bt + 1 < nfull) ? (bt + 1) * NB : bt * NB);      // last batch: harmless reload
#pragma unroll
        for (int i = 0; i < NB; ++i) nx[i] = xp[(nn + i) * a.C];
        float o[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) o[i] = step(xin[i]);
        if (APPLY) {
            if (a.C == 2) {
                // stereo: the two channels of a chunk are neighbouring lanes in lock-step.  They trade halves so that each owns two whole
                // frames of every four (even lane: frames f, f + 1; odd lane: f + 2, f + 3) and stores them as one 16-byte piece - the
                // pair writes 32 contiguous bytes per instruction instead of 8: a quarter of the L2 write transactions
                const bool odd = c != 0;
                float *fp = a.y + ((size_t)item * a.L + n_lo + bt * NB) * 2;
#pragma unroll
                for (int f = 0; f < NB; f += 4) {
                    const float ta = mst_lane_swap(odd ? o[f] : o[f + 2]), tb = mst_lane_swap(odd ? o[f + 1] : o[f + 3]);
                    const float4 v = odd ? make_float4(ta, o[f + 2], tb, o[f + 3]) : make_float4(o[f], ta, o[f + 1], tb);
                    *(float4 *)(fp + (f + (odd ? 2 : 0)) * 2) = v;
                }
            } else {
#pragma unroll
                for (int i = 0; i < NB; ++i) yp[(n_lo + bt * NB + i) * a.C] = o[i];
            }
        }
    }
    for (long n = n_lo + nfull * NB; n < n_hi; ++n) {
        const float out = step(xp[n * a.C]);
        if (APPLY) yp[n * a.C] = out;
    }
    if (!APPLY) {
#pragma unroll
        for (int b = 0; b < NBANDS; ++b) {
            *(double2 *)(a.ends + ((size_t)seq * a.nchunks + k) * (2 * MST_MAX_BANDS) + 2 * b) = double2{z1[b], z2[b]};
        }
    } else {
        if (a.out_sumsq) atomicAdd(&a.out_sumsq[item * MST_SUMSQ_SLOTS + (k & (MST_SUMSQ_SLOTS - 1))], ss);     // spread over the slots: few atomics per address
        if (a.out_in_sumsq) atomicAdd(&a.out_in_sumsq[item * MST_SUMSQ_SLOTS + (k & (MST_SUMSQ_SLOTS - 1))], ssx);
    }
}

// Pass 1 without the recursion (round 5).  The zero-state end state of a chunk is LINEAR in its M input samples:
//     e = sum_n h_(M-1-n) x[n],   h_m = A^m B = the cascade's state m steps after a unit impulse went in
// - 2 * n_bands independent multiply-add chains per sample instead of the cascade's 5 * n_bands DEPENDENT operations (the recursion pass is
// bound by that chain: one wave per SIMD, ~400 clocks per sample at five bands).  htab[m][S] = h_m comes from the host (float64, the same
// recursion run on an impulse; cached per coefficient set, mst_api.hip); a workgroup stages it through LDS in segments of TS samples and every
// lane reads its row as broadcast ds_read_b128s.  Same sum as the recursion up to float64 rounding (the terms are added in time order, not
// nested through the states).  The last chunk of a sequence, when it is short, has no successor: its end state is not needed (zeros).
template <int NBANDS>
__global__ __launch_bounds__(256) void fx_biquad_ends_kernel(BiquadChunkArgs a, const double *__restrict__ htab) {
    // FOUR lanes per (sequence, chunk), a quarter of the chunk's samples each (M is a multiple of 16): 4 x the waves of one lane per chunk - at
    // 61 696 chunks one lane each is one wave per SIMD, whose load latency, LDS reads and multiply-adds then run one after the other (measured
    // 49 us like the recursion pass it replaced).  Lanes: quarter fastest (the four partial sums meet by two lane exchanges), then channel, chunk, item.
    constexpr int S = 2 * NBANDS, TS = 32, NB = 4, PARTS = 4;          // TS = samples per quarter and table segment
    __shared__ __attribute__((aligned(16))) double tab[PARTS * TS * S];
    const long gid = (long)blockIdx.x * 256 + threadIdx.x, total = (long)a.n_seq * a.nchunks * PARTS;
    if (a.out_sumsq)          // the state pass clears the energy slots the apply pass adds to (no memset launch in front of the call)
        for (long i = gid; i < (long)(a.n_seq / a.C) * MST_SUMSQ_SLOTS; i += (long)gridDim.x * 256) a.out_sumsq[i] = 0.0;
    if (a.out_in_sumsq)
        for (long i = gid; i < (long)(a.n_seq / a.C) * MST_SUMSQ_SLOTS; i += (long)gridDim.x * 256) a.out_in_sumsq[i] = 0.0;
    const bool live = gid < total;
    const long gl = live ? gid : total - 1;                 // idle lanes of the last workgroup shadow a live one (they take part in the barriers)
    const int part = (int)(gl % PARTS);
    const int c = (int)((gl / PARTS) % a.C);
    const int k = (int)((gl / ((long)a.C * PARTS)) % a.nchunks);
    const int item = (int)(gl / ((long)a.C * PARTS * a.nchunks));
    const int seq = item * a.C + c;
    const int MQ = a.M / PARTS;                             // samples per quarter (a multiple of 4)
    const long n_lo = (long)k * a.M;
    const bool full = n_lo + a.M <= a.L;
    const float *xp = a.x + ((size_t)item * a.L + (full ? n_lo : 0) + (size_t)part * MQ) * a.C + c;      // a short chunk reads (and ignores) the sequence's first samples
    const float sf = a.in_scale ? (float)a.in_scale[item] : 1.0f;
    double acc[S];
#pragma unroll
    for (int j = 0; j < S; ++j) acc[j] = 0.0;
    for (int s0 = 0; s0 < MQ; s0 += TS) {
        const int cnt = MQ - s0 < TS ? MQ - s0 : TS;        // a multiple of 4
        __syncthreads();
        for (int e = threadIdx.x; e < PARTS * cnt * S; e += 256) {      // rows of the four quarters' samples s0 .. s0 + cnt - 1
            const int p = e / (cnt * S), r = e - p * cnt * S, i = r / S, j = r - i * S;
            tab[(p * TS + i) * S + j] = htab[(size_t)(a.M - 1 - (p * MQ + s0 + i)) * S + j];
        }
        __syncthreads();
        const double *mytab = tab + part * TS * S;
        for (int i0 = 0; i0 < cnt; i0 += NB) {
            float xin[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) xin[i] = xp[(size_t)(s0 + i0 + i) * a.C];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const double v = (double)(xin[i] * sf);
                const double *row = mytab + (i0 + i) * S;
#pragma unroll
                for (int j = 0; j < S; ++j) acc[j] = fma(row[j], v, acc[j]);
            }
        }
    }
    // the four quarters of a chunk sit in four neighbouring lanes (an aligned quad: 256 and 64 are multiples of 4)
#pragma unroll
    for (int j = 0; j < S; ++j) {
        acc[j] += __shfl_xor(acc[j], 1);
        acc[j] += __shfl_xor(acc[j], 2);
    }
    if (live && part == 0) {
#pragma unroll
        for (int b = 0; b < NBANDS; ++b)
            *(double2 *)(a.ends + ((size_t)seq * a.nchunks + k) * (2 * MST_MAX_BANDS) + 2 * b) = full ? double2{acc[2 * b], acc[2 * b + 1]} : double2{0.0, 0.0};
    }
}

// ------------------------------------------------------------------------------------------------
// STEREO chunks through LDS (round 5).  One lane per (channel, chunk) walks its chunk sample by sample; read straight from global memory
// that is 4 useful bytes per lane out of 32 different 128-byte lines per wave-instruction - measured: a state pass with next to no
// arithmetic (fx_biquad_ends_kernel) takes the same 50 us as the recursion it replaced; the passes are bound by the vector cache's line
// rate, not by float64.  Here a wave owns 32 consecutive chunks of stereo audio and moves them in SLABS of 16 frames: 16 stereo frames of a
// chunk are one 128-byte run (one aligned line when L * 8 is a multiple of 128, as for power-of-two segments), eight lanes fetch it as eight 16-byte pieces (4 wave
// loads per slab instead of 16, 8 lines per instruction instead of 32), the slab sits in LDS as 32 rows of 144 bytes and lane (c, chunk)
// reads its 16 samples from its row.  Four slabs' loads are in flight while the current one is used.
// Waves that hold the last chunk of a sequence (it may be short, and a line may run past the end of the buffer) guard every piece.
// Measured at 64 x [131072, 2], five bands (profiles/r05_fx_*): the state pass 49.4 us (recursion or dot products, one lane per chunk,
// straight from global memory) -> 49.6 (slabs, one in flight) -> 40.9 (four in flight).  The APPLY pass was built the same way (slabs in
// and out) and measured SLOWER than fx_biquad_chunk_kernel<true> (73-78 us against 62-64: 64 unrolled samples of a 25-coefficient recursion
// spill scalar registers): dropped, the apply pass keeps its lane-per-chunk loads.
// ------------------------------------------------------------------------------------------------
typedef float fx_f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // a 16-byte global access that is only dword aligned (odd L)
constexpr int FXS_TS = 16;          // frames per slab
constexpr int FXS_ROWB = 144;       // LDS row pitch in bytes: 128 + 16 (16-byte aligned pieces; rows 0 .. 15 start in distinct banks)
struct FxStereoRows {
    const float *src[4];            // frame 0 of this lane's four rows (row = (lane >> 3) + 8 i), advanced to the lane's 16-byte piece
    long room[4];                   // frames from the row's first frame to the end of its sequence; 0: no such row
    bool edge;                      // wave-uniform: some row may be short or absent -> guarded pieces
};
__device__ __forceinline__ FxStereoRows fx_stereo_rows(const float *x, long pair0, long npairs, int nchunks, int M, long L, int lane) {
    FxStereoRows r;
    r.edge = pair0 + 32 > npairs || (pair0 % nchunks) + 31 >= nchunks - 1;      // the wave holds the last chunk of a sequence
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long g = pair0 + (lane >> 3) + 8 * i;
        const bool ok = g < npairs;
        const long gg = ok ? g : npairs - 1;
        const long item = gg / nchunks;
        const long k = gg - item * nchunks;
        r.src[i] = x + ((size_t)item * L + (size_t)k * M) * 2 + 4 * (lane & 7);
        r.room[i] = ok ? L - k * M : 0;
    }
    return r;
}
__device__ __forceinline__ void fx_stereo_fetch(const FxStereoRows &r, int s, int lane, f32x4 (&v)[4]) {
    if (!r.edge) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *(const fx_f32x4u *)(r.src[i] + (size_t)s * 2);
    } else {
        const long f0 = s + 2 * (lane & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *p = r.src[i] + (size_t)s * 2;
            f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
            if (f0 + 1 < r.room[i]) {
                t = *(const fx_f32x4u *)p;
            } else if (f0 < r.room[i]) {
                t[0] = p[0];
                t[1] = p[1];
            }
            v[i] = t;
        }
    }
}
__device__ __forceinline__ void fx_stereo_to_lds(unsigned char *rows_lds, int lane, const f32x4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *(f32x4 *)(rows_lds + ((lane >> 3) + 8 * i) * FXS_ROWB + (lane & 7) * 16) = v[i];
}

// pass 1 for stereo audio: fx_biquad_ends_kernel's dot products (e = sum_n h_(M-1-n) x[n]) on slabs.  A workgroup = four waves = 128
// chunk pairs sharing the table segments (64 rows at a time).
template <int NBANDS>
__global__ __launch_bounds__(256) void fx_biquad_stereo_ends_kernel(BiquadChunkArgs a, const double *__restrict__ htab) {
    constexpr int S = 2 * NBANDS, TT = 64;
    __shared__ __attribute__((aligned(16))) double tab[TT * S];
    __shared__ __attribute__((aligned(16))) unsigned char slab[4][32 * FXS_ROWB];
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (a.out_sumsq)          // the state pass clears the energy slots the apply pass adds to (no memset launch in front of the call)
        for (long i = gid; i < (long)(a.n_seq / 2) * MST_SUMSQ_SLOTS; i += (long)gridDim.x * 256) a.out_sumsq[i] = 0.0;
    if (a.out_in_sumsq)
        for (long i = gid; i < (long)(a.n_seq / 2) * MST_SUMSQ_SLOTS; i += (long)gridDim.x * 256) a.out_in_sumsq[i] = 0.0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 1;
    const long npairs = (long)(a.n_seq / 2) * a.nchunks, pair0 = (long)blockIdx.x * 128 + wave * 32;
    const FxStereoRows rows = fx_stereo_rows(a.x, pair0, npairs, a.nchunks, a.M, a.L, lane);
    const long g = pair0 + (lane >> 1);
    const bool live = g < npairs;
    const long gg = live ? g : npairs - 1, item = gg / a.nchunks, k = gg - item * a.nchunks;
    const bool full = (k + 1) * a.M <= a.L;                  // a short last chunk has no successor: its end state is not needed
    const float sf = a.in_scale ? (float)a.in_scale[item] : 1.0f;
    unsigned char *my = slab[wave];
    double acc[S];
#pragma unroll
    for (int j = 0; j < S; ++j) acc[j] = 0.0;
    // FOUR slabs in flight per wave (a register ring; TT = 4 slabs): with one slab the pass is bound by memory-level parallelism - 964
    // waves x 4 KB in flight = 1.3 TB/s, measured 50 us whatever the access pattern
    constexpr int D = TT / FXS_TS;
    f32x4 ring[D][4];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d * FXS_TS < a.M) fx_stereo_fetch(rows, d * FXS_TS, lane, ring[d]);
    for (int seg = 0; seg < a.M; seg += TT) {
        const int cnt = a.M - seg < TT ? a.M - seg : TT;    // a multiple of 16
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * S; e += 256) {
            const int i = e / S, j = e - i * S;
            tab[e] = htab[(size_t)(a.M - 1 - (seg + i)) * S + j];
        }
        __syncthreads();
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int s = seg + d * FXS_TS;
            if (s < a.M) {                                    // uniform
                __builtin_amdgcn_wave_barrier();              // every lane is done with the previous slab
                fx_stereo_to_lds(my, lane, ring[d]);
                if (s + TT < a.M) fx_stereo_fetch(rows, s + TT, lane, ring[d]);
                mst_wave_lds_fence();
                const float *mine = (const float *)(my + (lane >> 1) * FXS_ROWB) + c;
#pragma unroll
                for (int f = 0; f < FXS_TS; ++f) {
                    const double xv = (double)(mine[2 * f] * sf);
                    const double *row = tab + (d * FXS_TS + f) * S;
#pragma unroll
                    for (int j = 0; j < S; ++j) acc[j] = fma(row[j], xv, acc[j]);
                }
            }
        }
    }
    if (live) {
        const int seq = (int)item * 2 + c;
#pragma unroll
        for (int b = 0; b < NBANDS; ++b)
            *(double2 *)(a.ends + ((size_t)seq * a.nchunks + k) * (2 * MST_MAX_BANDS) + 2 * b) = full ? double2{acc[2 * b], acc[2 * b + 1]} : double2{0.0, 0.0};
    }
}

// pass 1 for stereo audio on the float64 matrix cores (round 5).  fx_biquad_stereo_ends_kernel reads the impulse-state table from LDS - five
// broadcast ds_read_b128 per sample and lane for ten multiply-adds: four waves keep the CU's LDS pipe busy ~2500 clocks per slab against 800
// clocks of float64 arithmetic (40 us for 0.34 GFLOP).  The end states are a matrix product,  E[state][column] = sum_n H[state][n] X[n][column]
// (columns = the (chunk, channel) sequences, n = the sample inside the chunk), and v_mfma_f64_16x16x4_f64 takes both operands from
// registers: per 16-frame slab and 16 columns four MFMAs, the table as A fragments (one float64 per lane and MFMA, packed by the host in
// fragment order and streamed from L2: 512 bytes per MFMA, the same for every wave), the samples as B fragments (one ds_read_b32 of the slab
// per MFMA, converted on the way).  A wave owns 32 chunk pairs = 64 columns = four accumulator tiles for the whole chunk.  The instruction
// runs at the float64 VECTOR rate - the gain is operand delivery, not arithmetic.  Same products, added in sample order like the table
// kernel's fma chains; states beyond 2 * n_bands are rows of zeros.
typedef double fx_f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fx_biquad_stereo_ends_mfma_kernel(BiquadChunkArgs a, const double *__restrict__ afrag) {
    constexpr int D = 4;                               // slabs in flight
    __shared__ __attribute__((aligned(16))) unsigned char slab[4][32 * FXS_ROWB];
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (a.out_sumsq)          // the state pass clears the energy slots the apply pass adds to (no memset launch in front of the call)
        for (long i = gid; i < (long)(a.n_seq / 2) * MST_SUMSQ_SLOTS; i += (long)gridDim.x * 256) a.out_sumsq[i] = 0.0;
    if (a.out_in_sumsq)
        for (long i = gid; i < (long)(a.n_seq / 2) * MST_SUMSQ_SLOTS; i += (long)gridDim.x * 256) a.out_in_sumsq[i] = 0.0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
    const long npairs = (long)(a.n_seq / 2) * a.nchunks, pair0 = (long)blockIdx.x * 128 + wave * 32;
    if (pair0 >= npairs) return;                       // uniform per wave (no workgroup barrier below)
    const FxStereoRows rows = fx_stereo_rows(a.x, pair0, npairs, a.nchunks, a.M, a.L, lane);
    unsigned char *my = slab[wave];
    // column n of tile q = channel n & 1 of pair pair0 + 8 q + (n >> 1): its item's pending scale factor, its record
    float sfq[4];
    long rec[4];          // index of the column's record in `ends`, or -1 (no such pair)
    bool fullq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long g = pair0 + 8 * q + (n >> 1);
        const bool live = g < npairs;
        const long gg = live ? g : npairs - 1, item = gg / a.nchunks, k = gg - item * a.nchunks;
        sfq[q] = a.in_scale ? (float)a.in_scale[item] : 1.0f;
        rec[q] = live ? ((item * 2 + (n & 1)) * a.nchunks + k) * (2 * MST_MAX_BANDS) : -1;
        fullq[q] = (k + 1) * a.M <= a.L;                 // a short last chunk has no successor: its end state is not needed
    }
    fx_f64x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = fx_f64x4{0.0, 0.0, 0.0, 0.0};
    f32x4 ring[D][4];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d * FXS_TS < a.M) fx_stereo_fetch(rows, d * FXS_TS, lane, ring[d]);
    const double *af = afrag + lane;
    // the lane's sample inside a slab row: frame 4 kk + kq, channel n & 1; its row inside tile q: 8 q + (n >> 1)
    const unsigned char *bx = my + (n >> 1) * FXS_ROWB + (2 * kq + (n & 1)) * 4;
    // the A fragments of a slab are requested two slabs ahead, IN FRONT of that trip's sample fetch: the memory counter retires in order, so a
    // wait for them never waits for the four slabs of samples requested behind them
    const int nslab = a.M / FXS_TS;
    double av[3][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) av[t][kk] = af[(size_t)((t < nslab ? t : nslab - 1) * 4 + kk) * 64];
    for (int s0 = 0; s0 < a.M; s0 += 12 * FXS_TS) {         // 12 slabs per trip: the sample ring (4) and the fragment ring (3) both come round
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int s = s0 + u * FXS_TS, sb = s / FXS_TS;
            if (s < a.M) {                                    // uniform
                __builtin_amdgcn_wave_barrier();              // every lane is done with the previous slab
                fx_stereo_to_lds(my, lane, ring[u % D]);
                {
                    const int sn = sb + 2 < nslab ? sb + 2 : nslab - 1;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) av[(u + 2) % 3][kk] = af[(size_t)(sn * 4 + kk) * 64];
                }
                if (s + D * FXS_TS < a.M) fx_stereo_fetch(rows, s + D * FXS_TS, lane, ring[u % D]);
                mst_wave_lds_fence();
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float xv = *(const float *)(bx + 8 * q * FXS_ROWB + 32 * kk);
                        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u % 3][kk], (double)(xv * sfq[q]), acc[q], 0, 0, 0);
                    }
            }
        }
    }
    // D: column n, state kq + 4 r in register r
    const int S = 2 * a.n_bands;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (rec[q] >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = kq + 4 * r;
                if (j < S) a.ends[rec[q] + j] = fullq[q] ? acc[q][r] : 0.0;
            }
        }
    }
}

// pass 2 for stereo audio on slabs (round 5, second attempt).  With one lane per (channel, chunk) reading and writing global memory directly
// (fx_biquad_chunk_kernel<true>) the pass takes the same ~50 us from half a wave per SIMD to four (tools/micro/eq_apply_variants.hip): every
// wave-load touches 32 lines of 128 bytes and comes back for each of them sixteen times - the lines do not survive in the vector cache, the
// pass is bound by L2 -> L1 line traffic (16 x the bytes), not by float64.  Here the chunks move like in the state pass: 16-frame slabs, one
// 128-byte run per chunk and slab fetched as eight 16-byte pieces, four slabs in flight, a lane reads its 16 samples from its LDS row; the 16
// outputs go back through the same rows and leave as 16-byte pieces.  (The first attempt - EXPERIMENTS.md D.4 - unrolled four slabs and
// spilled scalar registers; this one runs ONE slab per loop trip behind a scheduling fence.)  Same recursion, same arithmetic per sample as
// fx_biquad_chunk_kernel<true>: the same bits; the energy sums are added per chunk in the same order.
template <int NBANDS>
__global__ __launch_bounds__(256) void fx_biquad_stereo_apply_kernel(BiquadChunkArgs a) {
    constexpr int D = 4;                               // slabs in flight
    __shared__ __attribute__((aligned(16))) unsigned char slab[4][32 * FXS_ROWB];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 1;
    const long npairs = (long)(a.n_seq / 2) * a.nchunks, pair0 = (long)blockIdx.x * 128 + wave * 32;
    if (pair0 >= npairs) return;                       // uniform per wave (no workgroup barrier below)
    const FxStereoRows rows = fx_stereo_rows(a.x, pair0, npairs, a.nchunks, a.M, a.L, lane);
    const long g = pair0 + (lane >> 1);
    const bool live = g < npairs;
    const long gg = live ? g : npairs - 1, item = gg / a.nchunks, k = gg - item * a.nchunks;
    const long room = live ? a.L - k * a.M : 0;        // frames from this lane's chunk start to the end of its sequence
    const int seq = (int)item * 2 + c;
    const float sf = a.in_scale ? (float)a.in_scale[item] : 1.0f;
    unsigned char *my = slab[wave];
    double z1[NBANDS], z2[NBANDS];
#pragma unroll
    for (int b = 0; b < NBANDS; ++b) {
        const double2 zz = *(const double2 *)(a.starts + ((size_t)seq * a.nchunks + k) * (2 * MST_MAX_BANDS) + 2 * b);
        z1[b] = zz.x;
        z2[b] = zz.y;
    }
    // the output rows: the pieces this lane stores (row = (lane >> 3) + 8 i, its 16-byte piece), like fx_stereo_rows' sources
    float *dst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = a.y + (rows.src[i] - a.x);
    double ss = 0.0, ssx = 0.0;
    f32x4 ring[D][4];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d * FXS_TS < a.M) fx_stereo_fetch(rows, d * FXS_TS, lane, ring[d]);
    // one slab: EDGE (a wave that holds the last chunk of a sequence - it may be short) guards every sample and every piece; the other
    // waves run the plain form.  The guards are selects, not branches: a branch per sample splits the unrolled recursion into basic blocks
    auto one_slab = [&](auto EDGE, f32x4 (&rg)[4], int s) {
        constexpr bool edge = decltype(EDGE)::value;
        __builtin_amdgcn_wave_barrier();              // every lane is done with the previous slab (its output pieces have been read)
        fx_stereo_to_lds(my, lane, rg);
        if (s + D * FXS_TS < a.M) fx_stereo_fetch(rows, s + D * FXS_TS, lane, rg);
        mst_wave_lds_fence();
        float *mine = (float *)(my + (lane >> 1) * FXS_ROWB) + c;
        float o[FXS_TS];
#pragma unroll
        for (int f = 0; f < FXS_TS; ++f) {
            const float xi = mine[2 * f];
            const bool in = !edge || s + f < room;
            const double sq = (double)(xi * xi);      // float32 square, float64 sum: the arithmetic of fx_sumsq_kernel
            ssx += in ? sq : 0.0;
            double v = (double)(xi * sf);
#pragma unroll
            for (int b = 0; b < NBANDS; ++b) v = fx_biquad_band(v, z1[b], z2[b], a.coef[b]);
            o[f] = (float)v;
            const double oq = (double)o[f] * (double)o[f];
            ss += in ? oq : 0.0;
        }
        __builtin_amdgcn_wave_barrier();              // every lane has read its 16 inputs
#pragma unroll
        for (int f = 0; f < FXS_TS; ++f) mine[2 * f] = o[f];
        mst_wave_lds_fence();
        // the slab leaves as 16-byte pieces: lane (row, piece) like the fetch
        const long f0 = s + 2 * (lane & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = *(const f32x4 *)(my + ((lane >> 3) + 8 * i) * FXS_ROWB + (lane & 7) * 16);
            float *p = dst[i] + (size_t)s * 2;
            if (!edge || f0 + 1 < rows.room[i]) {
                *(fx_f32x4u *)p = v;
            } else if (f0 < rows.room[i]) {
                p[0] = v[0];
                p[1] = v[1];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s0 = 0; s0 < a.M; s0 += D * FXS_TS) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int s = s0 + d * FXS_TS;
            if (s < a.M) {                                    // uniform
                if (rows.edge) one_slab(std::true_type{}, ring[d], s);
                else one_slab(std::false_type{}, ring[d], s);
            }
        }
    }
    if (live) {
        if (a.out_sumsq) atomicAdd(&a.out_sumsq[item * MST_SUMSQ_SLOTS + (k & (MST_SUMSQ_SLOTS - 1))], ss);     // spread over the slots: few atomics per address
        if (a.out_in_sumsq) atomicAdd(&a.out_in_sumsq[item * MST_SUMSQ_SLOTS + (k & (MST_SUMSQ_SLOTS - 1))], ssx);
    }
}

// s_0 = 0 ; s_k = A^M s_{k-1} + e_{k-1}: a first-order linear recurrence over the chunks with a matrix coefficient - scanned in
// parallel.  One workgroup of NB = 256 or 512 threads per sequence; a block of NB - 1 chunks at a time: element 0 is the carry (the
// start state of the block's first chunk), element i > 0 the zero-state end state of chunk i - 1, and a Hillis-Steele scan
//     t_i += (A^M)^(2^l) t_{i - 2^l}      l = 0 .. log2(NB) - 1
// leaves t_i = the true start state of chunk i (t_(NB-1) = the next block's carry).  The powers (A^M)^(2^l) come from
// the host (squared up from A^M, float64).  128 sequences x 128 chunks: 8 matrix-vector products of depth per
// sequence instead of 128 (round 1's kernel ran one LANE per sequence: 2 workgroups, 88 us); a 3-minute stem (7 752 chunks) is
// 31 blocks instead of 7 752 serial steps.  AM is [S][S] row-major with S = 2 * n_bands, state order (z1, z2) per band; ends /
// starts are laid out [sequence][chunk][16 states]: one 128-byte record per chunk, so that the scan's element i reads / writes
// record i (the first layout, [chunk][state][sequence], cost the scan 19 of its 39 us in 8-byte accesses 16 KB apart).
#define MST_BIQUAD_LEVELS 9                                                   // log2 of the largest scan block

// pm[l] = (A^M)^(2^l), l = 0 .. 8: the scan kernel reads the matrix elements with scalar loads (uniform addresses) and feeds them to v_fma_f64
// as SGPR operands - its first version kept them in LDS and spent its time on 100 broadcast ds_read_b64 per thread and level (40 us for 482
// chunks of 128 sequences).  The powers come from the host with the impulse-state table (mst_api.hip biquad_impulse_table, cached per
// coefficient set; until round 5 a one-workgroup kernel squared them up on the device in front of every scan).
template <int NBANDS, int NB>
__global__ __launch_bounds__(NB) void fx_biquad_scan_kernel(const double *ends, double *starts, const double *__restrict__ pm, int n_seq,
                                                           int nchunks) {
    constexpr int SM = 2 * MST_MAX_BANDS, S = 2 * NBANDS, NL = NB == 512 ? 9 : 8;      // table row stride / live states / levels
    static_assert(NB == 256 || NB == 512, "elements per block");
    static_assert(NL <= MST_BIQUAD_LEVELS, "powers available");
    __shared__ double st[2][S][NB];                  // state-major: neighbouring elements are neighbouring doubles
    const int seq = blockIdx.x, i = threadIdx.x;
    // every level reads its own 100 matrix elements with scalar loads exactly once: all of them cold misses (5 k clocks of waiting per
    // level).  Touch the whole table first - one line of 64 bytes per load, all in flight together - so that the levels hit the scalar cache.
    {
        double warm = 0.0;
#pragma unroll
        for (int k = 0; k < NL * S * S; k += 8) warm += pm[k];
        if (warm == 1.2345e301) starts[0] = warm;      // never true; keeps the loads
    }
    double carry[S];
#pragma unroll
    for (int j = 0; j < S; ++j) carry[j] = 0.0;
    for (int k0 = 0; k0 < nchunks; k0 += NB - 1) {
        // element i: the carry (i = 0) or the zero-state end state of chunk k0 + i - 1
        double t[S];
        const int kc = k0 + i - 1;
        const bool have = i > 0 && kc < nchunks;
        // never a predicated load (hipcc branches around it and drains vmcnt(0) behind each: ten serial memory latencies per block)
        const double2 *rec = (const double2 *)(ends + ((size_t)seq * nchunks + (have ? kc : 0)) * SM);
#pragma unroll
        for (int j = 0; j < S; j += 2) {
            const double2 v = rec[j / 2];
            t[j] = v.x;
            t[j + 1] = v.y;
        }
#pragma unroll
        for (int j = 0; j < S; ++j) {
            t[j] = i == 0 ? carry[j] : (have ? t[j] : 0.0);
            st[0][j][i] = t[j];
        }
        __syncthreads();
        int cur = 0;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int d = 1 << l;
            if (i >= d) {
                double u[S];
#pragma unroll
                for (int j = 0; j < S; ++j) u[j] = st[cur][j][i - d];
#pragma unroll
                for (int r = 0; r < S; ++r) {
                    double acc = t[r];
#pragma unroll
                    for (int j = 0; j < S; ++j) acc = fma(pm[(l * S + r) * S + j], u[j], acc);      // uniform address: a scalar load
                    t[r] = acc;
                }
            }
#pragma unroll
            for (int j = 0; j < S; ++j) st[cur ^ 1][j][i] = t[j];
            __syncthreads();
            cur ^= 1;
        }
        // t = start state of chunk k0 + i (i < NB - 1); the last element is the next block's carry
        if (i < NB - 1 && k0 + i < nchunks) {
            double2 *out = (double2 *)(starts + ((size_t)seq * nchunks + k0 + i) * SM);
#pragma unroll
            for (int j = 0; j < S; j += 2) out[j / 2] = double2{t[j], t[j + 1]};
        }
#pragma unroll
        for (int j = 0; j < S; ++j) carry[j] = st[cur][j][NB - 1];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// compressor_process (:529-587) as called by Compressor.process (:637-649, makeup 0).  One WAVE per
// (item, channel) sequence, 64 samples per step: the log-domain gain computer (log10) and the gain
// application (pow) run lane-parallel; only the branchy one-pole attack/release smoother is serial, walked
// in sample order by broadcasting x_l lane by lane (every lane carries the same recurrence state).
// ------------------------------------------------------------------------------------------------
struct CompArgs {
    const float *x;
    float *y;
    int n_seq, C;
    long L;
    double threshold, ratio, alpha_att, alpha_rel, makeup;
    // parameter-grid form (the normaliser's threshold x ratio search, utils_data_normalization.py:384-398): item i runs with
    // (thr_items[i], ratio_items[i]); with shared_x every item reads the SAME input signal (item 0 of x)
    const double *thr_items = nullptr, *ratio_items = nullptr;
    int shared_x = 0;
    // chain fusion: x is read as x * (float)in_scale[item]; the apply pass adds sum(y^2) per item to out_sumsq (both may be null)
    const double *in_scale = nullptr;
    double *out_sumsq = nullptr;
    double *out_ms = nullptr;      // stereo: the apply pass also adds sum((l + r)^2), sum((l - r)^2) per item to [item][MST_SUMSQ_SLOTS][2] (for the imager)
};

__global__ __launch_bounds__(256) void fx_compressor_kernel(CompArgs a) {
    const int lane = threadIdx.x & 63;
    const int seq = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seq >= a.n_seq) return;   // wave-uniform
    const int item = seq / a.C, c = seq % a.C;
    const float *xp = a.x + (size_t)item * a.L * a.C + c;
    float *yp = a.y + (size_t)item * a.L * a.C + c;
    double prev = 0.0;            // yL_prev is forced to 0 on entry (:553)
    for (long n0 = 0; n0 < a.L; n0 += 64) {
        const long n = n0 + lane;
        const double xv = (n < a.L) ? (double)xp[n * a.C] : 0.0;
        const double ax = fabs(xv);
        const double xg = (ax < 0.000001) ? -120.0 : 20.0 * log10(ax);
        double yg = 0.0;          // ratio == 1: neither branch assigns y_g (:564-573)
        if (a.ratio > 1.0)
            yg = (xg >= a.threshold) ? a.threshold + (xg - a.threshold) / a.ratio : xg;
        else if (a.ratio < 1.0)
            yg = (xg <= a.threshold) ? a.threshold + (xg - a.threshold) / (1.0 / a.ratio) : xg;
        const double xl = xg - yg;
        double yl = 0.0;
        const int cnt = (a.L - n0) < 64 ? (int)(a.L - n0) : 64;
        for (int i = 0; i < cnt; ++i) {
            const double v = __shfl(xl, i);
            if (v > prev)
                prev = a.alpha_att * prev + (1.0 - a.alpha_att) * v;
            else
                prev = a.alpha_rel * prev + (1.0 - a.alpha_rel) * v;
            if (lane == i) yl = prev;
        }
        if (n < a.L) yp[n * a.C] = (float)(xv * pow(10.0, (a.makeup - yl) / 20.0));
    }
}

// ------------------------------------------------------------------------------------------------
// The same compressor split by dependency structure, for signals too short for the time-parallel form below (fewer than four
// 32-step chunks; 8 bytes of scratch per sample):
//   fx_comp_gain_kernel    every sample in parallel: x_l = x_g - y_g (log10 + static curve), float64
//   fx_comp_smooth_kernel  one lane per sequence: ONLY the branchy one-pole recursion (2 FMAs + compare + select per
//                          sample), y_l written over x_l
//   fx_comp_apply_kernel<false>  every sample in parallel: y = x * 10^((makeup - y_l) / 20)
// Same float64 arithmetic per sample; the serial part shrinks to the recursion itself.
// Scratch layout: xl[n][seq] (TIME-major): the 64 lanes of the serial kernel - one sequence each - read and write 512
// contiguous bytes per step.  (Sequence-major, every lane touched its own cache line: 2 x 64 addresses per step through
// the texture addresser at one per clock = 140 clocks per step for a ~30-clock recursion, 7.6 ms per 128 x 131072 samples.)
// The two parallel kernels move 64 x 64 (time x sequence) tiles through LDS so that both their audio side
// ([item][n][c], time-contiguous) and their scratch side (sequence-contiguous) are coalesced.
// ------------------------------------------------------------------------------------------------
// log10 of a positive normal float32 value as a float64 (error ~1e-16 relative, like the library's log10 of the converted value, at
// a sixth of its instructions: the library routine carries double-double arithmetic for arguments it cannot know are float32).
// |x| = 2^e * m, m in [1, 2) with 23 mantissa bits; m_hi = the top 7 of them; r = (m - m_hi) / m_hi in [0, 2^-7) (m - m_hi is exact):
//     log10 |x| = e log10(2) + log10(m_hi) + log10(e) * log1p(r),   log1p(r) = r - r^2/2 + ... + r^7/7   (next term < 2e-18)
// tab[i] = log10(1 + i/128), tab[128 + i] = 1 / (1 + i/128): 2 KB of LDS filled by the workgroup with the library functions.
__device__ __forceinline__ void fx_log10_table_fill(double *tab, int tid, int nthreads) {
    for (int i = tid; i < 128; i += nthreads) {
        const double mh = 1.0 + (double)i * (1.0 / 128.0);
        tab[i] = log10(mh);
        tab[128 + i] = 1.0 / mh;
    }
}
__device__ __forceinline__ double fx_log10_f32(float ax, const double *tab) {
    const unsigned bits = __float_as_uint(ax);
    const int e = (int)(bits >> 23) - 127;
    const unsigned i = (bits >> 16) & 127u;
    const double m = (double)__uint_as_float((bits & 0x007fffffu) | 0x3f800000u);
    const double mh = (double)__uint_as_float((bits & 0x007f0000u) | 0x3f800000u);
    const double r = (m - mh) * tab[128 + i];
    double p = fma(r, 1.0 / 7.0, -1.0 / 6.0);
    p = fma(r, p, 1.0 / 5.0);
    p = fma(r, p, -1.0 / 4.0);
    p = fma(r, p, 1.0 / 3.0);
    p = fma(r, p, -0.5);
    const double l1p = fma(r * r, p, r);
    return fma(l1p, 0.43429448190325182765, fma((double)e, 0.30102999566398119521, tab[i]));
}

// x_l = x_g - y_g of one sample (:556-575).  thr / mul / mode are per sequence: mode 1 = compressor (ratio > 1, mul = 1 / ratio),
// 2 = expander (ratio < 1, mul = ratio: the reference divides by 1 / ratio), 0 = ratio == 1 (neither branch assigns y_g: it stays 0)
__device__ __forceinline__ double fx_comp_level_diff(float x, double thr, double mul, int mode, const double *tab) {
    const float axf = fabsf(x);
    const double xg = ((double)axf < 0.000001) ? -120.0 : 20.0 * fx_log10_f32(axf, tab);
    double yg = 0.0;
    if (mode == 1)
        yg = (xg >= thr) ? fma(xg - thr, mul, thr) : xg;
    else if (mode == 2)
        yg = (xg <= thr) ? fma(xg - thr, mul, thr) : xg;
    return xg - yg;
}

// the log10 table once per launch sequence (256 doubles in the caller's scratch): the time-parallel kernels copy it into LDS
__global__ __launch_bounds__(128) void fx_log10_table_kernel(double *tab) { fx_log10_table_fill(tab, threadIdx.x, 128); }

// the static curve of one sequence as fx_comp_level_diff takes it
struct FxCompCurve { double thr, mul; int mode; float sf; };
__device__ __forceinline__ FxCompCurve fx_comp_curve(const CompArgs &a, int item) {
    const double ratio = a.ratio_items ? a.ratio_items[item] : a.ratio;
    FxCompCurve c;
    c.thr = a.thr_items ? a.thr_items[item] : a.threshold;
    c.mode = ratio > 1.0 ? 1 : (ratio < 1.0 ? 2 : 0);
    c.mul = ratio > 1.0 ? 1.0 / ratio : ratio;
    c.sf = a.in_scale ? (float)a.in_scale[item] : 1.0f;
    return c;
}

// grid (ceil(L / 64), ceil(n_seq / 64)), 256 threads
__global__ __launch_bounds__(256) void fx_comp_gain_kernel(CompArgs a, double *xl) {
    __shared__ double t[64][65];
    __shared__ double tab[256];
    __shared__ double sthr[64], smul[64];
    __shared__ int smode[64];
    __shared__ float ssf[64];
    const long n0 = (long)blockIdx.x * 64;
    const int s0 = blockIdx.y * 64;
    fx_log10_table_fill(tab, threadIdx.x, 256);
    if (threadIdx.x < 64) {                               // the static curve of every sequence of the tile
        const int seq = s0 + threadIdx.x < a.n_seq ? s0 + threadIdx.x : a.n_seq - 1;
        const FxCompCurve c = fx_comp_curve(a, seq / a.C);
        sthr[threadIdx.x] = c.thr; smul[threadIdx.x] = c.mul; smode[threadIdx.x] = c.mode; ssf[threadIdx.x] = c.sf;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {                       // audio side: lanes run along time within one sequence
        const int idx = k * 256 + threadIdx.x, sl = idx >> 6, nl = idx & 63;
        const int seq = s0 + sl;
        const long n = n0 + nl;
        double v = 0.0;
        if (seq < a.n_seq && n < a.L)
            v = fx_comp_level_diff(a.x[((size_t)(a.shared_x ? 0 : seq / a.C) * a.L + n) * a.C + seq % a.C] * ssf[sl], sthr[sl], smul[sl],
                                   smode[sl], tab);
        t[nl][sl] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {                       // scratch side: lanes run along the sequences of one time step
        const int idx = k * 256 + threadIdx.x, nl = idx >> 6, sl = idx & 63;
        if (s0 + sl < a.n_seq && n0 + nl < a.L) xl[(size_t)(n0 + nl) * a.n_seq + s0 + sl] = t[nl][sl];
    }
}

__global__ __launch_bounds__(64) void fx_comp_smooth_kernel(CompArgs a, double *xl) {
    const int seq = blockIdx.x * 64 + threadIdx.x;
    const bool live = seq < a.n_seq;
    // y <- alpha y + (1 - alpha) x, alpha = attack coefficient when x > y else release, evaluated as y + c (x - y) with
    // c = 1 - alpha picked by the sign of d = x - y: three float64 operations per step (add, compare, fma) instead of five plus
    // 64-bit address arithmetic.  A float64 VALU op issues in 8 clocks on gfx950 and the steps are one dependent chain:
    // measured 140 clocks per step with per-lane cache lines, 99 with the time-major layout and the two-product form, 85 with
    // this one (computing both candidates and selecting afterwards is not faster: 88).  Same value up to float64 rounding.
    // Addresses are a uniform base + a 32-bit byte offset (scratch < 4 GiB is checked by the host).
    unsigned char *base = (unsigned char *)xl;
    const unsigned row = (unsigned)a.n_seq * 8u;                       // bytes per time step
    const unsigned off0 = (unsigned)(live ? seq : a.n_seq - 1) * 8u;   // idle lanes of the last wave shadow a live one, store nothing
    const double ca = 1.0 - a.alpha_att, cr = 1.0 - a.alpha_rel;
    double prev = 0.0;
    constexpr int NB = 16;
    const long nfull = a.L / NB;
    double nx[NB];
    if (nfull > 0) {
#pragma unroll
        for (int i = 0; i < NB; ++i) nx[i] = *(const double *)(base + (size_t)(off0 + (unsigned)i * row));
    }
    for (long bt = 0; bt < nfull; ++bt) {
        double v[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) v[i] = nx[i];
        const unsigned nn = (unsigned)((bt + 1 < nfull) ? (bt + 1) * NB : bt * NB);      // last batch: harmless reload
#pragma unroll
        for (int i = 0; i < NB; ++i) nx[i] = *(const double *)(base + (size_t)(off0 + (nn + (unsigned)i) * row));
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const double d = v[i] - prev;
            prev = fma(d > 0.0 ? ca : cr, d, prev);
            v[i] = prev;
        }
        if (live) {
#pragma unroll
            for (int i = 0; i < NB; ++i) *(double *)(base + (size_t)(off0 + ((unsigned)(bt * NB) + (unsigned)i) * row)) = v[i];
        }
    }
    for (long n = nfull * NB; n < a.L; ++n) {
        double *q = (double *)(base + (size_t)(off0 + (unsigned)n * row));
        const double d = *q - prev;
        prev = fma(d > 0.0 ? ca : cr, d, prev);
        if (live) *q = prev;
    }
}

// ------------------------------------------------------------------------------------------------
// The smoother, parallel in time.  One step is y <- f_x(y) = (x > y) ? aA y + cA x : aR y + cR x.  With aA <= aR (attack faster
// than release - every parameter range of the reference) f_x(y) = max(aA y + cA x, aR y + cR x): an increasing, convex,
// piecewise-linear map, and so is every composition F = f_xT o ... o f_x1 of a chunk - with exactly T + 1 linear pieces
// (aA > aR: the same with min / concave).  Two kernels replace the 131072 dependent steps per sequence, a third finishes:
//   fx_comp_map_kernel    one lane per (sequence, chunk of T steps): the pieces of the chunk's map F, sorted by value.
//   fx_comp_chain_kernel  one WAVE per sequence walks the chunks: lanes = pieces, y <- a_i y + b_i of the piece i that y falls in.
//   fx_comp_apply_kernel<FILL>  one lane per (sequence, chunk): the plain recursion inside the chunk from its true start value, on the
//                         way into the gain application.
// Exact arithmetic gives the serial result; in float64 the chunk start values differ from it by rounding (~1e-15 relative).
//
// What a chunk's map needs to carry.  A step x splits exactly one piece - the one whose range of VALUES contains x (f_x(x) = x) -
// sends the pieces below it through the attack branch and those above through the release branch.  So after n steps the piece at
// sorted position p has been through n - p attack steps and p release steps whatever the signal was: its slope is aA^(n-p) aR^p, a
// constant of the launch (CompMapArgs::slope).  F is continuous, so the pieces are fixed by the VALUES lb_p at which they meet plus the
// intercept b_0 of the lowest piece: the map kernel tracks and STORES only those (one float64 per piece: 272 bytes per chunk - the
// kernel is bound by writing its records); the chain kernel's helper waves rebuild intercepts b_p and crossing inputs u_p while the
// walker is busy:  u_1 = (lb_1 - b_0) / a_0,  u_(p+1) = u_p + (lb_(p+1) - lb_p) / a_p  (a prefix sum over the pieces),  b_p = lb_p - a_p u_p.
// A step on the sorted values is g_s = f_x(lb_s) (monotone: still sorted) followed by the insertion of x into the sorted list,
// new_s = max(g_(s-1), min(g_s, x)): five float64 operations per piece and step, no compare, no select, no data-dependent move.
// (Round 2's first version kept (a, b, lb) per piece and shifted them under exec masks: 4x the instructions, 2.5x the time.)
// ------------------------------------------------------------------------------------------------
#define MST_COMP_T 32                      // steps per chunk: T + 1 pieces
#define MST_COMP_NP (MST_COMP_T + 1)
#define MST_COMP_REC (MST_COMP_NP + 1)   // doubles per stored record: b_0, lb_1 .. lb_T (NEVER beyond the chunk's pieces), one pad
#define MST_COMP_NEVER 1e300

struct CompMapArgs {
    const double *log_tab; // fx_log10_table_kernel's 256 doubles
    double *maps;         // [n_seq][nchunks][MST_COMP_REC]  b_0, lb_1 .. lb_T per chunk
    double *ystart;       // [nchunks][n_seq]  smoother value at the start of each chunk
    int n_seq, nchunks;
    long L;
    double aA, aR;        // attack / release coefficients alpha
    int use_min;          // aA > aR: concave maps
    // slope of the piece at sorted position p and its reciprocal; [0]: a chunk of T steps, [1]: the last chunk when it is shorter
    // (0 beyond its pieces)
    double slope[2][MST_COMP_NP], inv_slope[2][MST_COMP_NP];
    // time slices (the host pipelines map -> chain -> apply over slices of whole batches, mst_api.hip compressor_run): the map launch covers
    // chunks chunk0 .. chunk0 + gridDim.x - 1, the chain launch batches batch0 .. batch1 - 1 (of MST_CHAIN_CB chunks); a chain launch with
    // batch0 > 0 starts from ycarry[seq] (what the launch before it left there) instead of yL_prev = 0, and every launch leaves its last value
    int chunk0 = 0, batch0 = 0, batch1 = 0;
    double *ycarry = nullptr;      // [n_seq]
};
#define MST_CHAIN_CB 32                    // chunks per batch of the chain kernel

// grid (nchunks, ceil(n_seq / 64)), 64 threads: lanes = sequences of one chunk (coalesced time-major loads).  Four waves per SIMD
// (<= 128 registers, 8.5 KB of LDS): the early steps of a chunk have few pieces and little to overlap within one wave.
template <bool USE_MIN>
__global__ __launch_bounds__(64) MST_WAVES_PER_SIMD(4) MST_HEAVY_UNROLL void fx_comp_map_kernel(CompMapArgs a, CompArgs ca) {
    constexpr int PASS = 16;                                        // doubles of every record that cross the LDS tile at a time (128 B)
    __shared__ double tr[64 * (PASS + 1)];
    __shared__ double tab[256];
    const int k = blockIdx.x + a.chunk0;
    const int seq = blockIdx.y * 64 + threadIdx.x;
    const bool live = seq < a.n_seq;
    const size_t sq = live ? seq : a.n_seq - 1;
    const double cA = 1.0 - a.aA, cR = 1.0 - a.aR;
    // the level differences x_l are computed here from the audio (a lane walks its own sequence: 32 frames = 256 contiguous bytes that
    // it shares with the lane of the other channel), not read from a float64 scratch that a separate pass would have to write
    for (int i = threadIdx.x; i < 256; i += 64) tab[i] = a.log_tab[i];
    const int item = (int)(sq / ca.C);
    const FxCompCurve cv = fx_comp_curve(ca, item);
    const float *xp = ca.x + ((size_t)(ca.shared_x ? 0 : item) * ca.L) * ca.C + sq % ca.C;
    __builtin_amdgcn_wave_barrier();
    double lb[MST_COMP_NP + 1];                                     // lb[1 .. t]: the values at which the pieces meet after t steps
    double b0 = 0.0;                                                // the lowest piece (always the attack branch); identity before step 1
#pragma unroll
    for (int t = 0; t < MST_COMP_T; ++t) {
        const long n = (long)k * MST_COMP_T + t;
        if (n < a.L) {                                              // uniform over the wave
            const double x = fx_comp_level_diff(xp[(size_t)n * ca.C] * cv.sf, cv.thr, cv.mul, cv.mode, tab);
            const double oA = cA * x, oR = cR * x;
            b0 = fma(a.aA, b0, oA);
            // downwards, in place: slot s reads the old slots s and s - 1.  g = f_x(old value) = max (convex) / min (concave) of the branches
            double m = x;                                           // min(g_s, x); above the top piece: x
#pragma unroll
            for (int s = t + 1; s >= 1; --s) {
                if (s > 1) {
                    const double v = lb[s - 1];
                    const double gA = fma(a.aA, v, oA), gR = fma(a.aR, v, oR);
                    const double g = USE_MIN ? fmin(gA, gR) : fmax(gA, gR);
                    lb[s] = fmax(g, m);
                    m = fmin(g, x);
                } else {
                    lb[1] = m;
                }
            }
        }
    }
    const long left = a.L - (long)k * MST_COMP_T;
    const int nsteps = left < MST_COMP_T ? (int)left : MST_COMP_T;  // uniform; >= 1
    // the record (b_0, lb_1 .. lb_T, pad) leaves in passes of 16 doubles through a [lane][16 + 1] tile: every store instruction of the
    // wave writes 128 contiguous bytes of four records
    const int nlive = a.n_seq - blockIdx.y * 64 < 64 ? a.n_seq - blockIdx.y * 64 : 64;
    double *row = tr + threadIdx.x * (PASS + 1);
    double *mbase = a.maps + ((size_t)(blockIdx.y * 64) * a.nchunks + k) * MST_COMP_REC;
    auto pass = [&](auto J) {                                       // two flat loops: both unroll early, lb stays in registers
        constexpr int p0 = decltype(J)::value * PASS, p1 = p0 + PASS < MST_COMP_REC ? p0 + PASS : MST_COMP_REC;
        constexpr int cnt = p1 - p0;                                // doubles in this pass
#pragma unroll
        for (int p = p0; p < p1; ++p) row[p - p0] = p == 0 ? b0 : ((p <= nsteps && p <= MST_COMP_T) ? lb[p <= MST_COMP_T ? p : 0] : MST_COMP_NEVER);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < cnt; ++i) {                             // 64 records x cnt doubles, lane-linear over (record, double)
            const int idx = i * 64 + threadIdx.x, r = idx / cnt, c = idx % cnt;
            if (r < nlive) mbase[(size_t)r * a.nchunks * MST_COMP_REC + p0 + c] = tr[r * (PASS + 1) + c];
        }
        __builtin_amdgcn_wave_barrier();
    };
    static_assert(MST_COMP_REC <= 3 * PASS, "three passes cover the record");
    pass(std::integral_constant<int, 0>{});
    pass(std::integral_constant<int, 1>{});
    pass(std::integral_constant<int, 2>{});
}

// grid n_seq, 384 threads: six waves per sequence.  Wave 0 walks the chunks (lanes = pieces); waves 1, 2, 3 and 5 run one batch of CB
// chunks ahead of it (a quarter of a batch each; with two of them the walker waited for its entries: 202 us against 153 alone; wave 4
// would share the walker's SIMD and only keeps the barriers company) and turn the stored records (b_0, lb_1 .. lb_T) into per-piece entries (b_p, u_p) in LDS: two chunks at
// a time, one per half-wave, lane = piece, the crossing inputs by a 32-lane DPP prefix sum (mst_half_prefix_sum_f64).  A chunk step FINDS
// the piece that applies to y: piece i applies from its crossing input u_i on, and the pieces sit in the walker's lanes in DESCENDING
// order, so the lowest lane with u <= y holds it - v_cmpx straight into EXEC, v_readfirstlane of that lane's a_i y + b_i
// (mst_wave_first_ge): one float64 op and one lane read on the dependent chain per chunk, instead of fma -> compare -> s_bcnt1 ->
// v_readlane (2.4x the latency) or a six-stage float64 wave reduction.  Near a crossing the two neighbouring pieces agree to rounding,
// so a test decided by rounding picks an equally valid piece.  The slope is a constant of the lane.  The 32 steps of a batch are
// unrolled: entries are prefetched three chunks ahead with immediate offsets, the chunk start values are parked in lane c of a
// register (v_writelane) and stored once per batch.
#define MST_CHAIN_HELPERS 4
#define MST_CHAIN_THREADS 384
__global__ __launch_bounds__(MST_CHAIN_THREADS) void fx_comp_chain_kernel(CompMapArgs a) {
    constexpr int CB = MST_CHAIN_CB, REC = MST_COMP_REC, NPE = MST_COMP_NP + 1, ENT = 2 * NPE;      // chunks per batch, doubles per record / per chunk of entries
    constexpr int NH = MST_CHAIN_HELPERS, HB = CB / NH;               // helper waves; chunks per helper wave and batch
    constexpr int NLD = (HB * REC / 2 + 63) / 64;                     // 16-byte loads per helper lane per batch
    static_assert(REC % 2 == 0 && HB % 2 == 0, "records are whole 16-byte units; a helper takes its chunks two at a time");
    __shared__ __attribute__((aligned(16))) double raw[NH][HB * REC];
    __shared__ __attribute__((aligned(16))) double cooked[2][CB * ENT];      // per (chunk, piece): b, u; entry NP is never selected
    static_assert(NH == 4, "helper waves 1, 2, 3, 5");
    const int seq = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hw = wave == 5 ? 3 : (wave > 0 ? wave - 1 : 0);
    const bool helper = wave != 0 && wave != 4;
    const double2 *m = (const double2 *)(a.maps + (size_t)seq * a.nchunks * REC);
    const size_t total = (size_t)a.nchunks * REC / 2;
    const int nbatch = a.batch1;                                      // this launch walks batches batch0 .. batch1 - 1
    // ---- waves 1, 2: global -> registers -> raw records in LDS -> entries
    const int hp = (lane & 31) + 1;                                   // the piece this helper lane rebuilds (1 .. T)
    const double sl_full = a.slope[0][hp], isl_full = a.inv_slope[0][hp - 1], sl_last = a.slope[1][hp], isl_last = a.inv_slope[1][hp - 1];
    double rx[NLD], ry[NLD];                                          // (plain doubles: an array of double2 stays in scratch memory)
    auto load = [&](int bt) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const size_t e = ((size_t)(bt < nbatch ? bt : nbatch - 1) * CB + hw * HB) * (REC / 2) + (size_t)i * 64 + lane;
            const double2 v = m[e < total ? e : total - 1];
            rx[i] = v.x;
            ry[i] = v.y;
        }
    };
    // the records of batch bt leave the registers for LDS, the loads of batch bt + 1 are issued at once (a whole batch of walking and
    // rebuilding covers their latency), then the entries of batch bt are rebuilt
    auto cook = [&](int buf, int bt) {
        double2 *dst = (double2 *)raw[hw];
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (i * 64 + lane < HB * REC / 2) dst[i * 64 + lane] = double2{rx[i], ry[i]};
        load(bt + 1);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < HB / 2; ++it) {
            const int c = 2 * it + (lane >> 5);                       // chunk of this half-wave within the helper's share
            const double *rec = raw[hw] + c * REC;
            const double lbp = rec[hp], lbm = rec[hp - 1];            // rec[0] = b_0
            const bool valid = lbp < 0.5 * MST_COMP_NEVER;            // the piece exists (the last chunk may be short)
            const bool shortc = (long)(bt * CB + hw * HB + c + 1) * MST_COMP_T > a.L;
            const double d = valid ? (lbp - lbm) * (shortc ? isl_last : isl_full) : 0.0;
            const double u = mst_half_prefix_sum_f64(d);
            double *q = cooked[buf] + ((hw * HB + c) * NPE + hp) * 2;
            q[0] = valid ? fma(-(shortc ? sl_last : sl_full), u, lbp) : 0.0;
            q[1] = valid ? u : MST_COMP_NEVER;
            if (hp == 1) {                                            // the lowest piece: reached by every y
                q[-2] = lbm;
                q[-1] = -MST_COMP_NEVER;
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    if (helper) {
        for (int i = lane; i < 2 * HB; i += 64) {                     // the entry the lanes without a piece read: written once
            double *q = &cooked[i / HB][((hw * HB + i % HB) * NPE + MST_COMP_NP) * 2];
            q[0] = 0.0;
            q[1] = MST_COMP_NEVER;
        }
        load(a.batch0);
        cook(0, a.batch0);
    }
    __syncthreads();
    // yL_prev = 0 on entry (common_audioeffects.py:553); a later time slice continues from the value the slice before it left
    MstUniformF64 yu = mst_wave_read_u64(a.batch0 > 0 ? a.ycarry[seq] : 0.0, 0);
    const int pl = lane < MST_COMP_NP ? MST_COMP_NP - 1 - lane : MST_COMP_NP;   // descending; lanes without a piece read the "never" entry
    const double a_full = pl < MST_COMP_NP ? a.slope[0][pl] : 0.0, a_last = pl < MST_COMP_NP ? a.slope[1][pl] : 0.0;
    for (int bt = a.batch0; bt < nbatch; ++bt) {
        const int cur = (bt - a.batch0) & 1;
        if (helper) {
            if (bt + 1 < nbatch) {
                cook(cur ^ 1, bt + 1);
            }
        } else if (wave == 0) {
            const int nc = a.nchunks - bt * CB < CB ? a.nchunks - bt * CB : CB;
            struct Piece { double pb, u; };
            const double *base = cooked[cur] + 2 * pl;
            auto fetch = [&](int c) {
                const double2 q = *(const double2 *)(base + ENT * (c < CB ? c : CB - 1));
                return Piece{q.x, q.y};
            };
            double keep = 0.0;                                      // lane c holds the start value of chunk c
            if ((long)(bt + 1) * CB * MST_COMP_T <= a.L) {          // a whole batch of whole chunks: 32 steps in one basic block, y stays in SGPRs
                Piece p[4];
                p[0] = fetch(0); p[1] = fetch(1); p[2] = fetch(2);
                yu = mst_wave_uniform(yu);
#define MST_CHAIN_STEP(c)                                                                                              \
    {                                                                                                                  \
        p[((c) + 3) & 3] = fetch((c) + 3);                                                                             \
        keep = mst_wave_park_f64<(c)>(keep, yu);                                                                       \
        const Piece &q = p[(c) & 3];                                                                                   \
        yu = mst_wave_first_ge(yu, q.u, fma(a_full, yu.value(), q.pb));                                                \
    }
#define MST_CHAIN_STEP8(c) MST_CHAIN_STEP(c) MST_CHAIN_STEP((c) + 1) MST_CHAIN_STEP((c) + 2) MST_CHAIN_STEP((c) + 3) \
    MST_CHAIN_STEP((c) + 4) MST_CHAIN_STEP((c) + 5) MST_CHAIN_STEP((c) + 6) MST_CHAIN_STEP((c) + 7)
                static_assert(CB == 32, "four groups of eight steps");
                MST_CHAIN_STEP8(0) MST_CHAIN_STEP8(8) MST_CHAIN_STEP8(16) MST_CHAIN_STEP8(24)
#undef MST_CHAIN_STEP8
#undef MST_CHAIN_STEP
            } else {                                                // the last batch: ragged, or its last chunk is short
                for (int c = 0; c < nc; ++c) {
                    const Piece q = fetch(c);
                    const double y = yu.value();
                    keep = lane == c ? y : keep;
                    const double sa = (long)(bt * CB + c + 1) * MST_COMP_T > a.L ? a_last : a_full;
                    yu = mst_wave_first_ge(yu, q.u, fma(sa, y, q.pb));
                }
            }
            if (lane < nc) a.ystart[(size_t)(bt * CB + lane) * a.n_seq + seq] = keep;
            if (bt + 1 == nbatch && lane == 0) a.ycarry[seq] = yu.value();
        }
        __syncthreads();
    }
}

// grid (ceil(L / 64), ceil(n_seq / 64)), 256 threads.  FILL: the whole tail of the compressor on a 64 x 64 (time x sequence) tile - level
// differences from the audio (yl = the log10 table), the smoother inside each chunk from its true start value (two chunks per tile, one
// (chunk, sequence) per thread of the first two waves), the gain application: neither x_l nor y_l ever travels to HBM.
// Energy sums for the chain fusion leave as per-tile partials (tile_sums[time tile][item][3]: sum y^2 and, for stereo with an imager
// downstream, sum (l + r)^2, sum (l - r)^2; otherwise [time tile][sequence]), reduced in a fixed order by fx_tile_sums_kernel: the same bits on
// every run (round 5 added them with float64 atomics from concurrently running launches).
template <bool FILL>
__global__ __launch_bounds__(256) void fx_comp_apply_kernel(CompArgs a, const double *yl, const double *ystart, int nchunks, int tile0, double *tile_sums) {
    __shared__ double t[64][65];
    const long tx = (long)blockIdx.x + tile0;                       // time tile (a launch covers the tiles of one time slice)
    const long n0 = tx * 64;
    const int s0 = blockIdx.y * 64;
    float xv[16];
    if constexpr (FILL) {
        static_assert(MST_COMP_T == 32, "two chunks per 64-step tile");
        // yl = the log10 table: the level differences are recomputed from the audio (audio side: lanes along time), the smoother runs
        // along time in LDS (sequence side), the result stays in the tile for the gain application below
        __shared__ double tab[256];
        __shared__ double sthr[64], smul[64];
        __shared__ int smode[64];
        __shared__ float ssf[64];
        tab[threadIdx.x] = yl[threadIdx.x];
        if (threadIdx.x < 64) {
            const int seq = s0 + threadIdx.x < a.n_seq ? s0 + threadIdx.x : a.n_seq - 1;
            const FxCompCurve c = fx_comp_curve(a, seq / a.C);
            sthr[threadIdx.x] = c.thr; smul[threadIdx.x] = c.mul; smode[threadIdx.x] = c.mode; ssf[threadIdx.x] = c.sf;
        }
        // this thread's 16 samples (scaled), all loads in flight before the first use; kept in registers for the gain application
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = k * 256 + threadIdx.x, sq = s0 + (idx >> 6);
            const long n = n0 + (idx & 63);
            const bool ok = sq < a.n_seq && n < a.L;
            const int sc = ok ? sq : 0;
            xv[k] = a.x[((size_t)(a.shared_x ? 0 : sc / a.C) * a.L + (ok ? n : 0)) * a.C + sc % a.C];      // never a predicated load
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {                               // (full unroll: xv stays in registers)
            const int idx = k * 256 + threadIdx.x, sl = idx >> 6, nl = idx & 63;
            xv[k] *= ssf[sl];
            t[nl][sl] = (s0 + sl < a.n_seq && n0 + nl < a.L) ? fx_comp_level_diff(xv[k], sthr[sl], smul[sl], smode[sl], tab) : 0.0;
        }
        __syncthreads();
        const int sl = threadIdx.x & 63, hh = threadIdx.x >> 6;
        const long k = tx * 2 + hh;
        if (hh < 2 && s0 + sl < a.n_seq && k < nchunks) {
            const double cA = 1.0 - a.alpha_att, cR = 1.0 - a.alpha_rel;
            double prev = ystart[(size_t)k * a.n_seq + s0 + sl];
#pragma unroll
            for (int i = 0; i < MST_COMP_T; ++i) {
                const double d = t[hh * MST_COMP_T + i][sl] - prev;
                prev = fma(d > 0.0 ? cA : cR, d, prev);
                t[hh * MST_COMP_T + i][sl] = prev;
            }
        }
    } else {
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int idx = k * 256 + threadIdx.x, nl = idx >> 6, sl = idx & 63;
            t[nl][sl] = (s0 + sl < a.n_seq && n0 + nl < a.L) ? yl[(size_t)(n0 + nl) * a.n_seq + s0 + sl] : 0.0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int idx = k * 256 + threadIdx.x, sl = idx >> 6, nl = idx & 63;
        const int seq = s0 + sl;
        const long n = n0 + nl;
        if (seq < a.n_seq && n < a.L) {
            const size_t e = ((size_t)(seq / a.C) * a.L + n) * a.C + seq % a.C;
            const size_t ex = a.shared_x ? (size_t)n * a.C + seq % a.C : e;
            const float xs = FILL ? xv[k] : a.x[ex] * (a.in_scale ? (float)a.in_scale[seq / a.C] : 1.0f);
            // 10^(v / 20) as exp(v ln10 / 20): the general pow() is five times the instructions for the same value to 1e-15 relative
            const float out = (float)((double)xs * exp((a.makeup - t[nl][sl]) * 0.11512925464970228420));
            a.y[e] = out;
            if (a.out_sumsq) t[nl][sl] = a.out_ms ? (double)out : (double)out * (double)out;      // this thread's own tile element: reused for the energy sums
        } else if (a.out_sumsq) {
            t[nl][sl] = 0.0;
        }
    }
    if (a.out_sumsq && a.out_ms) {
        // stereo chain, an imager downstream: the tile holds the OUTPUT samples (exact float32 values); thread (pair p = tid & 31, frames
        // nl = 8 (tid >> 5) .. + 7) forms l^2 + r^2 and the imager's mid / side terms - m = l + r, s = l - r and their squares in float32,
        // sums in float64, like fx_energy_parts_kernel - then lanes p and p + 32 meet by a shuffle, the four waves through LDS, and pair p
        // writes the tile's three partial sums of its item (C == 2: the tile's sequences 2 p, 2 p + 1 are the channels of one item)
        __shared__ double red[4][32][3];
        __syncthreads();
        const int p = threadIdx.x & 31, g8 = threadIdx.x >> 5;
        double e2 = 0.0, em = 0.0, es = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float l = (float)t[8 * g8 + i][2 * p], r = (float)t[8 * g8 + i][2 * p + 1];
            const float m = l + r, sd = l - r;
            e2 += (double)l * (double)l + (double)r * (double)r;
            em += (double)(m * m);
            es += (double)(sd * sd);
        }
        e2 += __shfl_xor(e2, 32);
        em += __shfl_xor(em, 32);
        es += __shfl_xor(es, 32);
        if ((threadIdx.x & 63) < 32) {
            red[threadIdx.x >> 6][p][0] = e2;
            red[threadIdx.x >> 6][p][1] = em;
            red[threadIdx.x >> 6][p][2] = es;
        }
        __syncthreads();
        if (threadIdx.x < 32 && s0 + 2 * p < a.n_seq) {
            double *q = tile_sums + ((size_t)tx * (a.n_seq / 2) + (s0 + 2 * p) / 2) * 3;
            q[0] = (red[0][p][0] + red[1][p][0]) + (red[2][p][0] + red[3][p][0]);
            q[1] = (red[0][p][1] + red[1][p][1]) + (red[2][p][1] + red[3][p][1]);
            q[2] = (red[0][p][2] + red[1][p][2]) + (red[2][p][2] + red[3][p][2]);
        }
    } else if (a.out_sumsq) {             // column sums of the tile: one partial per (tile, sequence); the reduction adds the channels of an item
        __syncthreads();
        if (threadIdx.x < 64 && s0 + (int)threadIdx.x < a.n_seq) {
            double cs = 0.0;
#pragma unroll 8
            for (int nl = 0; nl < 64; ++nl) cs += t[nl][threadIdx.x];
            tile_sums[(size_t)tx * a.n_seq + s0 + threadIdx.x] = cs;
        }
    }
}

// the per-tile partials of fx_comp_apply_kernel -> the MST_SUMSQ_SLOTS slots per item the chain fusion hands on: slot s of an item = its tiles
// s, s + SLOTS, s + 2 SLOTS, ... - sixteen interleaved sub-sequences of them (one thread each: at 131072 samples two loads per thread, all in
// flight at once - the kernel sits between the last apply launch and the next processor), joined in a fixed order; the channels of an item in
// channel order: the same bits on every run.  grid n_items, 16 * MST_SUMSQ_SLOTS threads.  stereo_ms: partials are [tile][item][3] and out_ms
// is written too, else [tile][sequence].
#define MST_TILE_SUBS 16
__global__ __launch_bounds__(MST_TILE_SUBS * MST_SUMSQ_SLOTS) void fx_tile_sums_kernel(const double *tile_sums, long ntiles, int n_items, int C, int stereo_ms,
                                                                                       double *out_sumsq, double *out_ms) {
    __shared__ double part[MST_TILE_SUBS][MST_SUMSQ_SLOTS][3];
    const int item = blockIdx.x, slot = threadIdx.x & (MST_SUMSQ_SLOTS - 1), sub = threadIdx.x / MST_SUMSQ_SLOTS;
    double e2 = 0.0, em = 0.0, es = 0.0;
#pragma unroll 4
    for (long tx = slot + (long)sub * MST_SUMSQ_SLOTS; tx < ntiles; tx += MST_TILE_SUBS * MST_SUMSQ_SLOTS) {
        if (stereo_ms) {
            const double *q = tile_sums + ((size_t)tx * n_items + item) * 3;
            e2 += q[0];
            em += q[1];
            es += q[2];
        } else {
            for (int c = 0; c < C; ++c) e2 += tile_sums[(size_t)tx * n_items * C + (size_t)item * C + c];
        }
    }
    part[sub][slot][0] = e2;
    part[sub][slot][1] = em;
    part[sub][slot][2] = es;
    __syncthreads();
    if (sub < 3 && (sub == 0 || stereo_ms)) {          // thread (sub = which of the three sums, slot): sixteen parts in ascending order
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < MST_TILE_SUBS; ++j) v += part[j][slot][sub];
        if (sub == 0) out_sumsq[item * MST_SUMSQ_SLOTS + slot] = v;
        else out_ms[(item * MST_SUMSQ_SLOTS + slot) * 2 + (sub - 1)] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// energy reductions (float64 accumulation; the reference sums in the input dtype, i.e. float32 pairwise
// for float32 audio - a 1e-6-relative difference documented in DESIGN.md).
//   mode 0: acc[item][0] += sum x^2                       (rms normalise, over all L*C samples)
//   mode 1: acc[item][0] += sum (l+r)^2 ; [1] += sum (l-r)^2   (mid / side energies, stereo frames)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fx_energy_kernel(const float *x, double *acc, long per_item, int mode,
                                                        int chunks) {
    __shared__ double red[2][4];
    const int item = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const long frames = (mode == 1) ? per_item / 2 : per_item;
    const long per_chunk = (frames + chunks - 1) / chunks;
    const long lo = chunk * per_chunk, hi = (lo + per_chunk < frames) ? lo + per_chunk : frames;
    const float *xp = x + (size_t)item * per_item;
    double s0 = 0.0, s1 = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        if (mode == 1) {
            const float l = xp[2 * i], r = xp[2 * i + 1];
            const float m = l + r, s = l - r;
            s0 += (double)(m * m);
            s1 += (double)(s * s);
        } else {
            const float v = xp[i];
            s0 += (double)(v * v);
        }
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s0;
        red[1][threadIdx.x >> 6] = s1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&acc[item * 2 + 0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        if (mode == 1) atomicAdd(&acc[item * 2 + 1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// the imager's two energies without atomics and without a cleared accumulator: workgroup (item, chunk) leaves its two partial sums at
// part[item][chunk][2]; the apply kernel adds the `chunks` partials of its item in order (a deterministic sum)
__global__ __launch_bounds__(256) void fx_energy_parts_kernel(const float *x, double *part, long L, int chunks) {
    __shared__ double red[2][4];
    const int item = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const long per_chunk = (L + chunks - 1) / chunks;
    const long lo = chunk * per_chunk, hi = (lo + per_chunk < L) ? lo + per_chunk : L;
    const float *xp = x + (size_t)item * L * 2;
    double s0 = 0.0, s1 = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const float l = xp[2 * i], r = xp[2 * i + 1];
        const float m = l + r, s = l - r;
        s0 += (double)(m * m);
        s1 += (double)(s * s);
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s0;
        red[1][threadIdx.x >> 6] = s1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((size_t)item * chunks + chunk) * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        part[((size_t)item * chunks + chunk) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// MidSideImager.process (:964-1007): gains from the two energies, applied in float32 like the reference
// chain fusion: in_scale = the pending rms factor of the previous processor (the energies of the raw input scale with its square, the
// samples are multiplied in float32 like the separate scale pass would); out_sumsq[item] = sum(y^2), here in closed form from the gains
// tail folding (in_sumsq non-null): the rms-normalise behind the imager and a gain behind that, in this pass: s is the factor
// fx_rms_pending_kernel computes (energy of the true input = sf^2 * sum of the raw input's slots), the store is (y * s) * post_gain
// like fx_scale_kernel's x * sf * g
__global__ __launch_bounds__(256) void fx_imager_apply_kernel(const float *x, float *y, const double *part, int chunks, long L,
                                                              double bal_rounded, const double *in_scale, double *out_sumsq,
                                                              const double *in_sumsq, float post_gain) {
    // a workgroup = MST_IMAGER_FRAMES frames of one item; its first wave adds the item's partial energies (and, when folding, the
    // slots of the input's energy) - one load per lane and the wave's butterfly sum, the order fx_rms_pending_kernel uses too
    __shared__ double sh[3];
    const int item = blockIdx.y;
    if (threadIdx.x < 64) {
        const int t = threadIdx.x;
        double e0 = t < chunks ? part[((size_t)item * chunks + t) * 2 + 0] : 0.0;
        double e1 = t < chunks ? part[((size_t)item * chunks + t) * 2 + 1] : 0.0;
        double q = in_sumsq ? in_sumsq[item * MST_SUMSQ_SLOTS + t] : 0.0;
        e0 = wave_sum(e0);
        e1 = wave_sum(e1);
        q = wave_sum(q);
        if (t == 0) {
            sh[0] = e0;
            sh[1] = e1;
            sh[2] = q;
        }
    }
    __syncthreads();
    const float sf = in_scale ? (float)in_scale[item] : 1.0f;
    const double s2 = (double)sf * (double)sf;
    const double mid_e = sh[0] * s2, side_e = sh[1] * s2;
    const double total_e = mid_e + side_e;
    const double max_side = sqrt(total_e / (side_e + 1e-3));
    const double side_gain = (bal_rounded <= 1.0) ? bal_rounded : max_side * (bal_rounded - 1.0);
    const double mid_gain = sqrt((total_e - side_e * side_gain * side_gain) / (mid_e + 1e-3));
    const float sg = (float)side_gain, mg = (float)mid_gain;
    const double sumsq_y = ((double)mg * mg * mid_e + (double)sg * sg * side_e) / 2.0;     // sum(l'^2 + r'^2), closed form
    if (out_sumsq && blockIdx.x == 0 && threadIdx.x < MST_SUMSQ_SLOTS)                     // slot 0; the others are cleared
        out_sumsq[item * MST_SUMSQ_SLOTS + threadIdx.x] = threadIdx.x == 0 ? sumsq_y : 0.0;
    float ps = 1.0f;
    if (in_sumsq) {
        const double ex = s2 * sh[2] / (double)(2 * L), ey = sumsq_y / (double)(2 * L);
        ps = (float)sqrt(ex / fmax(1e-7, ey));
    }
#pragma unroll
    for (int j = 0; j < MST_IMAGER_FRAMES / 256; ++j) {
        const long i = (long)blockIdx.x * MST_IMAGER_FRAMES + j * 256 + threadIdx.x;
        if (i < L) {
            const float *xp = x + ((size_t)item * L + i) * 2;
            float *yp = y + ((size_t)item * L + i) * 2;
            const float l = xp[0] * sf, r = xp[1] * sf;
            const float nm = (l + r) * mg, ns = (l - r) * sg;
            float o0 = (nm + ns) / 2.0f, o1 = (nm - ns) / 2.0f;
            if (in_sumsq) {
                o0 = o0 * ps * post_gain;
                o1 = o1 * ps * post_gain;
            }
            yp[0] = o0;
            yp[1] = o1;
        }
    }
}

// Gain.process (:1041-1051) and the final multiply of the rms normalise (:145-146)
// mode 0: y = g * x ; mode 1: y *= sqrt(ex / max(1e-7, ey)) with ex = acc_x/per_item, ey = acc_y/per_item
// in_scale (chain fusion, mode 0): y = (x * (float)in_scale[item]) * g - the pending rms factor first, like the separate pass would
__global__ __launch_bounds__(256) void fx_scale_kernel(const float *x, float *y, long per_item, float g,
                                                       const double *acc_x, const double *acc_y, int mode, long per_x,
                                                       const double *in_scale) {
    const int item = blockIdx.y;
    const float sf = in_scale ? (float)in_scale[item] : 1.0f;
    float scale = g;
    if (mode == 1) {        // mean(x^2) over x's own size, mean(y^2) over y's (a processor may change the channel count)
        const double ex = acc_x[item * 2] / (double)per_x, ey = acc_y[item * 2] / (double)per_item;
        scale = (float)sqrt(ex / fmax(1e-7, ey));
    }
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_item) return;
    const size_t off = (size_t)item * per_item + i;
    y[off] = (mode == 1 ? y[off] : x[off] * sf) * scale;
}

// Haas effect, haas_process (:768-786): y = x; y[:, ch] += feedback * np.roll(x[:, ch], delay).  np.roll is circular:
// roll(x, d)[i] = x[(i - d) mod L]; `shift` = delay mod L in [0, L).  float32 like numpy on a float32 array: one
// rounded multiply, one rounded add (no fused multiply-add).  A mono input [L, 1] is repeated to stereo first (:838-839).
__global__ __launch_bounds__(256) void fx_haas_kernel(const float *x, float *y, long L, int c_in, long shift, float fb, int ch) {
    const int item = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= L) return;
    const float *xp = x + (size_t)item * L * c_in;
    float *yp = y + ((size_t)item * L + i) * 2;
    float l = xp[i * c_in], r = xp[i * c_in + (c_in - 1)];
    long k = i - shift;
    if (k < 0) k += L;
    float wet = fb * xp[k * c_in + (c_in == 2 ? ch : 0)];
    MST_NO_CONTRACT(wet);              // numpy rounds the product, then the sum: keep hipcc from fusing them into an fma
    if (ch == 0) l += wet;
    else r += wet;
    yp[0] = l;
    yp[1] = r;
}

// Panner.process (:927-943): x * gains, mono repeated to stereo first; the gains come from the pan law on the host
__global__ __launch_bounds__(256) void fx_panner_kernel(const float *x, float *y, long L, int c_in, float g0, float g1) {
    const int item = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= L) return;
    const float *xp = x + ((size_t)item * L + i) * c_in;
    float *yp = y + ((size_t)item * L + i) * 2;
    yp[0] = xp[0] * g0;
    yp[1] = xp[c_in - 1] * g1;
}

// ---- FFT convolution (ConvolutionalReverb.process, common_audioeffects.py:727-764; the normaliser's 1001-tap FIR) ----------------
// The two FFTs and the inverse are csrc/fft_kernels.h (round 4; hipFFT before); everything around them is here.  A signal that is long
// against the impulse response is cut into nb overlapping blocks (overlap-save): block b holds the samples b * step - shift + i,
// i in [0, n_fft), its circular convolution with the zero-padded response is the linear convolution at the outputs b * step + j,
// j in [0, step), found at position shift + j.  A short signal is one block with step = n_fft, shift = 0.
// pack: interleaved [n_items][L][C] -> nb zero-padded real blocks of n_fft samples per (item, channel)
__global__ __launch_bounds__(256) void fx_conv_pack_kernel(const float *x, float *seq, long L, int C, long n_fft, int nb, long step, long shift) {
    const int blk = blockIdx.y, sq = blk / nb, b = blk % nb, item = sq / C, c = sq % C;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_fft) return;
    const long n = (long)b * step - shift + i;
    seq[(size_t)blk * n_fft + i] = (n >= 0 && n < L) ? x[((size_t)item * L + n) * C + c] : 0.0f;
}

// spectrum product X[block of sequence s][k] *= H[s % C][k] / n_fft (the inverse transform is unnormalised)
__global__ __launch_bounds__(256) void fx_conv_mul_kernel(float2 *X, const float2 *H, long nbin, int C, int nb, float scale) {
    const int blk = blockIdx.y, sq = blk / nb;
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= nbin) return;
    const float2 a = X[(size_t)blk * nbin + k], b = H[(size_t)(sq % C) * nbin + k];
    X[(size_t)blk * nbin + k] = make_float2((a.x * b.x - a.y * b.y) * scale, (a.x * b.y + a.y * b.x) * scale);
}

// y[item][t][c] = dry * x[item][t][c] + wet * conv[item, c][offset + t]   (the reference cuts y[idx : idx + len(x)], :754-761)
__global__ __launch_bounds__(256) void fx_conv_mix_kernel(const float *x, const float *seq, float *y, long L, int C, long n_fft, int nb,
                                                          long step, long shift, long offset, float dry, float wet) {
    const int item = blockIdx.y;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;          // element of the [L][C] item
    if (e >= L * C) return;
    const long t = e / C, m = offset + t, b = m / step;
    const int c = (int)(e % C);
    const float v = seq[(((size_t)item * C + c) * nb + b) * n_fft + shift + (m - b * step)];
    y[(size_t)item * L * C + e] = dry * x[(size_t)item * L * C + e] + wet * v;
}

// =================================================================================================
// Kernels under the input normaliser (reference mixing_manipulator/data_normalization.py, utils_data_normalization.py,
// fx_utils.py): reductions over sample ranges, the STFT front end of the EQ matching, the onset-detection function
// of the compressor matching.
// =================================================================================================

// out[r] = sum of squares (mode 0, float64) or max |x| (mode 1) over x[item[r]][lo[r] .. hi[r])[ch]; one workgroup per
// range.  Used for the BS.1770 gating-block energies (fx_utils.py:220-238 via the loudness meter) and for the peak of
// every inter-onset interval (utils_data_normalization.py:316-321: x[onset_i + argmax |x[onset_i : onset_i+1]|]).
__global__ __launch_bounds__(256) void fx_range_reduce_kernel(const float *x, long L, int C, int ch, const int *item,
                                                              const long *lo, const long *hi, int mode, double *out) {
    __shared__ double red[4];
    const int r = blockIdx.x;
    const float *xp = x + (size_t)item[r] * L * C + ch;
    const long a = lo[r] < 0 ? 0 : lo[r], b = hi[r] > L ? L : hi[r];
    double acc = 0.0;
    for (long i = a + threadIdx.x; i < b; i += 256) {
        const double v = (double)xp[i * C];
        acc = mode == 0 ? acc + v * v : fmax(acc, fabs(v));
    }
    if (mode == 0) acc = wave_sum(acc);
    else {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc = fmax(acc, __shfl_xor(acc, m));
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        out[r] = mode == 0 ? (red[0] + red[1]) + (red[2] + red[3]) : fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// STFT framing (librosa.stft(center=False) as called by common_miscellaneous.py:72): frames[f][i] = x[(f0 + f) * hop + i] * win[i]
// for one channel of one [L][C] signal; frames past the last full one are zero (they add nothing to the magnitude sum).
__global__ __launch_bounds__(256) void fx_stft_frame_kernel(const float *x, float *frames, const float *win, long L, int C, int ch,
                                                            long n_fft, long hop, long f0, long n_frames_total) {
    const long f = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_fft) return;
    const long t = (f0 + f) * hop + i;
    frames[(size_t)f * n_fft + i] = (f0 + f < n_frames_total && t < L) ? x[t * C + ch] * win[i] : 0.0f;
}

// acc[k] += sum over the batch's frames of |X[f][k]| (float32 magnitudes like the reference's complex64 STFT)
__global__ __launch_bounds__(256) void fx_stft_mag_accum_kernel(const float2 *X, float *acc, long nbin, int n_frames) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= nbin) return;
    float s = acc[k];
    for (int f = 0; f < n_frames; ++f) {
        const float2 v = X[(size_t)f * nbin + k];
        s += sqrtf(v.x * v.x + v.y * v.y);
    }
    acc[k] = s;
}

// Onset-detection function of aubio.onset('hfc', buf_size = hop_size = N) as get_mean_peak drives it
// (utils_data_normalization.py:304-314): per hop of N samples
//     phase vocoder frame = x[f*N .. (f+1)*N) * hanningz window (0.5 * (1 - cos(2 pi i / N))), halves swapped, FFT;
//     norm_k = log(|X_k| + 1)           (the 'hfc' default: logarithmic magnitude compression, lambda = 1)
//     hfc    = sum_{k=0..N/2} (k + 1) * norm_k
// plus the frame's mean square (the silence gate compares 10 log10 of it with -70 dB).  One workgroup per frame, radix-2
// FFT of N <= 2048 points in LDS, float32 like aubio.  out[f] = (hfc, mean square).
template <int N>
__global__ __launch_bounds__(256) void fx_onset_hfc_kernel(const float *x, long L, int C, int ch, long n_frames, float2 *out) {
    __shared__ float re[N], im[N];
    __shared__ float red[2][4];
    constexpr int LOGN = N == 2048 ? 11 : (N == 1024 ? 10 : (N == 512 ? 9 : 8));
    const long f = blockIdx.x % n_frames;
    const int item = (int)(blockIdx.x / n_frames);
    const float *xp = x + (size_t)item * L * C + ch + (size_t)f * N * C;
    float ms = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float v = xp[(size_t)i * C];
        ms += v * v;
        const float w = 0.5f * (1.0f - cosf(6.28318530717958647692f * (float)i / (float)N));
        const int j = (i + N / 2) & (N - 1);                  // swapped halves (zero-phase windowing)
        const int rj = (int)(__brev((unsigned)j) >> (32 - LOGN));   // bit-reversed slot for the in-place DIT transform
        re[rj] = v * w;
        im[rj] = 0.0f;
    }
    __syncthreads();
    for (int s = 1; s <= LOGN; ++s) {
        const int half = 1 << (s - 1);
        for (int t = threadIdx.x; t < N / 2; t += 256) {
            const int grp = t / half, pos = t % half;
            const int i0 = grp * 2 * half + pos, i1 = i0 + half;
            const float ang = -3.14159265358979323846f * (float)pos / (float)half;
            const float c = cosf(ang), sn = sinf(ang);
            const float tr = re[i1] * c - im[i1] * sn, ti = re[i1] * sn + im[i1] * c;
            re[i1] = re[i0] - tr;
            im[i1] = im[i0] - ti;
            re[i0] += tr;
            im[i0] += ti;
        }
        __syncthreads();
    }
    float hfc = 0.0f;
    for (int k = threadIdx.x; k <= N / 2; k += 256) hfc += (float)(k + 1) * logf(sqrtf(re[k] * re[k] + im[k] * im[k]) + 1.0f);
    hfc = wave_sum(hfc);
    ms = wave_sum(ms);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = hfc;
        red[1][threadIdx.x >> 6] = ms;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        out[(size_t)item * n_frames + f] = make_float2((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]),
                                                       ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)N);
}

// peak[item][chunk] = max |y| over the chunk-th 1/64 of the item; grid (64, n_items)
__global__ __launch_bounds__(256) void fx_item_peak_kernel(const float *y, long per_item, double *peak) {
    __shared__ float red[4];
    const int item = blockIdx.y, chunk = blockIdx.x;
    const long per_chunk = (per_item + 63) / 64;
    const long lo = chunk * per_chunk, hi = (lo + per_chunk < per_item) ? lo + per_chunk : per_item;
    const float *p = y + (size_t)item * per_item;
    float m = 0.0f;
    for (long i = lo + threadIdx.x; i < hi; i += 256) m = fmaxf(m, fabsf(p[i]));
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) peak[item * 64 + chunk] = (double)fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// hard clip of the items whose peak reached 1 (utils_data_normalization.py:352-353: `if max|y| >= 1: clip`)
__global__ __launch_bounds__(256) void fx_clip_if_kernel(float *y, long per_item, const double *peak) {
    const int item = blockIdx.y;
    double pk = 0.0;
    for (int c = 0; c < 64; ++c) pk = fmax(pk, peak[item * 64 + c]);
    if (pk < 1.0) return;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_item) return;
    float *p = y + (size_t)item * per_item + i;
    *p = fminf(1.0f, fmaxf(-1.0f, *p));
}

__global__ __launch_bounds__(256) void fx_scale_inplace_kernel(float *v, long n, float g) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] *= g;
}

// second moments of a stereo signal, float64: acc[item] = (sum L^2, sum R^2, sum L R); grid (chunks, n_items), acc zeroed by the host.
// Everything normalize_imager / process_balance need (normalization_imager.py:22-99): mid / side / left / right energies of
// any 2x2 re-mix of (L, R) follow from these three numbers.
__global__ __launch_bounds__(256) void fx_stereo_moments_kernel(const float *x, long L, double *acc) {
    __shared__ double red[3][4];
    const int item = blockIdx.y;
    const long per_chunk = (L + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per_chunk, hi = (lo + per_chunk < L) ? lo + per_chunk : L;
    const f32x2 *p = (const f32x2 *)(x + (size_t)item * L * 2);
    double s[3] = {0.0, 0.0, 0.0};
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const f32x2 v = p[i];
        s[0] += (double)v.x * (double)v.x;
        s[1] += (double)v.y * (double)v.y;
        s[2] += (double)v.x * (double)v.y;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s[k] = wave_sum(s[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(&acc[item * 3 + threadIdx.x], (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]));
}

// y = M x per stereo sample: (l', r') = (m00 l + m01 r, m10 l + m11 r)
__global__ __launch_bounds__(256) void fx_stereo_mix_kernel(const float *x, float *y, long n, float m00, float m01, float m10, float m11) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const f32x2 v = ((const f32x2 *)x)[i];
    ((f32x2 *)y)[i] = f32x2{m00 * v.x + m01 * v.y, m10 * v.x + m11 * v.y};
}

// =================================================================================================
// AlgorithmicReverb (reference common_audioeffects.py:1429-1537): per channel, feedback comb filters with a damped feedback path
// and four all-pass sections in series - the Schroeder / "Freeverb" structure of pymixconsole.components.comb / .allpass
// (pymixconsole==0.0.1 is not vendored: restated from the published structure, parity unpinned):
//   comb(D, damp, fb):  out_n = buf[n mod D];  store_n = out_n (1 - damp) + store_{n-1} damp;  buf[n mod D] = in_n + store_n fb;  y_n = out_n
//   allpass(D, fb):     out_n = buf[n mod D];  y_n = out_n - in_n;  buf[n mod D] = in_n + out_n fb
// float64 inside.  Time-parallel forms: the comb's feedback reaches back D samples, so inside a block of D samples out_n is known
// from the previous block and store_n = damp store_{n-1} + u_n is a first-order linear scan (per-lane runs + a wave scan of the
// carries); the all-pass couples only samples D apart: D independent recurrences.
// =================================================================================================
struct CombArgs {
    const float *x;      // [n_items][L][C]
    double *y;           // [n_combs][n_items * 2][L]   comb outputs per (item, side)
    long L;
    int C, n_items, n_combs;
    int delay[8][2];     // [comb][side]
    double damp, feedback, in_gain;
};

// grid (n_combs, n_items * 2), 64 threads: one wave per (comb, item, side)
__global__ __launch_bounds__(64) void fx_comb_kernel(CombArgs a) {
    constexpr int EMAX = 32;                                    // delay <= 64 * EMAX = 2048 samples
    __shared__ double store_prev[64 * EMAX];                    // store values of the previous block
    const int comb = blockIdx.x, sq = blockIdx.y, item = sq >> 1, side = sq & 1, lane = threadIdx.x;
    const int D = a.delay[comb][side];
    const int E = (D + 63) / 64;                                // consecutive elements per lane
    const int i0 = lane * E;
    const int cnt = D - i0 < 0 ? 0 : (D - i0 < E ? D - i0 : E); // this lane's share of the D samples of a block
    const float *xp = a.x + (size_t)item * a.L * a.C + (a.C == 2 ? side : 0);
    double *yp = a.y + ((size_t)comb * a.n_items * 2 + sq) * a.L;
    const double d1 = a.damp, d2 = 1.0 - a.damp;
    double dC = 1.0;                                            // damp^cnt: the decay a carry suffers across this lane's run
    for (int i = 0; i < cnt; ++i) dC *= d1;
    double carry_in = 0.0;                                      // store value just before the block
    for (long n0 = 0; n0 < a.L; n0 += D) {
        double run[EMAX], out[EMAX];
        double s = 0.0;
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            double o = 0.0;
            if (e < cnt) {
                const long n = n0 + i0 + e;
                if (n0 > 0 && n < a.L) o = (double)xp[(n - D) * a.C] * a.in_gain + store_prev[i0 + e] * a.feedback;
                s = d1 * s + d2 * o;                            // the run from a zero carry
            }
            out[e] = o;
            run[e] = s;
        }
        // wave scan of the affine maps carry -> A carry + B of the lanes' runs (inclusive), then shifted by one lane
        double A = dC, B = s;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const double Ap = __shfl_up(A, m), Bp = __shfl_up(B, m);
            if (lane >= m) { B = A * Bp + B; A = A * Ap; }
        }
        double Ax = __shfl_up(A, 1), Bx = __shfl_up(B, 1);
        if (lane == 0) { Ax = 1.0; Bx = 0.0; }
        const double c = Ax * carry_in + Bx;                    // store value entering this lane's run
        const double last = __shfl(A, 63) * carry_in + __shfl(B, 63);
        __builtin_amdgcn_wave_barrier();
        double p = d1;
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            if (e < cnt) {
                const long n = n0 + i0 + e;
                store_prev[i0 + e] = run[e] + p * c;
                if (n < a.L) yp[n] = out[e];
                p *= d1;
            }
        }
        carry_in = last;
        __builtin_amdgcn_wave_barrier();
    }
}

// one all-pass section in place over v[sq][L]: thread p of a workgroup owns the samples n = p, p + D, p + 2D, ...;
// sum_combs > 0: the input is first formed as the sum of that many comb outputs (y layout of fx_comb_kernel)
__global__ __launch_bounds__(1024) void fx_allpass_kernel(double *v, const double *combs, int sum_combs, int first_comb, long L, int n_seq,
                                                          int delay_l, int delay_r, double feedback) {
    const int sq = blockIdx.x, side = sq & 1;
    const int D = side ? delay_r : delay_l;
    double *vp = v + (size_t)sq * L;
    for (int p = threadIdx.x; p < D; p += blockDim.x) {
        double buf = 0.0;
        for (long n = p; n < L; n += D) {
            double in = vp[n];
            if (sum_combs > 0) {
                in = 0.0;
                for (int k = 0; k < sum_combs; ++k) in += combs[((size_t)(first_comb + k) * n_seq + sq) * L + n];
            }
            const double out = buf;
            vp[n] = out - in;
            buf = in + out * feedback;
        }
    }
}

// output[:, 0] = wet1 xL + wet2 xR + dry dataL ; output[:, 1] = wet1 xR + wet2 xL + dry dataR  (:1462-1463)
__global__ __launch_bounds__(256) void fx_reverb_mix_kernel(const float *x, const double *wet, float *y, long L, int C, double wet1, double wet2,
                                                            double dry) {
    const int item = blockIdx.y;
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= L) return;
    const double xl = wet[((size_t)item * 2 + 0) * L + n], xr = wet[((size_t)item * 2 + 1) * L + n];
    const float *xp = x + ((size_t)item * L + n) * C;
    const double dl = (double)xp[0], dr = (double)xp[C == 2 ? 1 : 0];
    float *yp = y + ((size_t)item * L + n) * 2;
    yp[0] = (float)(wet1 * xl + wet2 * xr + dry * dl);
    yp[1] = (float)(wet1 * xr + wet2 * xl + dry * dr);
}

// chain fusion: the pending factor of an rms-normalise step (apply_processor :143-146) from the sums the producers left behind:
// s_out = sqrt(mean(x_true^2) / max(1e-7, mean(y^2))), x_true = x_raw * s_x  =>  mean(x_true^2) = s_x^2 sumsq_x / per_x.
// Rounded to float32 like the reference's scale factor.
__global__ __launch_bounds__(64) void fx_rms_pending_kernel(const double *s_x, const double *sumsq_x, long per_x, const double *sumsq_y,
                                                           long per_y, double *s_out, int n_items) {
    const int i = blockIdx.x;          // one wave per item: a slot per lane, the wave's butterfly sum
    const float sx = s_x ? (float)s_x[i] : 1.0f;
    const double qx = wave_sum(sumsq_x[i * MST_SUMSQ_SLOTS + threadIdx.x]), qy = wave_sum(sumsq_y[i * MST_SUMSQ_SLOTS + threadIdx.x]);
    const double ex = (double)sx * (double)sx * qx / (double)per_x, ey = qy / (double)per_y;
    if (threadIdx.x == 0) s_out[i] = (double)(float)sqrt(ex / fmax(1e-7, ey));
}

// out[item][chunk] = sum of x^2 over the chunk-th part of the item (float64), slots chunks .. 63 cleared; grid n_items * chunks, chunks <= 64
__global__ __launch_bounds__(256) void fx_sumsq_kernel(const float *x, double *out, long per_item, int chunks) {
    __shared__ double red[4];
    const int item = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const long per_chunk = (per_item + chunks - 1) / chunks;
    const long lo = chunk * per_chunk, hi = (lo + per_chunk < per_item) ? lo + per_chunk : per_item;
    const float *xp = x + (size_t)item * per_item;
    double s = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const float v = xp[i];
        s += (double)(v * v);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[item * MST_SUMSQ_SLOTS + chunk] = (red[0] + red[1]) + (red[2] + red[3]);
    if (chunk == 0 && threadIdx.x >= chunks && threadIdx.x < MST_SUMSQ_SLOTS) out[item * MST_SUMSQ_SLOTS + threadIdx.x] = 0.0;
}
