// gfx950 kernels of the MixFXcloner TCN (reference networks/architectures.py:177-234 TCNBlock.forward,
// :135-147 TCNModel.forward, networks/network_utils.py:180-182 FiLM).
//
// Data layout in HBM: activations between blocks are TIME-MAJOR / CHANNEL-MINOR ("NLC"):
//     act[b][t][c]   c = 0..127 contiguous (256 B per time step in bf16, 512 B in fp32), row pitch Lp.
// so that (a) one time step is one contiguous row: any set of time steps - in particular the dilated
// taps t + (j-7)*d - is gathered at full cache-line efficiency whatever d is, and (b) the 8 consecutive
// input channels a bf16 MFMA lane needs are one 16-byte LDS read.
//
// Tiling ("polyphase rows"): a dilated conv with dilation d is d independent dense convs over the
// phases t = m*d + phi.  A workgroup owns P consecutive phases x Mt consecutive steps (P*Mt = 256 output
// time steps, P | d) and all 128 output channels.  Its input rows are staged ONCE into LDS as a flat list
//     row r  <->  step (m0 + r/P - 7), phase (phi0 + r%P)         r in [0, 256 + 14*P)
// so output o = (m - m0)*P + (phi - phi0) reads row o + j*P for tap j: inside LDS the dilated conv is a
// conv with stride-P taps, and the input is read from HBM (256+14P)/256 = 1.1-1.9x instead of 15x.
// The main loop then has NO barrier: B operands (activations) come from LDS, A operands (BN-folded
// weights, pre-packed in fragment order) stream from L2 straight into registers.
//
// MFMA orientation: D[co][t] = sum_k W'[co][k] * X[k][t]; wave w owns output channels 32w..32w+31 and all
// 256 time steps of the tile (bf16 / bf16x3: 2 x 16 accumulator tiles of 16x16; fp32: 8 tiles of 32x32).  A lane's 4 consecutive
// accumulator rows are 4 consecutive channels -> one 8-byte (bf16) / 16-byte (fp32) store per time step.
//
// Epilogue (fused): + BN shift -> LeakyReLU(0.01) -> FiLM r*y+b -> + res_scale * x_in  -> store.
#pragma once
#include <type_traits>

#include "mst_dev.h"

struct TcnBlockArgs {
    const void *x;        // act in  [B][Lp][128]
    void *y;              // act out [B][Lp][128]
    const void *wpk;      // packed BN-folded conv weights (fragment order, see pack_* in mst_api)
    const float *shift;   // [128] BN shift  beta - mean*gamma/sqrt(var+eps)
    const float *film;    // [film_rows][256]  r = [0,128), b = [128,256)
    const float *res;     // [128] grouped 1x1 residual scale
    int film_rows;        // 1 (broadcast) or B
    int B, L, Lp, d;
    int tiles_phase;      // d / P
    int tiles_step;       // ceil(ceil(L/d) / (256/P))
    const float *out_w;   // fused output head (last block only): 1x1 conv [nout][128], bias [nout], y [B][nout][L] fp32
    const float *out_b;
    float *y_out;
    int nout;
    const void *zeros;    // >= 256 bytes of zeros in device memory: the row staged for time steps outside the segment
    int xcd_tiles;        // > 0: tiles per XCD; workgroup i (dispatched to XCD i % 8) takes tile (i % 8) * xcd_tiles + i / 8, so that
                          // neighbouring time tiles (which share their halo rows) run on the same XCD and meet in its L2
    // block 0 fused into this launch (duo kernel with FUSE0, d = P = 2: a tile's rows are consecutive samples): the loader waves compute
    // the tile's input rows from the waveform with block 0's weights instead of fetching them (x is not read)
    const float *x0 = nullptr;         // waveform [B][2][L]
    const void *w0pk = nullptr;        // block 0's A fragments (TcnBlock0Args::wpk16)
    const float *shift0 = nullptr, *film0 = nullptr, *res0 = nullptr;      // block 0's BN shift, FiLM rows (film_rows of them), residual scale
};

// ------------------------------------------------------------------------------------------------
// Block 0's arithmetic in the bf16 mode, shared by tcn_block0_mfma_kernel and the duo kernel's FUSE0 loader so that both produce the same
// bits: the B fragments of one k-step (16 of the 32 k = ci * 15 + j; k >= 30 is padding) for the 32 output times o0 .. o0 + 31 of a
// waveform window xs[ci][k] = x[ci][t_first - 7 + k], split hi + lo; and one output element.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tcn_block0_bfrag(const float *xs0, const float *xs1, int o, int sI, int h, bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 16 * sI + 8 * h + e;               // ci = k / 15, tap j = k % 15
        const int ci = k >= 15 ? 1 : 0, j = k - 15 * ci;
        const float v = k < 30 ? (ci ? xs1 : xs0)[o + j] : 0.0f;
        const __bf16 vh = (__bf16)v;
        hi[e] = vh;
        lo[e] = (__bf16)(v - (float)vh);
    }
}
__device__ __forceinline__ __bf16 tcn_block0_out(float v, float fr, float fb, float rs, float xres) {
    // explicit fused multiply-adds: the two call sites must not be contracted differently (measured: they were - 1 ulp of bf16 apart)
    return (__bf16)fmaf(fr, mst_fmax(v, MST_LEAKY * v), fmaf(rs, xres, fb));
}

// ------------------------------------------------------------------------------------------------
// The class-major main loop (B fragments reused across taps, tcn_reuse_class below) for the one-tile kernel's 128-time tiles that span their
// WHOLE phase sequence: four phases x 32 steps (d = 4096 at L = 131072) and eight phases x 16 steps (d = 8192, the last block).
// Fully unrolled - every index is a compile-time constant - so that the (column tile, tap) pairs that only see zero padding are simply not
// there (P = 4: 8 of 120, P = 8: 24 of 120).  Taps of one residue class mod S = 16 / P share their B fragments (tap j + S, column tile
// q - 1 = the rows of tap j, column tile q); a class is walked in groups of <= 4 taps ("pseudo-classes": 4 + 4 + 4 + 3 taps at both P)
// so that a group's A fragments are 32 registers, double-buffered over the 16 phases (pseudo-class x k-step).
// ------------------------------------------------------------------------------------------------
// H = halo steps kept in the LDS image on either side: 7 (the general image) for the 128-time tiles.  Round 6: 256-time tiles that span their whole
// phase sequence - sixteen phases x 16 steps (the last block, d = 8192 at L = 131072), eight x 32 (d = 4096), four x 64 (d = 2048) - keep only the
// 16 / P - 1 halo steps a row window (16 rows = 16 / P steps) can straddle into: windows that are all zero padding are neither staged nor read, the
// image is 256 / 272 / 280 rows instead of 480 / 368 / 312, and the windows of a pseudo-class are walked over their live range only.
template <int P, int NC, int H>
struct tcn_whole_tab {
    static constexpr int S = 16 / P, MT = 16 * NC / P;
    static constexpr int j0_of(int pc) { return pc % S + 4 * S * (pc / S); }          // first tap of pseudo-class pc
    static constexpr int nu_of(int pc) { return pc < 3 ? 4 : 3; }
    static constexpr int nw_of(int pc) { return NC - 1 + nu_of(pc); }
    // live window range of a pseudo-class: everything with the full halo; else the windows that reach into the tile's own steps (window i of the
    // pseudo-class with first tap j0 covers steps j0 + S i - 7 ... + S - 1)
    static constexpr int cdiv(int a, int b) { return a <= 0 ? 0 : (a + b - 1) / b; }
    static constexpr int lo_of(int pc) { return H == 7 ? 0 : cdiv(8 - S - j0_of(pc), S); }
    static constexpr int hi_of(int pc) { return H == 7 ? nw_of(pc) : (cdiv(MT + 7 - j0_of(pc), S) < nw_of(pc) ? cdiv(MT + 7 - j0_of(pc), S) : nw_of(pc)); }
    static constexpr int cnt_of(int pc) { return hi_of(pc) - lo_of(pc); }
    // the window sequence as TABLES (built at compile time): with the loop counters of the unrolled main loop as indices the optimiser folds a table
    // lookup reliably; a search loop per window it did not always fold (a first build of the eight- and four-phase forms ran 25 x slower: 35 ms)
    struct Seq {
        int n0[17];                         // windows in front of phase ph = (pseudo-class ph >> 2, k-step ph & 3); n0[16] = all of them
        int ph[16 * (NC + 3)], i[16 * (NC + 3)];
    };
    static constexpr Seq make() {
        Seq q{};
        int n = 0;
        for (int ph = 0; ph < 16; ++ph) {
            q.n0[ph] = n;
            for (int i = lo_of(ph >> 2); i < hi_of(ph >> 2); ++i) {
                q.ph[n] = ph;
                q.i[n] = i;
                ++n;
            }
        }
        q.n0[16] = n;
        return q;
    }
    static constexpr Seq seq = make();
    static constexpr int NWIN = seq.n0[16];
};
template <int P, int NC, int H = 7>
__device__ __forceinline__ void tcn_class_major_whole_tile(f32x4 (&acc)[2][NC], const unsigned char *smem, const MstStream16 &wst, unsigned aoff,
                                                           int l16, int g) {
    static_assert(((P == 4 || P == 8) && NC == 8 && H == 7) || ((P == 16 || P == 8 || P == 4) && NC == 16 && H == 16 / P - 1),
                  "128-time tiles of four / eight phases with the full halo; 256-time tiles of four / eight / sixteen phases with the straddled halo steps only");
    using TB = tcn_whole_tab<P, NC, H>;
    constexpr int S = TB::S, MT = TB::MT, NWIN = TB::NWIN;
    auto dead = [](int q, int j) {          // rows of steps < 0 / >= MT are padding (the tile starts at the first step and ends at the last)
        const int s_lo = (16 * q) / P + j - 7, s_hi = (16 * q + 15) / P + j - 7;
        return s_hi < 0 || s_lo >= MT;
    };
    auto window = [&](int n) -> const unsigned char * {                      // LDS address of this lane's 16 bytes of window n
        const int ph = TB::seq.ph[n], i = TB::seq.i[n];
        const int row0 = P * TB::j0_of(ph >> 2) - (7 - H) * P, kk = ph & 3;          // ((7 - H) P is a multiple of 16: the swizzle term is that of the full image)
        return smem + (row0 + 16 * i + l16) * 256 + (((4 * kk + g) ^ ((row0 + l16) & 15)) << 4);
    };
    auto load_a = [&](bf16x8 (&A)[4][2], int ph) {
        const int pc = ph >> 2, kk = ph & 3;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u >= TB::nu_of(pc)) continue;
            const unsigned so = (unsigned)((TB::j0_of(pc) + S * u) * 4 + kk) * 8192u;
            A[u][0] = __builtin_bit_cast(bf16x8, mst_stream_load16(wst, aoff, so));
            A[u][1] = __builtin_bit_cast(bf16x8, mst_stream_load16(wst, aoff + 4096u, so));
        }
    };
    bf16x8 A0[4][2], A1[4][2], ring[4];
    load_a(A0, 0);
#pragma unroll
    for (int n = 0; n < 4; ++n) ring[n] = *(const bf16x8 *)window(n);
#pragma unroll
    for (int ph = 0; ph < 16; ++ph) {
        bf16x8 (&cur)[4][2] = (ph & 1) ? A1 : A0;
        bf16x8 (&nxt)[4][2] = (ph & 1) ? A0 : A1;
        if (ph < 15) load_a(nxt, ph + 1);
        const int pc = ph >> 2, n0 = TB::seq.n0[ph], lo = TB::lo_of(pc), hi = TB::hi_of(pc);
#pragma unroll
        for (int i = lo; i < hi; ++i) {
            const int n = n0 + i - lo;
            const bf16x8 b = ring[n & 3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = i - u, j = TB::j0_of(pc) + S * u;
                if (u < TB::nu_of(pc) && q >= 0 && q < NC && !dead(q, j)) {
                    acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[u][0], b, acc[0][q], 0, 0, 0);
                    acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[u][1], b, acc[1][q], 0, 0, 0);
                }
            }
            if (n + 4 < NWIN) ring[n & 3] = *(const bf16x8 *)window(n + 4);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// blocks 1..n-1, bf16 MFMA (v_mfma_f32_16x16x32_bf16), bf16 activations.  MFMA-bound:
// 2*128*1920 = 491 520 FLOP per output time step against 512 B of HBM traffic.
// ------------------------------------------------------------------------------------------------
// NQ = groups of 32 output times per wave: 8 -> 256-time tiles (2 workgroups per CU at P <= 4), 4 -> 128-time tiles
// (half the accumulators and LDS: 3 workgroups per CU, A fragments re-streamed twice as often).
// The matrix instruction is v_mfma_f32_16x16x32_bf16 (round 2): the wave's 32 channels are two row tiles of 16, its times 2 NQ column
// tiles of 16, a k-step is 32 input channels of one tap.  Per FLOP it moves exactly the operands the 32 x 32 x 16 form moved (every
// B fragment feeds two MFMAs) - and under the chip's power limit it runs 15 % faster: on realistic operands the bare instruction
// stream sustains 1934-1982 TFLOP/s against 1666-1685, the whole main loop 1570-1585 against 1367-1377
// (tools/micro/tcn_mainloop_variants.hip, profiles/archive/r02_micro_tcn_mainloop_variants_16x16.txt).
// The tap-major loop below serves the tiles that do not span their phase sequence; whole-sequence 128-time tiles run
// tcn_class_major_whole_tile, the duo kernel's 256-time tiles tcn_reuse_class (B fragments reused across taps).
// TCN_LIVE_MIN_P: phases per tile from which all-padding (column tile, tap) pairs are skipped under a branch.  8 was measured and dropped for this kernel (round 4, same-box
// A/B at 32 x 131072): the d = 4096 / 8192 blocks skip 10 % / 20 % of their MFMAs but run 1.58 -> 1.85 / 1.57 -> 1.74 ms - the wave-uniform
// branches around the MFMA pairs break the (mfma, mfma, ds_read) software pipeline; the split-bf16 kernel (6 MFMAs per branch) gains 3-5 %
constexpr int TCN_LIVE_MIN_P = 16;
// WHOLE (round 6): every tile of the launch spans its whole phase sequence (tiles_step == 1 and ceil(L / d) == MT: the host checks) - the
// kernel holds ONLY the unrolled class-major loop.  With both loops in one kernel (chosen per workgroup) the register allocation was the
// maximum over the two and <4, false, 4> spilled 10 VGPRs at three workgroups per CU (44 bytes of scratch per lane); split, neither form spills.
// WHOLE: 0 tap-major loop; 1 every tile spans its whole phase sequence (128-time tiles, dead pairs left out);
// 2 (round 6, mst_tcn_set_tuning bit 7): 256-time tiles of two / four phases with the duo kernel's class-major loop (tcn_reuse_class: the duo
// kernel's products in the duo kernel's order - bit-identical to it), two workgroups per CU = two matrix waves per SIMD whose staging and
// epilogue hide behind each other's main loops.  Same-box alternating A/B at 32 x 131072, d = 4 ... 2048 (profiles/r06_tcn_forms_onetile_ab.txt):
// duo kernel 1.404-1.409 ms per launch, 128-time class-major tiles at three workgroups per CU 1.336-1.338, this form 1.312-1.318.  The
// three-workgroup form keeps the matrix pipe busy 0.94 of the kernel's cycles (profiles/r06_pmc_sq_tcn_block_bf16_cm128.txt) but pulls the
// shader clock to ~1.55-1.7 GHz under the chip's power limit: what a tile costs in ENERGY decides, and 256-time tiles stream every weight
// fragment from L2 half as often (8 GB per launch instead of 16) and stage 1.22 instead of 1.44 rows per output row.
template <int P, int NU, bool LAST, int NUMAX, bool WRAP = true>
__device__ __forceinline__ void tcn_reuse_class(f32x4 (&acc)[2][16], bf16x8 (&A0)[NUMAX][2], bf16x8 (&A1)[NUMAX][2], bf16x8 (&ring)[4],
                                                const unsigned char *sm, const MstStream16 &wst, unsigned aoff, int c, int cn, int l16, int g);
// FUSE0 (round 6; two-phase class-major tiles, d = 2: a tile's rows are consecutive samples): block 0 is not launched - the workgroup computes
// its tile's input rows from the waveform with tcn_block0_mfma_kernel's arithmetic (the duo kernel's FUSE0 loader, same fragments, same MFMA
// order, same epilogue function: the same bits) instead of fetching them; the other workgroup of the CU runs its main loop meanwhile.
template <int P, bool FUSE_OUT, int NQ, int WHOLE = 0, bool FUSE0 = false>
__global__ __launch_bounds__(256, (NQ == 4 ? (P <= 4 ? 3 : 2) : (P <= 4 || WHOLE == 1 ? 2 : 1))) void tcn_block_bf16_kernel(TcnBlockArgs a) {
    static_assert(WHOLE != 1 || ((P == 8 || P == 4) && NQ == 4) || ((P == 16 || P == 8 || P == 4) && NQ == 8),
                  "whole-sequence tiles: 128 times of four / eight phases, 256 times of four / eight / sixteen");
    static_assert(WHOLE != 2 || ((P == 4 || P == 2) && NQ == 8), "class-major 256-time tiles of two / four phases");
    static_assert(!FUSE0 || (P == 2 && NQ == 8 && WHOLE == 2 && !FUSE_OUT), "block 0 is fused into the d = 2 block's two-phase class-major tiles");
    constexpr int H = (WHOLE == 1 && NQ == 8) ? 16 / P - 1 : 7;    // halo steps in the LDS image (256-time whole-sequence tiles: only what a row window straddles, tcn_class_major_whole_tile)
    constexpr int T = 32 * NQ, R = T + 2 * H * P, MT = T / P, NC = 2 * NQ;
    __shared__ __attribute__((aligned(16))) unsigned char smem[R * 256];
    __shared__ __attribute__((aligned(16))) float par[4 * 128];     // shift | FiLM r | FiLM b | res of this block / batch item
    constexpr int XWP = 304;                                         // FUSE0: waveform samples per channel a tile needs (R + 14 = 298), padded
    __shared__ float xs0[FUSE0 ? 2 * XWP : 1];                       // FUSE0: the tile's waveform window
    __shared__ __attribute__((aligned(16))) float par0[FUSE0 ? 4 * 128 : 4];          // FUSE0: block 0's shift | FiLM r | FiLM b | res
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l16 = lane & 15, g = lane >> 4;

    int tile = a.xcd_tiles > 0 ? (int)(blockIdx.x & 7) * a.xcd_tiles + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int mg = tile % a.tiles_step;
    tile /= a.tiles_step;
    const int pg = tile % a.tiles_phase;
    const int b = tile / a.tiles_phase;
    const int m0 = mg * MT, phi0 = pg * P;
    const __bf16 *xb = (const __bf16 *)a.x + (size_t)b * a.Lp * 128;
    __bf16 *yb = (__bf16 *)a.y + (size_t)b * a.Lp * 128;
    // ---- stage the (256 + 14P) input rows: 16 lanes x 16 B per row, XOR-swizzled 16-B slots so that the
    //      32 consecutive rows of one B-fragment read hit 16 distinct slots per ds_read_b128 lane group
    if constexpr (FUSE0) {
        // the rows of tile (b, m0) are block 0's outputs at the consecutive times t_first + r; rows outside the segment are zero rows (this
        // block's padding).  Wave w computes the 32-row groups w and w + 4 (all four channel quarters) and its quarter of the halo group 8.
        const int ln = lane & 31, h = lane >> 5;
        float *xw0 = xs0, *xw1 = xs0 + XWP;
        const long t_first = (long)(m0 - 7) * a.d + phi0;           // time of row 0
        for (int i = tid; i < 2 * XWP; i += 256) {
            const int ci = i >= XWP ? 1 : 0, k = i - ci * XWP;
            const long t = t_first - 7 + k;
            (ci ? xw1 : xw0)[k] = (t >= 0 && t < a.L) ? a.x0[((size_t)b * 2 + ci) * a.L + t] : 0.0f;
        }
        if (tid < 128) {
            const float *frow0 = a.film + (a.film_rows > 1 ? (size_t)b * 256 : 0);
            const float *frow00 = a.film0 + (a.film_rows > 1 ? (size_t)b * 256 : 0);
            par[tid] = a.shift[tid];
            par[128 + tid] = frow0[tid];
            par[256 + tid] = frow0[128 + tid];
            par[384 + tid] = a.res[tid];
            par0[tid] = a.shift0[tid];
            par0[128 + tid] = frow00[tid];
            par0[256 + tid] = frow00[128 + tid];
            par0[384 + tid] = a.res0[tid];
        }
        __syncthreads();
        const bf16x8 *const w0p = (const bf16x8 *)a.w0pk + lane;
        static_assert(!FUSE0 || (R + 31) / 32 == 9, "eight row groups + one halo group");
#pragma unroll 1
        for (int q = w; q < 12; q += 4) {
            const bool halo = q >= 8;                                  // third trip: group 8, one quarter
            if (halo) q = 8;
            bf16x8 bh[2], bl[2];
#pragma unroll
            for (int sI = 0; sI < 2; ++sI) tcn_block0_bfrag(xw0, xw1, 32 * q + ln, sI, h, bh[sI], bl[sI]);
            const int o = 32 * q + ln;
            const long t = t_first + o;
            const bool inside = t >= 0 && t < a.L;
#pragma unroll 1
            for (int cw = halo ? w : 0; cw < (halo ? w + 1 : 4); ++cw) {      // channel quarters (what the four waves of the block-0 kernel do)
                f32x16 acc0;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 sh = *(const f32x4 *)(par0 + 32 * cw + 8 * gq + 4 * h);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc0[4 * gq + i] = sh[i];
                }
#pragma unroll
                for (int sI = 0; sI < 2; ++sI) {
                    const bf16x8 af = w0p[(sI ? 256 : 0) + 64 * cw];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bh[sI], acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bl[sI], acc0, 0, 0, 0);
                }
                const float xres = ((cw >> 1) ? xw1 : xw0)[o + 7];      // grouped residual: channels 0..63 read input 0, 64..127 input 1
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int co0 = 32 * cw + 8 * gq + 4 * h;
                    const f32x4 fr = *(const f32x4 *)(par0 + 128 + co0);
                    const f32x4 fb = *(const f32x4 *)(par0 + 256 + co0);
                    const f32x4 rs = *(const f32x4 *)(par0 + 384 + co0);
                    bf16x4 out;
#pragma unroll
                    for (int i = 0; i < 4; ++i) out[i] = inside ? tcn_block0_out(acc0[4 * gq + i], fr[i], fb[i], rs[i], xres) : (__bf16)0.0f;
                    if (o < R) *(bf16x4 *)(smem + o * 256 + (((co0 >> 3) ^ (o & 15)) << 4) + 8 * h) = out;
                }
            }
        }
    } else {
        // all (R+15)/16 row loads of a thread are issued back to back (one exposed memory latency per tile;
        // the accumulators are not live yet, so the registers are free), then written to LDS
        // consecutive passes of a thread are 16 / P dilation steps apart in time and 4096 bytes apart in LDS (the XOR
        // swizzle depends on row & 15 = the thread's first row only): running pointers, no per-row address arithmetic.
        // NEVER a predicated load: hipcc branches around each and, here, drained vmcnt(0) behind the second one (a register-tuple
        // copy) - a second exposed latency per tile.  Rows outside the segment read a row of zeros instead.
        static_assert(16 % P == 0, "row passes advance by a whole number of steps");
        const int slot = tid & 15, prow = tid >> 4;
        constexpr int NPASS = (R + 15) / 16;
        const long dt = (long)(16 / P) * a.d;
        long t = (long)(m0 + prow / P - H) * a.d + phi0 + (prow % P);
        const __bf16 *src = xb + t * 128 + slot * 8;
        const __bf16 *zsrc = (const __bf16 *)a.zeros + slot * 8;
        bf16x8 v[NPASS];
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const bool ok = prow + 16 * i < R && t >= 0 && t < a.L;
            v[i] = *(const bf16x8 *)(ok ? src : zsrc);
            t += dt;
            src += dt * 128;
        }
        // per-channel epilogue parameters -> LDS (one broadcast ds_read_b128 each in the epilogue instead of four exposed L2 round
        // trips); requested BEHIND the row loads, so that their latency is not a third one in front of the tile's
        float pv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        {
            const int pc = tid & 127;
            const float *frow0 = a.film + (a.film_rows > 1 ? (size_t)b * 256 : 0);
            pv[0] = a.shift[pc];
            pv[1] = frow0[pc];
            pv[2] = frow0[128 + pc];
            pv[3] = a.res[pc];
        }
        unsigned char *dst = smem + prow * 256 + ((slot ^ (prow & 15)) << 4);
#pragma unroll
        for (int i = 0; i < NPASS; ++i)
            if (prow + 16 * i < R) *(bf16x8 *)(dst + i * 4096) = v[i];
        if (tid < 128) {
            par[tid] = pv[0];
            par[128 + tid] = pv[1];
            par[256 + tid] = pv[2];
            par[384 + tid] = pv[3];
        }
    }
    __syncthreads();

    // the accumulators start from the BN shift of their channel (row 4 g + i of the row tile's 16): no add later
    f32x4 acc[2][NC];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const f32x4 sh = *(const f32x4 *)(par + 32 * w + 16 * m + 4 * g);
#pragma unroll
        for (int q = 0; q < NC; ++q) acc[m][q] = sh;
    }

    // A fragments: wpk[ks = j*4 + kk][row tile m][wave][lane] = 8 bf16 = W'[32w + 16m + (lane & 15)][32kk + 8 (lane >> 4) + e][j]
    {
        // software pipeline: a ring of 8 B fragments - the fragment eight column tiles ahead (of this k-step, or of the next one) is
        // requested from LDS right behind the two MFMAs that free its register: 16 MFMAs = 256 clocks of latency cover; the A fragments
        // of (j+1, kk) are requested from L2 as soon as (j, kk) has been consumed.
        // A fragment addresses = uniform (scalar) base of the k-step + a fixed 32-bit lane offset: no per-load vector address math
        // (buffer loads: the k-step part of the address is an SGPR offset - no 64-bit vector address per load; the loads of a k-step are
        //  pinned behind its last MFMA group - left to itself hipcc sinks all eight loads of a tap behind the tap's last MFMA, so that the first
        //  k-step of the next tap waits for an L2 round trip)
        const MstStream16 wst = mst_stream16(a.wpk, 60u * 2u * 4096u);
        const unsigned aoff = (unsigned)(w * 64 + lane) * 16u;
        constexpr int RB = 8;
        static_assert(NC % RB == 0, "the ring divides the column tiles");
        // A (column tile q, tap j) pair whose 16 input rows are all zero padding contributes nothing.
        //  * P = 16 tiles (largest dilation on a very short segment): skipped under a wave-uniform branch.
        //  * 128-time tiles of four / eight phases that span their WHOLE phase sequence (d = 4096 / 8192 at L = 131072, the last two
        //    blocks): the pattern is known at COMPILE time - tcn_class_major_whole_tile is unrolled and the dead pairs are simply not there
        //    (7 % / 20 % of the tile's MFMAs).  A branch per MFMA pair instead was measured slower than not skipping at all (it breaks
        //    the mfma / mfma / ds_read pipeline: 1.58 -> 1.85 ms, TCN_LIVE_MIN_P).
        const int nsteps = (int)(((long)a.L + a.d - 1) / a.d);
        auto tap_major = [&]() {
            bf16x8 af[2][4], bf[RB];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int m = 0; m < 2; ++m) af[m][kk] = __builtin_bit_cast(bf16x8, mst_stream_load16(wst, aoff + m * 4096, (unsigned)kk * 8192u));
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                const unsigned char *rp0 = smem + l16 * 256 + ((g ^ l16) << 4);
#pragma unroll
                for (int q = 0; q < RB; ++q) bf[q] = *(const bf16x8 *)(rp0 + q * 4096);
            }
            for (int j = 0; j < 15; ++j) {
                const int jn = j < 14 ? j + 1 : 14;
                const int rb0 = j * P + l16, rb1 = jn * P + l16;
                unsigned live = 0xffffu;
                if constexpr (P >= TCN_LIVE_MIN_P) {
                    live = 0;
#pragma unroll
                    for (int q = 0; q < NC; ++q) {
                        const int s_lo = m0 + (16 * q) / P + j - 7, s_hi = m0 + (16 * q + 15) / P + j - 7;
                        if (!(s_hi < 0 || s_lo >= nsteps)) live |= 1u << q;
                    }
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int rbn = (kk == 3) ? rb1 : rb0;
                    const int kn = (kk + 1) & 3;
                    const unsigned char *cp = smem + rb0 * 256 + (((4 * kk + g) ^ (rb0 & 15)) << 4);
                    const unsigned char *np = smem + rbn * 256 + (((4 * kn + g) ^ (rbn & 15)) << 4);
#pragma unroll
                    for (int q = 0; q < NC; ++q) {
                        if (P < TCN_LIVE_MIN_P || ((live >> q) & 1u)) {
                            acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][kk], bf[q % RB], acc[0][q], 0, 0, 0);
                            acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][kk], bf[q % RB], acc[1][q], 0, 0, 0);
                        }
                        bf[q % RB] = (q + RB < NC) ? *(const bf16x8 *)(cp + (q + RB) * 4096) : *(const bf16x8 *)(np + (q + RB - NC) * 4096);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
#pragma unroll
                    for (int m = 0; m < 2; ++m) af[m][kk] = __builtin_bit_cast(bf16x8, mst_stream_load16(wst, aoff + m * 4096, (unsigned)(jn * 4 + kk) * 8192u));
                    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                }
            }
        };
        // only a tile that spans its WHOLE phase sequence takes the unrolled class-major form (the one-sided forms for a sequence of two
        // tiles - 10 % fewer MFMAs each - were measured slower in the tap-major order: two unrolled 12 KB loops alternating on a CU)
        if constexpr (WHOLE == 1) tcn_class_major_whole_tile<P, NC, H>(acc, smem, wst, aoff, l16, g);
        else if constexpr (WHOLE == 2) {
            constexpr int NCLS = 16 / P, NUMAX = (15 + NCLS - 1) / NCLS;
            bf16x8 A0[NUMAX][2], A1[NUMAX][2], ring[4];
#pragma unroll
            for (int u = 0; u < NUMAX; ++u) {
#pragma unroll
                for (int m = 0; m < 2; ++m) A0[u][m] = __builtin_bit_cast(bf16x8, mst_stream_load16(wst, aoff + m * 4096, (unsigned)(4 * NCLS * u) * 8192u));
            }
            {
                const unsigned char *rp0 = smem + l16 * 256 + ((g ^ l16) << 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) ring[i] = *(const bf16x8 *)(rp0 + i * 4096);
            }
#pragma unroll 1
            for (int c = 0; c < NCLS - 1; ++c) tcn_reuse_class<P, NUMAX, false, NUMAX>(acc, A0, A1, ring, smem, wst, aoff, c, c + 1, l16, g);
            tcn_reuse_class<P, 15 / NCLS, true, NUMAX, false>(acc, A0, A1, ring, smem, wst, aoff, NCLS - 1, 0, l16, g);
        } else tap_major();
    }

    // ---- fused epilogue
    // residual inputs (centre tap rows) -> registers, then the input tile is dead and LDS is reused to transpose
    // the output tile so that global stores are whole 256-byte rows, 16 B per lane.  A lane's four accumulator rows of a tile are four
    // consecutive channels co0 .. co0 + 3 of output time 16 q + l16.
    // The row pass behind the epilogue starts from an OPAQUE copy of the thread index: its lane coordinates (slot, row, time, pointers) equal
    // the staging pass's, and hipcc otherwise keeps them alive across the main loop - at three workgroups per CU that was the spill.
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    bf16x4 xin[2][NC];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int co0 = 32 * w + 16 * m + 4 * g;
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int row = 16 * q + l16 + H * P;
            xin[m][q] = *(const bf16x4 *)(smem + row * 256 + (((co0 >> 3) ^ (row & 15)) << 4) + 2 * (co0 & 7));
        }
    }
    __syncthreads();
    float hs0[NC], hs1[NC];            // FUSE_OUT: this lane's partial sums of the 1x1 output head, per column tile
#pragma unroll
    for (int q = 0; q < NC; ++q) hs0[q] = hs1[q] = 0.0f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int co0 = 32 * w + 16 * m + 4 * g;
        const f32x4 fr = *(const f32x4 *)(par + 128 + co0);
        const f32x4 fb = *(const f32x4 *)(par + 256 + co0);
        const f32x4 rs = *(const f32x4 *)(par + 384 + co0);
        f32x4 ow0 = {0.0f, 0.0f, 0.0f, 0.0f}, ow1 = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (FUSE_OUT) {
            ow0 = *(const f32x4 *)(a.out_w + co0);
            if (a.nout > 1) ow1 = *(const f32x4 *)(a.out_w + 128 + co0);
        }
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int o = 16 * q + l16;
            const float v4[4] = {acc[m][q][0], acc[m][q][1], acc[m][q][2], acc[m][q][3]};
            const bf16x4 out = tcn_epilogue4(v4, fr, fb, rs, xin[m][q]);
            if constexpr (FUSE_OUT) {          // the head reads the bf16-rounded activation, like the separate output kernel
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hs0[q] = fmaf(ow0[i], (float)out[i], hs0[q]);
                    hs1[q] = fmaf(ow1[i], (float)out[i], hs1[q]);
                }
            }
            if constexpr (!FUSE_OUT) *(bf16x4 *)(smem + o * 256 + (((co0 >> 3) ^ (o & 15)) << 4) + 2 * (co0 & 7)) = out;
        }
    }
    if constexpr (FUSE_OUT) {
        // last block: 1x1 output conv + bias + clamp(-1, 1) (reference architectures.py:133,145) straight from the
        // registers - the last activation never travels to HBM.  The four lanes l16 + 16 g hold the four channel groups of a
        // column tile, the four waves the four channel quarters: two shuffles, then a 4-way sum through LDS.
        float *part = (float *)smem;                 // [4 waves][2 outputs][T columns]
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            hs0[q] += __shfl_xor(hs0[q], 16);
            hs1[q] += __shfl_xor(hs1[q], 16);
            hs0[q] += __shfl_xor(hs0[q], 32);
            hs1[q] += __shfl_xor(hs1[q], 32);
            if (g == 0) {
                part[(w * 2 + 0) * T + 16 * q + l16] = hs0[q];
                part[(w * 2 + 1) * T + 16 * q + l16] = hs1[q];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2 * T / 256; ++i) {
            const int idx = tid_e + 256 * i, c = idx / T, o = idx % T;
            const long t = (long)(m0 + o / P) * a.d + phi0 + (o % P);
            if (c < a.nout && t < a.L) {
                const float v = part[(0 * 2 + c) * T + o] + part[(1 * 2 + c) * T + o] + part[(2 * 2 + c) * T + o] +
                                part[(3 * 2 + c) * T + o] + a.out_b[c];
                a.y_out[((size_t)b * a.nout + c) * a.L + t] = fminf(1.0f, fmaxf(-1.0f, v));
            }
        }
    } else {
        __syncthreads();
        const int slot = tid_e & 15, prow = tid_e >> 4;
        const long dt = (long)(16 / P) * a.d;
        long t = (long)(m0 + prow / P) * a.d + phi0 + (prow % P);
        __bf16 *dstp = yb + t * 128 + slot * 8;
        const unsigned char *srcp = smem + prow * 256 + ((slot ^ (prow & 15)) << 4);
#pragma unroll
        for (int i = 0; i < T / 16; ++i) {
            if (t < a.L) *(bf16x8 *)dstp = *(const bf16x8 *)(srcp + i * 4096);
            t += dt;
            dstp += dt * 128;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Main loop of the 256-time tile with B-FRAGMENT REUSE ACROSS TAPS (round 4, REUSE form of the duo kernel).  With P phases per tile the
// B fragment of (tap j, column tile q) is rows P j + 16 q .. + 15 of the LDS image - the same rows as (tap j + 16 / P, column tile q - 1):
// at P = 4 the tap-major loop reads each of the 75 x 4 distinct fragments up to four times (960 ds_read_b128 per tile and wave, one per
// MFMA pair).  Class-major order: taps fall into NCLS = 16 / P classes c = j mod NCLS; for one class and one k-step kk the wave holds the
// A fragments of the class's NU taps x 2 row tiles (double-buffered over the k-steps: A0 / A1, fetched a whole phase ahead) and walks the
// class's 16 + NU - 1 row windows once; window i feeds tap c + NCLS u into column tile i - u.  P = 4: 304 LDS reads per tile instead of
// 960, up to eight MFMAs per read (P = 2: 544, up to four), the weight stream unchanged (128 fragments per tile against 120).  Same
// products, summed class by class instead of tap by tap (fp32 accumulation: results differ from the tap-major loop by rounding only).
// Measured, same box: the bare main loop on realistic operands 1.145 ms against 1.230 (tools/micro/tcn_mainloop_variants.hip reuse16x16 /
// base16x16, profiles/r04_micro_mainloop_reuse.txt); the block kernel at d = 4 ... 2048 1.40 ms against 1.46 (profiles/r04_tcn_forms_reuse.log).
// On entry ring[0..3] = windows 0..3 of (c, kk = 0) and A0 = the fragments of (c, kk = 0); on exit the same for class cn (LAST: the
// next tile's image may not have landed yet - no window of it is read here; A0 is the next tile's first phase: the same weights).
// ------------------------------------------------------------------------------------------------
// WRAP = false (one tile per workgroup): behind the last class there is no next tile - its first weight fragments (32 KB per workgroup, 6.5 % of
// the tile's weight stream) are not requested.
template <int P, int NU, bool LAST, int NUMAX, bool WRAP>
__device__ __forceinline__ void tcn_reuse_class(f32x4 (&acc)[2][16], bf16x8 (&A0)[NUMAX][2], bf16x8 (&A1)[NUMAX][2], bf16x8 (&ring)[4],
                                                const unsigned char *sm, const MstStream16 &wst, unsigned aoff, int c, int cn, int l16, int g) {
    constexpr int NW = 15 + NU, NCLS = 16 / P;
    static_assert(NW >= 4, "the ring is four windows deep");
    const int rsw = (P * c + l16) & 15, rswn = (P * cn + l16) & 15;
    const unsigned char *rowb = sm + (P * c + l16) * 256, *rowbn = sm + (P * cn + l16) * 256;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        bf16x8 (&cur)[NUMAX][2] = (kk & 1) ? A1 : A0;
        bf16x8 (&nxt)[NUMAX][2] = (kk & 1) ? A0 : A1;
#pragma unroll
        for (int u = 0; u < NUMAX; ++u) {
            if (kk < 3 && u >= NU) continue;
            if (LAST && !WRAP && kk == 3) continue;
            int j = (kk < 3 ? c : cn) + NCLS * u;
            j = j < 15 ? j : 14;                                   // the last class has one tap less: that slot holds a fragment nobody uses
            const unsigned so = (unsigned)(j * 4 + (kk < 3 ? kk + 1 : 0)) * 8192u;
            nxt[u][0] = __builtin_bit_cast(bf16x8, mst_stream_load16(wst, aoff, so));
            nxt[u][1] = __builtin_bit_cast(bf16x8, mst_stream_load16(wst, aoff + 4096u, so));
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int n = kk * NW + i;
            const bf16x8 b = ring[n & 3];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int q = i - u;
                if (q >= 0 && q < 16) {
                    acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[u][0], b, acc[0][q], 0, 0, 0);
                    acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[u][1], b, acc[1][q], 0, 0, 0);
                }
            }
            const int n2 = n + 4, k2 = n2 / NW, i2 = n2 % NW;      // the window four ahead: this class, or the first k-step of the next
            if (k2 < 4)
                ring[n & 3] = *(const bf16x8 *)(rowb + (((4 * k2 + g) ^ rsw) << 4) + i2 * 4096);
            else if (!LAST)
                ring[n & 3] = *(const bf16x8 *)(rowbn + ((g ^ rswn) << 4) + i2 * 4096);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// blocks 1..n-1, bf16: the PERSISTENT DOUBLE-TILE form of tcn_block_bf16_kernel ("duo" kernel, round 3) - same tiles, same LDS image, same
// main loop and epilogue arithmetic, bit-identical results; what changes is who does what:
//   * ONE workgroup of EIGHT waves per CU, persistent, walking its share of the tiles of one XCD's contiguous tile range;
//   * waves 0-3 (one per SIMD) are the MATRIX waves: main loop, epilogue arithmetic, transposed tile -> LDS.  They issue no staging
//     load, no DMA and no global store;
//   * waves 4-7 (one per SIMD) are the LOADER waves: while the matrix waves work on tile i (buffer i & 1) they store tile i - 1 out of
//     the other buffer (whole 256-byte rows) and then refill that buffer with tile i + 1 by LDS-DMA (global_load_lds_dwordx4: no VGPR
//     round trip, no ds_write; the XOR swizzle of the LDS image sits on the SOURCE side: lane l of a 4-row piece fetches slot
//     (l & 15) ^ (row & 15)).  An LDS-DMA piece occupies its wave for ~100-150 clocks at issue - on a loader wave that is free;
//   * TWO tile buffers (2 x 78 KB at P = 4) and two workgroup barriers per tile: (1) the matrix waves are done reading tile i AND
//     tile i + 1 has landed, (2) the transposed output tile is complete.
// Measured motivation (MI355X, 32 x 131072, d = 4 ... 2048; profiles/r03_tcn_block_forms_summary.md): the main loop alone runs at
// 1.27 ms per launch at one wave per SIMD as at two; the one-tile-per-workgroup kernel 1.50-1.53 ms; a first persistent double-tile
// form whose four waves did everything themselves 1.60 ms (0.2 ms for issuing the copy, 0.2 ms for the epilogue).
// ------------------------------------------------------------------------------------------------
template <int P, bool FUSE_OUT, int NQ, bool REUSE = false, bool FUSE0 = false>
__global__ __launch_bounds__(512, 1) void tcn_block_bf16_duo_kernel(TcnBlockArgs a) {
    static_assert(!REUSE || ((P == 4 || P == 2) && NQ == 8), "the class-major main loop is written for 256-time tiles of two / four phases");
    static_assert(!FUSE0 || (P == 2 && NQ == 8 && !FUSE_OUT), "block 0 is fused into the d = 2 block's two-phase tiles (consecutive samples)");
    constexpr int T = 32 * NQ, R = T + 14 * P, R4 = (R + 3) / 4 * 4, MT = T / P, NC = 2 * NQ;
    constexpr int NK = R4 / 4;                   // 1 KB DMA pieces (4 rows x 256 B) per tile
    constexpr int NI = (NK + 3) / 4;             // pieces per loader wave
    constexpr int BUF = R4 * 256;
    static_assert(2 * BUF + 2048 <= 160 * 1024, "two tiles + parameters fit the CU's LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUF];
    __shared__ __attribute__((aligned(16))) float par[4 * 128];     // shift | FiLM r | FiLM b | res
    constexpr int XWP = 304;                                         // FUSE0: waveform samples per channel a tile needs (R + 14 = 298), padded
    __shared__ float xs0[FUSE0 ? 4 * 2 * XWP : 1];                   // FUSE0: one private waveform window per loader wave
    __shared__ __attribute__((aligned(16))) float par0[FUSE0 ? 4 * 128 : 4];          // FUSE0: block 0's shift | FiLM r | FiLM b | res for the loader waves
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wv >= 4;
    const int w = wv & 3;                        // matrix wave: its channel quarter; loader wave: its share of pieces / rows
    const int l16 = lane & 15, g = lane >> 4;

    // ---- this workgroup's tiles: workgroup i runs on XCD i % 8 and walks tiles (i % 8) * xcd_tiles + i / 8 + n * gridDim.x / 8
    // (32-bit tile numbers - the launcher checks the count: a 64-bit division is ~100 instructions, taken from the matrix wave's issue slots)
    const unsigned ntiles = (unsigned)a.B * (unsigned)a.tiles_phase * (unsigned)a.tiles_step;
    unsigned tile, tstep, tend;
    if (a.xcd_tiles > 0) {
        tile = (blockIdx.x & 7) * (unsigned)a.xcd_tiles + (blockIdx.x >> 3);
        tstep = gridDim.x >> 3;
        tend = ((blockIdx.x & 7) + 1) * (unsigned)a.xcd_tiles;
        if (tend > ntiles) tend = ntiles;
    } else {
        tile = blockIdx.x;
        tstep = gridDim.x;
        tend = ntiles;
    }
    if (tile >= tend) return;       // uniform

    auto tile_geometry = [&](unsigned tl, int &b, int &m0, int &phi0) {
        const unsigned r = tl / (unsigned)a.tiles_step, mg = tl - r * (unsigned)a.tiles_step;
        const unsigned bb = r / (unsigned)a.tiles_phase;
        phi0 = (int)(r - bb * (unsigned)a.tiles_phase) * P;
        b = (int)bb;
        m0 = (int)mg * MT;
    };
    // loader waves: all pieces of tile (b, m0, phi0) that belong to this wave (k = w mod 4), into buffer buf.  A tile that lies inside the
    // segment (all but the first / last of a phase group) needs no bounds test: the pieces of a wave are 16 rows = 16 / P steps apart,
    // one 64-bit add per piece (the loader shares its SIMD's issue slots with a matrix wave: every instruction here is taken from it)
    auto dma_tile = [&](int b, int m0, int phi0, int buf) {
        static_assert(16 % P == 0, "pieces advance by a whole number of steps");
        const int row0 = 4 * w + (lane >> 4);
        const long t0 = (long)(m0 + row0 / P - 7) * a.d + phi0 + (row0 % P);
        const long dt = (long)(16 / P) * a.d;
        const int slot = (lane & 15) ^ (row0 & 15);          // row & 15 is the same for all pieces of a lane
        const long t_first = (long)(m0 - 7) * a.d + phi0, t_last = (long)(m0 + (R4 - 1) / P - 7) * a.d + phi0 + (P - 1);
        unsigned char *dst = smem + buf * BUF + w * 1024;
        if (t_first >= 0 && t_last < a.L) {                   // uniform
            const unsigned char *src = (const unsigned char *)a.x + ((size_t)b * a.Lp + t0) * 256 + slot * 16;
#pragma unroll 1
            for (int k = w; k < NK; k += 4) {
                mst_dma16(src, dst);
                src += dt * 256;
                dst += 4096;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int row = row0 + 16 * i;
                const long t = t0 + i * dt;
                const bool ok = row < R && t >= 0 && t < a.L;
                const unsigned char *src = (ok ? (const unsigned char *)a.x + ((size_t)b * a.Lp + t) * 256 : (const unsigned char *)a.zeros) + slot * 16;
                if (w + 4 * i < NK) mst_dma16(src, dst + i * 4096);
            }
        }
    };
    // FUSE0, loader waves: the rows of tile (b, m0) are block 0's outputs at the consecutive times t_first + r (d = P = 2, one phase group) -
    // computed here exactly like tcn_block0_mfma_kernel computes them (same fragments, same MFMA order, same epilogue: the same bits) and
    // written where the DMA would have put them; rows outside the segment are zero rows (this block's padding).  Loader wave w owns the
    // 32-row column tiles q = w, w + 4, w + 8 of the image - ALL 128 channels of them, and (see the store phase) the same rows of the
    // output tile: a wave only overwrites rows it has stored out itself, no barrier between the loader waves.  Its waveform window is its own.
    auto compute_tile0 = [&](int b, int m0, int phi0, int buf) {
        if constexpr (FUSE0) {
            const int ln = lane & 31, h = lane >> 5;
            float *xw0 = xs0 + w * 2 * XWP, *xw1 = xw0 + XWP;
            const long t_first = (long)(m0 - 7) * a.d + phi0;           // time of row 0
            __builtin_amdgcn_wave_barrier();          // every lane is past its reads of the rows / the window this call overwrites
            for (int i = lane; i < 2 * XWP; i += 64) {
                const int ci = i >= XWP ? 1 : 0, k = i - ci * XWP;
                const long t = t_first - 7 + k;
                (ci ? xw1 : xw0)[k] = (t >= 0 && t < a.L) ? a.x0[((size_t)b * 2 + ci) * a.L + t] : 0.0f;
            }
            unsigned char *img = smem + buf * BUF;
            // block 0's parameters with the FiLM row of item b: ONE copy that every loader wave writes in full - they fill the same tile between the
            // same two workgroup barriers, so concurrent writers write the same values, and a wave reads behind its own fence
            float *pw = par0;
            {
                const float *frow0 = a.film0 + (a.film_rows > 1 ? (size_t)b * 256 : 0);
                for (int i = lane; i < 128; i += 64) {
                    pw[i] = a.shift0[i];
                    pw[128 + i] = frow0[i];
                    pw[256 + i] = frow0[128 + i];
                    pw[384 + i] = a.res0[i];
                }
            }
            mst_wave_lds_fence();                                       // the window and the parameters are in LDS
            const bf16x8 *const w0p = (const bf16x8 *)a.w0pk + lane;
            // rows 0 .. 255 (the rows the store phase reads): 32-row groups w and w + 4, all four channel quarters; the halo group 8 (rows 256 ..
            // 283, never stored) is shared: every wave computes ITS quarter of it - nine (group, quarter) units per wave
            static_assert((R + 31) / 32 == 9, "eight stored row groups + one halo group");
#pragma unroll 1
            for (int q = w; q < 12; q += 4) {
                const bool halo = q >= 8;                                  // third trip: group 8, one quarter
                if (halo) q = 8;
                bf16x8 bh[2], bl[2];
#pragma unroll
                for (int sI = 0; sI < 2; ++sI) tcn_block0_bfrag(xw0, xw1, 32 * q + ln, sI, h, bh[sI], bl[sI]);
                const int o = 32 * q + ln;
                const long t = t_first + o;
                const bool inside = t >= 0 && t < a.L;
#pragma unroll 1
                for (int cw = halo ? w : 0; cw < (halo ? w + 1 : 4); ++cw) {      // channel quarters (what the four waves of the block-0 kernel do)
                    f32x16 acc;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const f32x4 sh = *(const f32x4 *)(pw + 32 * cw + 8 * gq + 4 * h);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[4 * gq + i] = sh[i];
                    }
#pragma unroll
                    for (int sI = 0; sI < 2; ++sI) {
                        const bf16x8 af = w0p[(sI ? 256 : 0) + 64 * cw];
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bh[sI], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bl[sI], acc, 0, 0, 0);
                    }
                    const float xres = ((cw >> 1) ? xw1 : xw0)[o + 7];      // grouped residual: channels 0..63 read input 0, 64..127 input 1
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int co0 = 32 * cw + 8 * gq + 4 * h;
                        const f32x4 fr = *(const f32x4 *)(pw + 128 + co0);
                        const f32x4 fb = *(const f32x4 *)(pw + 256 + co0);
                        const f32x4 rs = *(const f32x4 *)(pw + 384 + co0);
                        bf16x4 out;
#pragma unroll
                        for (int i = 0; i < 4; ++i) out[i] = inside ? tcn_block0_out(acc[4 * gq + i], fr[i], fb[i], rs[i], xres) : (__bf16)0.0f;
                        if (o < R) *(bf16x4 *)(img + o * 256 + (((co0 >> 3) ^ (o & 15)) << 4) + 8 * h) = out;
                    }
                }
            }
        }
    };
    auto fill_tile = [&](int b, int m0, int phi0, int buf) {
        if constexpr (FUSE0) compute_tile0(b, m0, phi0, buf);
        else dma_tile(b, m0, phi0, buf);
    };
    auto stage_film = [&](int b) {          // matrix waves only (tid < 256)
        if (tid < 128) {
            const float *frow0 = a.film + (a.film_rows > 1 ? (size_t)b * 256 : 0);
            par[128 + tid] = frow0[tid];
            par[256 + tid] = frow0[128 + tid];
        }
    };

    int tb, tm0, tphi0;
    tile_geometry(tile, tb, tm0, tphi0);
    if (loader) {
        fill_tile(tb, tm0, tphi0, 0);
    } else {
        if (tid < 128) {
            par[tid] = a.shift[tid];
            par[384 + tid] = a.res[tid];
        }
        stage_film(tb);
    }
    mst_dma_wait_barrier<0>();

    if (loader) {
        // =================================================================== loader waves
        // iteration i: [next tile -> the other buffer] (barrier 1 of tile i) (barrier 2 of tile i) [tile i's output rows -> global memory]
        int cur = 0;
        const int lt = tid - 256;                      // 0..255: thread (prow, slot) of the row passes, like the one-tile kernel's
        for (;;) {
            const int b = tb, m0 = tm0, phi0 = tphi0;
            const unsigned tnext = tile + tstep;
            const bool has_next = tnext < tend;
            if (has_next) {
                tile_geometry(tnext, tb, tm0, tphi0);
                fill_tile(tb, tm0, tphi0, cur ^ 1);     // the buffer whose rows this wave stored out itself one iteration ago
            }
            mst_dma_wait_barrier<0>();                  // (1) the next tile has landed (and this wave's stores have left)
            if constexpr (FUSE_OUT) {
                mst_dma_wait_barrier<63>();             // (2a) the output head's partial sums are complete (matrix waves finish the tile)
                mst_dma_wait_barrier<63>();             // (2b) ... and have been read: the buffer may be refilled
            } else {
                mst_dma_wait_barrier<63>();             // (2) the transposed output tile is complete
                const unsigned char *sm = smem + cur * BUF;
                __bf16 *yb = (__bf16 *)a.y + (size_t)b * a.Lp * 128;
                if constexpr (FUSE0) {
                    // loader wave w stores the rows it will overwrite with the next tile: the 32-row groups q = w, w + 4 (d = P = 2: row o is time 2 m0 + o)
                    const int slot = lane & 15, rsub = lane >> 4;
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int o = 32 * (w + 4 * qq) + rsub + 4 * i;
                            const long t = (long)(m0 + o / P) * a.d + phi0 + (o % P);
                            if (t < a.L) *(bf16x8 *)(yb + t * 128 + slot * 8) = *(const bf16x8 *)(sm + o * 256 + ((slot ^ (o & 15)) << 4));
                        }
                } else {
                const int slot = lt & 15, prow = lt >> 4;
                const long dt = (long)(16 / P) * a.d;
                long t = (long)(m0 + prow / P) * a.d + phi0 + (prow % P);
                __bf16 *dstp = yb + t * 128 + slot * 8;
                const unsigned char *srcp = sm + prow * 256 + ((slot ^ (prow & 15)) << 4);
                if ((long)(m0 + (T - 1) / P) * a.d + phi0 + (P - 1) < a.L) {        // uniform: every row of the tile is inside the segment
#pragma unroll
                    for (int i = 0; i < T / 16; ++i) {
                        *(bf16x8 *)dstp = *(const bf16x8 *)(srcp + i * 4096);
                        dstp += dt * 128;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < T / 16; ++i) {
                        if (t < a.L) *(bf16x8 *)dstp = *(const bf16x8 *)(srcp + i * 4096);
                        t += dt;
                        dstp += dt * 128;
                    }
                }
                }
            }
            if (!has_next) break;
            tile = tnext;
            cur ^= 1;
        }
        return;
    }

    // ======================================================================= matrix waves
    __builtin_amdgcn_s_setprio(2);          // the loader wave of the SIMD takes the issue slots this wave leaves, never the other way round
    // A fragments: wpk[ks = j*4 + kk][row tile m][wave][lane] = 8 bf16; a ring of four k-steps that runs on across tiles (every tile
    // multiplies by the same weights: behind the last tap the fragments of tap 0 are requested again)
    const MstStream16 wst = mst_stream16(a.wpk, 60u * 2u * 4096u);
    const unsigned aoff = (unsigned)(w * 64 + lane) * 16u;
    constexpr int RB = 8;
    static_assert(NC % RB == 0, "the ring divides the column tiles");
    bf16x8 af[2][4], bf[RB];
    constexpr int NCLS = 16 / P, NUMAX = (15 + NCLS - 1) / NCLS;      // REUSE: tap classes, taps per class (the last class: 15 / NCLS)
    bf16x8 A0[NUMAX][2], A1[NUMAX][2], ring[4];                      // REUSE: the class-major loop's operands (tcn_reuse_class)
    if constexpr (REUSE) {
#pragma unroll
        for (int u = 0; u < NUMAX; ++u) {
#pragma unroll
            for (int m = 0; m < 2; ++m) A0[u][m] = __builtin_bit_cast(bf16x8, mst_stream_load16(wst, aoff + m * 4096, (unsigned)(4 * NCLS * u) * 8192u));
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int m = 0; m < 2; ++m) af[m][kk] = __builtin_bit_cast(bf16x8, mst_stream_load16(wst, aoff + m * 4096, (unsigned)kk * 8192u));
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    int cur = 0, bprev = tb;
    // (round 5, phase clocks of this wave - tools/probe_tcn_phases.py: the per-tile bookkeeping, i.e. the two 32-bit divisions of
    //  tile_geometry, cost 1250 of the tile's 42 600 clocks.  A matrix wave needs the next tile's coordinates only for the fused output
    //  head; otherwise only its batch item, and that only when every item has its own FiLM row: one division, or none)
    const unsigned tiles_item = (unsigned)a.tiles_phase * (unsigned)a.tiles_step;
    for (;;) {
        const int b = tb, m0 = tm0, phi0 = tphi0;
        const unsigned tnext = tile + tstep;
        const bool has_next = tnext < tend;
        if (has_next) {
            if constexpr (FUSE_OUT) tile_geometry(tnext, tb, tm0, tphi0);
            else if (a.film_rows > 1) tb = (int)(tnext / tiles_item);
        }
        unsigned char *const sm = smem + cur * BUF;
        if (b != bprev) {              // a new batch item: its FiLM row (every matrix wave is past the previous tile's epilogue: barrier 2)
            stage_film(b);
            bprev = b;
        }

        f32x4 acc[2][NC];
#pragma unroll
        for (int m = 0; m < 2; ++m) {          // the accumulators start from the BN shift of their channel
            const f32x4 sh = *(const f32x4 *)(par + 32 * w + 16 * m + 4 * g);
#pragma unroll
            for (int q = 0; q < NC; ++q) acc[m][q] = sh;
        }
        if constexpr (REUSE) {
            {
                const unsigned char *rp0 = sm + l16 * 256 + ((g ^ l16) << 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) ring[i] = *(const bf16x8 *)(rp0 + i * 4096);
            }
#pragma unroll 1
            for (int c = 0; c < NCLS - 1; ++c) tcn_reuse_class<P, NUMAX, false, NUMAX>(acc, A0, A1, ring, sm, wst, aoff, c, c + 1, l16, g);
            tcn_reuse_class<P, 15 / NCLS, true, NUMAX>(acc, A0, A1, ring, sm, wst, aoff, NCLS - 1, 0, l16, g);
        } else {
        {
            const unsigned char *rp0 = sm + l16 * 256 + ((g ^ l16) << 4);
#pragma unroll
            for (int q = 0; q < RB; ++q) bf[q] = *(const bf16x8 *)(rp0 + q * 4096);
        }
#pragma unroll 1
        for (int j = 0; j < 15; ++j) {
            const int jn = j < 14 ? j + 1 : 0;
            const int rb0 = j * P + l16, rb1 = (j < 14 ? j + 1 : 14) * P + l16;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int rbn = (kk == 3) ? rb1 : rb0;
                const int kn = (kk + 1) & 3;
                const unsigned char *cp = sm + rb0 * 256 + (((4 * kk + g) ^ (rb0 & 15)) << 4);
                const unsigned char *np = sm + rbn * 256 + (((4 * kn + g) ^ (rbn & 15)) << 4);
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][kk], bf[q % RB], acc[0][q], 0, 0, 0);
                    acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][kk], bf[q % RB], acc[1][q], 0, 0, 0);
                    bf[q % RB] = (q + RB < NC) ? *(const bf16x8 *)(cp + (q + RB) * 4096) : *(const bf16x8 *)(np + (q + RB - NC) * 4096);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) af[m][kk] = __builtin_bit_cast(bf16x8, mst_stream_load16(wst, aoff + m * 4096, (unsigned)(jn * 4 + kk) * 8192u));
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            }
        }
        }

        // ---- epilogue arithmetic (the one of tcn_block_bf16_kernel): residual rows -> registers, barrier, transposed tile -> LDS, barrier
        // (the residual rows of row tile 1 are read behind the barrier: they sit in this wave's own channel columns, which no other wave
        //  writes and which its own row-tile-0 results do not touch - 32 registers instead of 64 next to the accumulators)
        // (lane coordinates made opaque once per tile: otherwise hipcc keeps ~36 tile-invariant LDS addresses of this epilogue live across
        //  the main loop - and, at the 256-register limit of an eight-wave workgroup, spills them)
        int l16e = l16, ge = g;
        asm volatile("" : "+v"(l16e), "+v"(ge));
        bf16x4 xin[NC];
        auto read_xin = [&](int m) {
            const int co0 = 32 * w + 16 * m + 4 * ge;
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int row = 16 * q + l16e + 7 * P;
                xin[q] = *(const bf16x4 *)(sm + row * 256 + (((co0 >> 3) ^ (row & 15)) << 4) + 2 * (co0 & 7));
            }
        };
        read_xin(0);
        mst_dma_wait_barrier<63>();            // (1) every matrix wave is done reading this tile (the weight fragments in flight stay in flight)
        float hs0[NC], hs1[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) hs0[q] = hs1[q] = 0.0f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (m) {
                read_xin(1);
                __builtin_amdgcn_wave_barrier();      // all lanes of the wave have read before any of them writes these columns (lockstep on the GPU)
            }
            const int co0 = 32 * w + 16 * m + 4 * ge;
            const f32x4 fr = *(const f32x4 *)(par + 128 + co0);
            const f32x4 fb = *(const f32x4 *)(par + 256 + co0);
            const f32x4 rs = *(const f32x4 *)(par + 384 + co0);
            f32x4 ow0 = {0.0f, 0.0f, 0.0f, 0.0f}, ow1 = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (FUSE_OUT) {
                ow0 = *(const f32x4 *)(a.out_w + co0);
                if (a.nout > 1) ow1 = *(const f32x4 *)(a.out_w + 128 + co0);
            }
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int o = 16 * q + l16e;
                const float v4[4] = {acc[m][q][0], acc[m][q][1], acc[m][q][2], acc[m][q][3]};
                const bf16x4 out = tcn_epilogue4(v4, fr, fb, rs, xin[q]);
                if constexpr (FUSE_OUT) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        hs0[q] = fmaf(ow0[i], (float)out[i], hs0[q]);
                        hs1[q] = fmaf(ow1[i], (float)out[i], hs1[q]);
                    }
                }
                if constexpr (!FUSE_OUT) *(bf16x4 *)(sm + o * 256 + (((co0 >> 3) ^ (o & 15)) << 4) + 2 * (co0 & 7)) = out;
            }
        }
        if constexpr (FUSE_OUT) {
            float *part = (float *)sm;                 // [4 waves][2 outputs][T columns]
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                hs0[q] += __shfl_xor(hs0[q], 16);
                hs1[q] += __shfl_xor(hs1[q], 16);
                hs0[q] += __shfl_xor(hs0[q], 32);
                hs1[q] += __shfl_xor(hs1[q], 32);
                if (g == 0) {
                    part[(w * 2 + 0) * T + 16 * q + l16] = hs0[q];
                    part[(w * 2 + 1) * T + 16 * q + l16] = hs1[q];
                }
            }
            mst_dma_wait_barrier<63>();          // (2a)
#pragma unroll
            for (int i = 0; i < 2 * T / 256; ++i) {
                const int idx = tid + 256 * i, c = idx / T, o = idx % T;
                const long t = (long)(m0 + o / P) * a.d + phi0 + (o % P);
                if (c < a.nout && t < a.L) {
                    const float v = part[(0 * 2 + c) * T + o] + part[(1 * 2 + c) * T + o] + part[(2 * 2 + c) * T + o] +
                                    part[(3 * 2 + c) * T + o] + a.out_b[c];
                    a.y_out[((size_t)b * a.nout + c) * a.L + t] = fminf(1.0f, fmaxf(-1.0f, v));
                }
            }
            mst_dma_wait_barrier<63>();          // (2b) the partial sums have been read: the loader waves may refill this buffer
        } else {
            mst_dma_wait_barrier<63>();          // (2) the transposed output tile is complete: the loader waves store it
        }
        if (!has_next) break;
        tile = tnext;
        cur ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------
// blocks 1..n-1, "bf16x3" mode: fp32-class accuracy on the bf16 matrix cores.  Every fp32 operand is split
//     x = x_hi + x_lo,  x_hi = bf16(x),  x_lo = bf16(x - x_hi)          (16 significant bits; products of two bf16 values are exact in fp32)
// for the activations (while the tile is staged into LDS: two tiles, hi and lo) and for the BN-folded weights (on the host,
// two fragment images), and the contraction keeps the three leading terms
//     W x ~= W_hi x_hi + W_hi x_lo + W_lo x_hi                          (the dropped W_lo x_lo term is ~2^-16 of the result)
// as three v_mfma_f32_16x16x32_bf16 into the same fp32 accumulator.  Activations stay fp32 in HBM (1024 B of traffic per
// time step against 3 * 491 520 FLOP: compute bound), the epilogue is the exact fp32 one of the parity kernel and reads the
// residual input from global memory (L2-hot: the tile's centre rows were just staged).  Measured against the oracle: 5e-6
// max-abs on the output waveform (exact-fp32 mode: 9e-7; plain bf16 mode: 4e-3) at three MFMAs per bf16-mode MFMA.
// Same polyphase tiling, LDS swizzle and fragment order as tcn_block_bf16_kernel; one workgroup per CU (two 78 KB tiles).
// ------------------------------------------------------------------------------------------------
// The class-major loop (tcn_reuse_class) in split arithmetic: per window one hi and one lo B fragment, per tap of the class three MFMAs per
// row tile (W_lo x_hi, W_hi x_lo, W_hi x_hi: the small terms first, like the tap-major loop).  Two phases per tile (P = 2): eight classes of
// two taps (the last: one) - every B fragment pair is read once per class instead of once per tap.
template <int P, int NU, bool LASTC, int NUMAX, int NC>
__device__ __forceinline__ void tcn_reuse_class_x3(f32x4 (&acc)[2][NC], bf16x8 (&H0)[NUMAX][2], bf16x8 (&L0)[NUMAX][2], bf16x8 (&H1)[NUMAX][2],
                                                   bf16x8 (&L1)[NUMAX][2], bf16x8 (&rh)[4], bf16x8 (&rl)[4], const unsigned char *sm_hi,
                                                   const unsigned char *sm_lo, const unsigned char *wbase, size_t lo_img, unsigned aoff, int c, int cn,
                                                   int l16, int g) {
    constexpr int NW = NC - 1 + NU, NCLS = 16 / P;
    static_assert(NW >= 4, "the ring is four windows deep");
    const int rsw = (P * c + l16) & 15, rswn = (P * cn + l16) & 15;
    const int rowb = (P * c + l16) * 256, rowbn = (P * cn + l16) * 256;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        bf16x8 (&ch)[NUMAX][2] = (kk & 1) ? H1 : H0;
        bf16x8 (&cl)[NUMAX][2] = (kk & 1) ? L1 : L0;
        bf16x8 (&nh)[NUMAX][2] = (kk & 1) ? H0 : H1;
        bf16x8 (&nl)[NUMAX][2] = (kk & 1) ? L0 : L1;
        if (!(LASTC && kk == 3)) {          // behind the tile's last phase nothing is fetched (one tile per workgroup)
#pragma unroll
            for (int u = 0; u < NUMAX; ++u) {
                if (kk < 3 && u >= NU) continue;
                int j = (kk < 3 ? c : cn) + NCLS * u;
                j = j < 15 ? j : 14;
                const size_t so = (size_t)((j * 4 + (kk < 3 ? kk + 1 : 0)) * 2) * 4096 + aoff;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    nh[u][m] = *(const bf16x8 *)(wbase + so + (size_t)m * 4096);
                    nl[u][m] = *(const bf16x8 *)(wbase + lo_img + so + (size_t)m * 4096);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int n = kk * NW + i;
            const bf16x8 bh = rh[n & 3], bl = rl[n & 3];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int q = i - u;
                if (q >= 0 && q < NC) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cl[u][m], bh, acc[m][q], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch[u][m], bl, acc[m][q], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch[u][m], bh, acc[m][q], 0, 0, 0);
                }
            }
            const int n2 = n + 4, k2 = n2 / NW, i2 = n2 % NW;
            if (k2 < 4) {
                const int ofs = rowb + (((4 * k2 + g) ^ rsw) << 4) + i2 * 4096;
                rh[n & 3] = *(const bf16x8 *)(sm_hi + ofs);
                rl[n & 3] = *(const bf16x8 *)(sm_lo + ofs);
            } else if (!LASTC) {
                const int ofs = rowbn + ((g ^ rswn) << 4) + i2 * 4096;
                rh[n & 3] = *(const bf16x8 *)(sm_hi + ofs);
                rl[n & 3] = *(const bf16x8 *)(sm_lo + ofs);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int P, int NQ>
__global__ __launch_bounds__(256, (NQ == 4 && P <= 2) ? 2 : 1) void tcn_block_bf16x3_kernel(TcnBlockArgs a) {
    constexpr int T = 32 * NQ, R = T + 14 * P, MT = T / P;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * R * 256];      // [hi | lo] tiles; later the fp32 output tile (T x 512 B <= 2 R x 256 B)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned char *const sm_hi = smem, *const sm_lo = smem + R * 256;

    int tile = a.xcd_tiles > 0 ? (int)(blockIdx.x & 7) * a.xcd_tiles + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int mg = tile % a.tiles_step;
    tile /= a.tiles_step;
    const int pg = tile % a.tiles_phase;
    const int b = tile / a.tiles_phase;
    const int m0 = mg * MT, phi0 = pg * P;
    const float *xb = (const float *)a.x + (size_t)b * a.Lp * 128;
    float *yb = (float *)a.y + (size_t)b * a.Lp * 128;

    // ---- stage the rows: a thread owns one 16-byte slot (8 channels) of rows prow, prow + 16, ...; fp32 in, (hi, lo) bf16 out
    {
        static_assert(16 % P == 0, "row passes advance by a whole number of steps");
        const int slot = tid & 15, prow = tid >> 4;
        constexpr int NPASS = (R + 15) / 16, HALF = (NPASS + 1) / 2;
        const long dt = (long)(16 / P) * a.d;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            long t = (long)(m0 + prow / P - 7) * a.d + phi0 + (prow % P) + (long)half * HALF * dt;
            const float *src = xb + t * 128 + slot * 8;
            const float *zsrc = (const float *)a.zeros + slot * 8;
            f32x4 v0[HALF], v1[HALF];
#pragma unroll
            for (int i = 0; i < HALF; ++i) {
                // never a predicated load (hipcc branches around one and waits for it before the next: the row passes of a half ran one
                // memory latency after the other): rows outside the segment are fetched from the zero page
                const float *p = (prow + 16 * (half * HALF + i) < R && t >= 0 && t < a.L) ? src : zsrc;
                v0[i] = *(const f32x4 *)p;
                v1[i] = *(const f32x4 *)(p + 4);
                t += dt;
                src += dt * 128;
            }
            const int off = prow * 256 + ((slot ^ (prow & 15)) << 4) + half * HALF * 4096;
#pragma unroll
            for (int i = 0; i < HALF; ++i) {
                if (prow + 16 * (half * HALF + i) < R) {
                    bf16x8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[e] = (__bf16)v0[i][e];
                        lo[e] = (__bf16)(v0[i][e] - (float)hi[e]);
                        hi[4 + e] = (__bf16)v1[i][e];
                        lo[4 + e] = (__bf16)(v1[i][e] - (float)hi[4 + e]);
                    }
                    *(bf16x8 *)(sm_hi + off + i * 4096) = hi;
                    *(bf16x8 *)(sm_lo + off + i * 4096) = lo;
                }
            }
        }
    }
    __syncthreads();

    // v_mfma_f32_16x16x32_bf16 like the bf16 kernel (same operand traffic per FLOP as the 32 x 32 x 16 form, more throughput under the
    // power limit): two row tiles of 16 channels x NC column tiles of 16 times per wave
    constexpr int NC = 2 * NQ;
    const int l16 = lane & 15, g = lane >> 4;
    f32x4 acc[2][NC];
#pragma unroll
    for (int m = 0; m < 2; ++m) {          // accumulators start from the BN shift of their channel
        const f32x4 sh = *(const f32x4 *)(a.shift + 32 * w + 16 * m + 4 * g);
#pragma unroll
        for (int q = 0; q < NC; ++q) acc[m][q] = sh;
    }

    // A fragments: wpk[part][ks = j*4 + kk][row tile m][wave][lane] = 8 bf16, part 0 = hi, 1 = lo (each image 120 * 4096 bytes)
    const unsigned char *wbase = (const unsigned char *)a.wpk;
    const unsigned aoff = (unsigned)(w * 64 + lane) * 16u;
    constexpr size_t LO_IMG = (size_t)120 * 4096;
    // ring of 2 k-steps of A fragments (hi and lo, two row tiles): the set of k-step ks + 2 is requested from L2 when ks has been consumed;
    // ring of 8 column tiles of B fragments (hi and lo)
    if constexpr (P == 2 && NQ == 4) {
        // class-major (round 4): B fragment pairs reused by the two taps of a class (j mod 8) - 576 instead of 960 LDS reads per tile
        constexpr int NCLS = 16 / P, NUMAX = 2;
        bf16x8 H0[NUMAX][2], L0[NUMAX][2], H1[NUMAX][2], L1[NUMAX][2], rh[4], rl[4];
#pragma unroll
        for (int u = 0; u < NUMAX; ++u)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                H0[u][m] = *(const bf16x8 *)(wbase + (size_t)((NCLS * u * 4) * 2 + m) * 4096 + aoff);
                L0[u][m] = *(const bf16x8 *)(wbase + LO_IMG + (size_t)((NCLS * u * 4) * 2 + m) * 4096 + aoff);
            }
        {
            const int o0 = l16 * 256 + ((g ^ l16) << 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rh[i] = *(const bf16x8 *)(sm_hi + o0 + i * 4096);
                rl[i] = *(const bf16x8 *)(sm_lo + o0 + i * 4096);
            }
        }
#pragma unroll 1
        for (int c = 0; c < NCLS - 1; ++c)
            tcn_reuse_class_x3<P, NUMAX, false, NUMAX, NC>(acc, H0, L0, H1, L1, rh, rl, sm_hi, sm_lo, wbase, LO_IMG, aoff, c, c + 1, l16, g);
        tcn_reuse_class_x3<P, 15 / NCLS, true, NUMAX, NC>(acc, H0, L0, H1, L1, rh, rl, sm_hi, sm_lo, wbase, LO_IMG, aoff, NCLS - 1, 0, l16, g);
    } else {
        constexpr int RB = 8;
        static_assert(NC % RB == 0, "the ring divides the column tiles");
        bf16x8 ah[2][2], al[2][2], bh[RB], bl[RB];
    #pragma unroll
        for (int kk = 0; kk < 2; ++kk)
    #pragma unroll
            for (int m = 0; m < 2; ++m) {
                ah[kk][m] = *(const bf16x8 *)(wbase + (size_t)(kk * 2 + m) * 4096 + aoff);
                al[kk][m] = *(const bf16x8 *)(wbase + LO_IMG + (size_t)(kk * 2 + m) * 4096 + aoff);
            }
        {
            const int o0 = l16 * 256 + ((g ^ l16) << 4);
    #pragma unroll
            for (int q = 0; q < RB; ++q) {
                bh[q] = *(const bf16x8 *)(sm_hi + o0 + q * 4096);
                bl[q] = *(const bf16x8 *)(sm_lo + o0 + q * 4096);
            }
        }
        for (int j = 0; j < 15; ++j) {
            const int jn = j < 14 ? j + 1 : 14;
            const int rb0 = j * P + l16, rb1 = jn * P + l16;
    #pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int rbn = (kk == 3) ? rb1 : rb0;
                const int kn = (kk + 1) & 3;
                const int oc = rb0 * 256 + (((4 * kk + g) ^ (rb0 & 15)) << 4);
                const int on = rbn * 256 + (((4 * kn + g) ^ (rbn & 15)) << 4);
                const int s = kk & 1;
                // per column tile six MFMAs (two row tiles x three terms, the small terms first): consecutive MFMAs alternate between the two
                // accumulators; the fragment pair eight column tiles ahead is requested behind them
    #pragma unroll
                for (int q = 0; q < NC; ++q) {
    #pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[s][m], bh[q % RB], acc[m][q], 0, 0, 0);
    #pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[s][m], bl[q % RB], acc[m][q], 0, 0, 0);
    #pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[s][m], bh[q % RB], acc[m][q], 0, 0, 0);
                    const int ofs = (q + RB < NC) ? oc + (q + RB) * 4096 : on + (q + RB - NC) * 4096;
                    bh[q % RB] = *(const bf16x8 *)(sm_hi + ofs);
                    bl[q % RB] = *(const bf16x8 *)(sm_lo + ofs);
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                int ksn = j * 4 + kk + 2;
                ksn = ksn < 60 ? ksn : 59;
    #pragma unroll
                for (int m = 0; m < 2; ++m) {
                    ah[s][m] = *(const bf16x8 *)(wbase + (size_t)(ksn * 2 + m) * 4096 + aoff);
                    al[s][m] = *(const bf16x8 *)(wbase + LO_IMG + (size_t)(ksn * 2 + m) * 4096 + aoff);
                }
            }
        }
    }

    // ---- exact fp32 epilogue: LeakyReLU -> FiLM in the accumulator layout, transposed through LDS (the input tiles are dead; fp32 rows of
    //      512 B, 16-byte slots XOR-swizzled by the row), then whole rows: + res * x_in (x_in re-read from global memory, L2-hot) -> store.
    //      Straight from the accumulator layout a lane's 16 bytes are a 64-byte run of its row (16 rows per instruction, for the residual
    //      load as for the store): measured on the bf16 twin of this access pattern at ~0.3 ms per launch (profiles/r03_tcn_block_forms_summary.md)
    const float *frow = a.film + (a.film_rows > 1 ? (size_t)b * 256 : 0);
    __syncthreads();                       // every wave is done reading the input tiles
    float *st = (float *)smem;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int co0 = 32 * w + 16 * m + 4 * g;
        const f32x4 fr = *(const f32x4 *)(frow + co0);
        const f32x4 fb = *(const f32x4 *)(frow + 128 + co0);
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int o = 16 * q + l16;
            f32x4 z;
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = fr[i] * leaky_relu(acc[m][q][i]) + fb[i];
            *(f32x4 *)(st + o * 128 + (((co0 >> 2) ^ (o & 31)) << 2)) = z;
        }
    }
    __syncthreads();
    {
        const int s4 = tid & 31;                                   // this thread's 4 channels, the same in every pass
        const f32x4 rs = *(const f32x4 *)(a.res + 4 * s4);
        if (a.y_out) {
            // last block: the 1x1 output conv + bias + clamp(-1, 1) (reference architectures.py:133,145) on the finished rows - a row's 128
            // channels sit in 32 lanes (half a wave): 4 products per lane, a 5-step butterfly; the activation itself is not stored
            // (the separate head kernel read 2.1 GB for it at 32 x 131072)
            const f32x4 ow0 = *(const f32x4 *)(a.out_w + 4 * s4);
            const f32x4 ow1 = a.nout > 1 ? *(const f32x4 *)(a.out_w + 128 + 4 * s4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < T / 8; ++i) {
                const int o = (tid >> 5) + 8 * i;
                const long t = (long)(m0 + o / P) * a.d + phi0 + (o % P);
                float h0 = 0.0f, h1 = 0.0f;
                if (t < a.L) {
                    const f32x4 z = *(const f32x4 *)(st + o * 128 + ((s4 ^ (o & 31)) << 2));
                    const f32x4 xin = *(const f32x4 *)(xb + t * 128 + 4 * s4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float out = z[k] + rs[k] * xin[k];
                        h0 = fmaf(ow0[k], out, h0);
                        h1 = fmaf(ow1[k], out, h1);
                    }
                }
#pragma unroll
                for (int mm = 16; mm >= 1; mm >>= 1) {
                    h0 += __shfl_xor(h0, mm);
                    h1 += __shfl_xor(h1, mm);
                }
                if (s4 < a.nout && t < a.L)
                    a.y_out[((size_t)b * a.nout + s4) * a.L + t] = fminf(1.0f, fmaxf(-1.0f, (s4 ? h1 : h0) + a.out_b[s4]));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < T / 8; ++i) {
            const int o = (tid >> 5) + 8 * i;
            const long t = (long)(m0 + o / P) * a.d + phi0 + (o % P);
            if (t < a.L) {
                const f32x4 z = *(const f32x4 *)(st + o * 128 + ((s4 ^ (o & 31)) << 2));
                const f32x4 xin = *(const f32x4 *)(xb + t * 128 + 4 * s4);
                f32x4 out;
#pragma unroll
                for (int k = 0; k < 4; ++k) out[k] = z[k] + rs[k] * xin[k];
                *(f32x4 *)(yb + t * 128 + 4 * s4) = out;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16x3, large dilations: the same kernel with the input staged in TWO HALVES of 64 channels (round 3).  The 8-phase tiles the
// d >= 4096 blocks need (few steps per phase: 32 / 16 at L = 131072) have 240 rows; two whole (hi, lo) tiles are 120 KB of LDS = one
// workgroup per CU (measured 6.4 ms per launch against 4.76 ms for the blocks that run two per CU).  Staged as [240 rows][64 channels]
// hi + lo = 60 KB the kernel keeps two workgroups per CU; the price is one more barrier pair per tile and a reduction that runs
// half-major (channels 0..63 of all taps, then 64..127): same terms, another fp32 summation order.
// LDS image: 128-byte rows (8 slots of 16 B), slot ^ ((row >> 1) & 7): conflict free for ds_read_b128's lane groups at the start rows
// that occur (multiples of 8).
// ------------------------------------------------------------------------------------------------
// The class-major loop for the eight-phase half-tile kernel below (128-byte rows of 64 channels, two k-steps kl per staged half): taps of one
// parity share their B fragments (tap j + 2, column tile q - 1 = the rows of tap j, column tile q); a pseudo-class is two taps j0, j0 + 2 (the
// last one: tap 13 alone), its A fragments (2 taps x 2 row tiles x hi / lo) double-buffered over the two k-steps, B fragment pairs in a ring
// of two windows.  The all-padding (column tile, tap) pairs are skipped under a wave-uniform branch like in the tap-major loop (live[u]).
template <int NU, bool LASTC, int NC>
__device__ __forceinline__ void tcn_reuse_class_x3_half(f32x4 (&acc)[2][NC], bf16x8 (&H0)[2][2], bf16x8 (&L0)[2][2], bf16x8 (&H1)[2][2],
                                                        bf16x8 (&L1)[2][2], bf16x8 (&rh)[2], bf16x8 (&rl)[2], const unsigned char *sm_hi,
                                                        const unsigned char *sm_lo, const unsigned char *wbase, size_t lo_img, unsigned aoff, int half,
                                                        int j0, int j0n, unsigned live0, unsigned live1, int l16, int g) {
    constexpr int P = 8, NW = NC - 1 + NU;
    static_assert((2 * NW) % 2 == 0, "the ring is two windows deep");
    const int rowb = (P * j0 + l16) * 128, rowbn = (P * j0n + l16) * 128;
    const int swz = (4 * (j0 & 1) + (l16 >> 1)) & 7, swzn = (4 * (j0n & 1) + (l16 >> 1)) & 7;
#pragma unroll
    for (int kl = 0; kl < 2; ++kl) {
        bf16x8 (&ch)[2][2] = kl ? H1 : H0;
        bf16x8 (&cl)[2][2] = kl ? L1 : L0;
        bf16x8 (&nh)[2][2] = kl ? H0 : H1;
        bf16x8 (&nl)[2][2] = kl ? L0 : L1;
        if (!(LASTC && kl == 1)) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (kl == 0 && u >= NU) continue;
                int j = (kl == 0 ? j0 : j0n) + 2 * u;
                j = j < 15 ? j : 14;                               // the last pseudo-class has one tap: that slot holds a fragment nobody uses
                const size_t so = (size_t)((j * 4 + 2 * half + (kl == 0 ? 1 : 0)) * 2) * 4096 + aoff;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    nh[u][m] = *(const bf16x8 *)(wbase + so + (size_t)m * 4096);
                    nl[u][m] = *(const bf16x8 *)(wbase + lo_img + so + (size_t)m * 4096);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int n = kl * NW + i;
            const bf16x8 bh = rh[n & 1], bl = rl[n & 1];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int q = i - u;
                if (q >= 0 && q < NC) {
                    if (((u ? live1 : live0) >> q) & 1u) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cl[u][m], bh, acc[m][q], 0, 0, 0);
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch[u][m], bl, acc[m][q], 0, 0, 0);
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch[u][m], bh, acc[m][q], 0, 0, 0);
                    }
                }
            }
            const int n2 = n + 2, k2 = n2 / NW, i2 = n2 % NW;
            if (k2 < 2) {
                const int ofs = rowb + (((4 * k2 + g) ^ swz) << 4) + i2 * 2048;
                rh[n & 1] = *(const bf16x8 *)(sm_hi + ofs);
                rl[n & 1] = *(const bf16x8 *)(sm_lo + ofs);
            } else if (!LASTC) {
                const int ofs = rowbn + ((g ^ swzn) << 4) + i2 * 2048;
                rh[n & 1] = *(const bf16x8 *)(sm_hi + ofs);
                rl[n & 1] = *(const bf16x8 *)(sm_lo + ofs);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int P, int NQ, bool CM = false>
__global__ __launch_bounds__(256, 2) void tcn_block_bf16x3_half_kernel(TcnBlockArgs a) {
    static_assert(!CM || (P == 8 && NQ == 4), "the class-major loop is written for eight phases x 16 steps");
    constexpr int T = 32 * NQ, R = T + 14 * P, MT = T / P, NC = 2 * NQ;
    static_assert(P % 8 == 0, "the swizzle is conflict free for start rows that are multiples of 8");
    __shared__ __attribute__((aligned(16))) unsigned char smem[(2 * R * 128 > T * 512) ? 2 * R * 128 : T * 512];      // [hi | lo] half tiles; later the fp32 output tile
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned char *const sm_hi = smem, *const sm_lo = smem + R * 128;
    const int l16 = lane & 15, g = lane >> 4;

    int tile = a.xcd_tiles > 0 ? (int)(blockIdx.x & 7) * a.xcd_tiles + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int mg = tile % a.tiles_step;
    tile /= a.tiles_step;
    const int pg = tile % a.tiles_phase;
    const int b = tile / a.tiles_phase;
    const int m0 = mg * MT, phi0 = pg * P;
    const float *xb = (const float *)a.x + (size_t)b * a.Lp * 128;
    float *yb = (float *)a.y + (size_t)b * a.Lp * 128;

    f32x4 acc[2][NC];
#pragma unroll
    for (int m = 0; m < 2; ++m) {          // accumulators start from the BN shift of their channel
        const f32x4 sh = *(const f32x4 *)(a.shift + 32 * w + 16 * m + 4 * g);
#pragma unroll
        for (int q = 0; q < NC; ++q) acc[m][q] = sh;
    }
    const unsigned char *wbase = (const unsigned char *)a.wpk;
    const unsigned aoff = (unsigned)(w * 64 + lane) * 16u;
    constexpr size_t LO_IMG = (size_t)120 * 4096;
    constexpr int RB = 8;
    static_assert(NC == RB, "one ring turn per k-step");

#pragma unroll 1
    for (int c = 0; c < 2; ++c) {
        if (c) __syncthreads();            // every wave is done reading the first half
        // ---- stage channels 64 c .. 64 c + 63 of the R rows: a thread owns one 16-byte slot (8 channels) of rows prow, prow + 32, ...
        {
            const int slot = tid & 7, prow = tid >> 3;
            constexpr int NPASS = (R + 31) / 32;
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int row = prow + 32 * i;
                const long t = (long)(m0 + row / P - 7) * a.d + phi0 + (row % P);
                const float *src = (row < R && t >= 0 && t < a.L) ? xb + t * 128 + 64 * c + slot * 8 : (const float *)a.zeros + slot * 8;
                const f32x4 v0 = *(const f32x4 *)src, v1 = *(const f32x4 *)(src + 4);          // (never a predicated load: zero page)
                if (row < R) {
                    bf16x8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[e] = (__bf16)v0[e];
                        lo[e] = (__bf16)(v0[e] - (float)hi[e]);
                        hi[4 + e] = (__bf16)v1[e];
                        lo[4 + e] = (__bf16)(v1[e] - (float)hi[4 + e]);
                    }
                    const int off = row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
                    *(bf16x8 *)(sm_hi + off) = hi;
                    *(bf16x8 *)(sm_lo + off) = lo;
                }
            }
        }
        __syncthreads();

        if constexpr (CM) {
            // class-major (mst_tcn_set_tuning bit 6): eight pseudo-classes of two taps (0, 2), (4, 6), (8, 10), (12, 14), (1, 3), (5, 7), (9, 11), (13)
            const int nsteps_cm = (int)(((long)a.L + a.d - 1) / a.d);
            auto live_of = [&](int j) {
                unsigned live = 0;
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const int s_lo = m0 + (16 * q) / P + j - 7, s_hi = m0 + (16 * q + 15) / P + j - 7;
                    if (!(s_hi < 0 || s_lo >= nsteps_cm)) live |= 1u << q;
                }
                return live;
            };
            auto j0_of = [](int pc) { return 4 * (pc & 3) + (pc >> 2); };
            bf16x8 H0[2][2], L0[2][2], H1[2][2], L1[2][2], rh[2], rl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    H0[u][m] = *(const bf16x8 *)(wbase + (size_t)(((2 * u) * 4 + 2 * c) * 2 + m) * 4096 + aoff);
                    L0[u][m] = *(const bf16x8 *)(wbase + LO_IMG + (size_t)(((2 * u) * 4 + 2 * c) * 2 + m) * 4096 + aoff);
                }
            {
                const int o0 = l16 * 128 + ((g ^ ((l16 >> 1) & 7)) << 4);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    rh[i] = *(const bf16x8 *)(sm_hi + o0 + i * 2048);
                    rl[i] = *(const bf16x8 *)(sm_lo + o0 + i * 2048);
                }
            }
#pragma unroll 1
            for (int pc = 0; pc < 7; ++pc) {
                const int j0 = j0_of(pc), j0n = j0_of(pc + 1);
                tcn_reuse_class_x3_half<2, false, NC>(acc, H0, L0, H1, L1, rh, rl, sm_hi, sm_lo, wbase, LO_IMG, aoff, c, j0, j0n, live_of(j0),
                                                      live_of(j0 + 2), l16, g);
            }
            tcn_reuse_class_x3_half<1, true, NC>(acc, H0, L0, H1, L1, rh, rl, sm_hi, sm_lo, wbase, LO_IMG, aoff, c, 13, 13, live_of(13), 0u, l16, g);
            continue;
        }
        // A fragments: wpk[part][ks = j*4 + kk][row tile m][wave][lane], kk = 2 c + kl; one k-step of (hi, lo) fragments in flight ahead
        bf16x8 ah[2][2], al[2][2], bh[RB], bl[RB];
#pragma unroll
        for (int kl = 0; kl < 2; ++kl)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ah[kl][m] = *(const bf16x8 *)(wbase + (size_t)((2 * c + kl) * 2 + m) * 4096 + aoff);
                al[kl][m] = *(const bf16x8 *)(wbase + LO_IMG + (size_t)((2 * c + kl) * 2 + m) * 4096 + aoff);
            }
        {
            const int o0 = l16 * 128 + ((g ^ ((l16 >> 1) & 7)) << 4);
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                bh[q] = *(const bf16x8 *)(sm_hi + o0 + q * 2048);
                bl[q] = *(const bf16x8 *)(sm_lo + o0 + q * 2048);
            }
        }
        // a (column tile q, tap j) pair whose 16 input rows all lie outside the segment (zero padding) contributes nothing: skipped,
        // wave-uniform (20 % of the MFMAs of the d = 8192 block at L = 131072, 10 % of the d = 4096 block).  (An unrolled form with the dead
        // pairs removed at compile time - what the bf16 kernel does for tiles that span their whole phase sequence - spills 134 registers here:
        // hipcc hoists the unrolled loop's ~30 lane-dependent LDS addresses out of the half loop.  Not used.)
        // a (column tile q, tap j) pair whose 16 input rows all lie outside the segment (zero padding) contributes nothing: skipped,
        // wave-uniform (20 % of the MFMAs of the d = 8192 block at L = 131072, 10 % of the d = 4096 block)
        const int nsteps = (int)(((long)a.L + a.d - 1) / a.d);
        for (int j = 0; j < 15; ++j) {
            const int jn = j < 14 ? j + 1 : 14;
            const int rb0 = j * P + l16, rb1 = jn * P + l16;
            unsigned live = 0;
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int s_lo = m0 + (16 * q) / P + j - 7, s_hi = m0 + (16 * q + 15) / P + j - 7;
                if (!(s_hi < 0 || s_lo >= nsteps)) live |= 1u << q;
            }
#pragma unroll
            for (int kl = 0; kl < 2; ++kl) {
                const int rbn = kl ? rb1 : rb0;
                const int kn = kl ^ 1;
                const int on = rbn * 128 + (((4 * kn + g) ^ ((rbn >> 1) & 7)) << 4);
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    if ((live >> q) & 1u) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[kl][m], bh[q], acc[m][q], 0, 0, 0);
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[kl][m], bl[q], acc[m][q], 0, 0, 0);
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[kl][m], bh[q], acc[m][q], 0, 0, 0);
                    }
                    bh[q] = *(const bf16x8 *)(sm_hi + on + q * 2048);
                    bl[q] = *(const bf16x8 *)(sm_lo + on + q * 2048);
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                const int ksn = jn * 4 + 2 * c + kl;          // the same k-step of the next tap (behind the last tap: loaded again, unused)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    ah[kl][m] = *(const bf16x8 *)(wbase + (size_t)(ksn * 2 + m) * 4096 + aoff);
                    al[kl][m] = *(const bf16x8 *)(wbase + LO_IMG + (size_t)(ksn * 2 + m) * 4096 + aoff);
                }
            }
        }
    }

    // ---- exact fp32 epilogue: LeakyReLU -> FiLM in the accumulator layout, transposed through LDS (the input tiles are dead; fp32 rows of
    //      512 B, 16-byte slots XOR-swizzled by the row), then whole rows: + res * x_in (x_in re-read from global memory, L2-hot) -> store.
    //      Straight from the accumulator layout a lane's 16 bytes are a 64-byte run of its row (16 rows per instruction, for the residual
    //      load as for the store): measured on the bf16 twin of this access pattern at ~0.3 ms per launch (profiles/r03_tcn_block_forms_summary.md)
    const float *frow = a.film + (a.film_rows > 1 ? (size_t)b * 256 : 0);
    __syncthreads();                       // every wave is done reading the input tiles
    float *st = (float *)smem;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int co0 = 32 * w + 16 * m + 4 * g;
        const f32x4 fr = *(const f32x4 *)(frow + co0);
        const f32x4 fb = *(const f32x4 *)(frow + 128 + co0);
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int o = 16 * q + l16;
            f32x4 z;
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = fr[i] * leaky_relu(acc[m][q][i]) + fb[i];
            *(f32x4 *)(st + o * 128 + (((co0 >> 2) ^ (o & 31)) << 2)) = z;
        }
    }
    __syncthreads();
    {
        const int s4 = tid & 31;                                   // this thread's 4 channels, the same in every pass
        const f32x4 rs = *(const f32x4 *)(a.res + 4 * s4);
        if (a.y_out) {
            // last block: the 1x1 output conv + bias + clamp(-1, 1) (reference architectures.py:133,145) on the finished rows - a row's 128
            // channels sit in 32 lanes (half a wave): 4 products per lane, a 5-step butterfly; the activation itself is not stored
            // (the separate head kernel read 2.1 GB for it at 32 x 131072)
            const f32x4 ow0 = *(const f32x4 *)(a.out_w + 4 * s4);
            const f32x4 ow1 = a.nout > 1 ? *(const f32x4 *)(a.out_w + 128 + 4 * s4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < T / 8; ++i) {
                const int o = (tid >> 5) + 8 * i;
                const long t = (long)(m0 + o / P) * a.d + phi0 + (o % P);
                float h0 = 0.0f, h1 = 0.0f;
                if (t < a.L) {
                    const f32x4 z = *(const f32x4 *)(st + o * 128 + ((s4 ^ (o & 31)) << 2));
                    const f32x4 xin = *(const f32x4 *)(xb + t * 128 + 4 * s4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float out = z[k] + rs[k] * xin[k];
                        h0 = fmaf(ow0[k], out, h0);
                        h1 = fmaf(ow1[k], out, h1);
                    }
                }
#pragma unroll
                for (int mm = 16; mm >= 1; mm >>= 1) {
                    h0 += __shfl_xor(h0, mm);
                    h1 += __shfl_xor(h1, mm);
                }
                if (s4 < a.nout && t < a.L)
                    a.y_out[((size_t)b * a.nout + s4) * a.L + t] = fminf(1.0f, fmaxf(-1.0f, (s4 ? h1 : h0) + a.out_b[s4]));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < T / 8; ++i) {
            const int o = (tid >> 5) + 8 * i;
            const long t = (long)(m0 + o / P) * a.d + phi0 + (o % P);
            if (t < a.L) {
                const f32x4 z = *(const f32x4 *)(st + o * 128 + ((s4 ^ (o & 31)) << 2));
                const f32x4 xin = *(const f32x4 *)(xb + t * 128 + 4 * s4);
                f32x4 out;
#pragma unroll
                for (int k = 0; k < 4; ++k) out[k] = z[k] + rs[k] * xin[k];
                *(f32x4 *)(yb + t * 128 + 4 * s4) = out;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// blocks 1..n-1, exact fp32 (v_mfma_f32_32x32x2_f32 == k-ordered fmaf chain), fp32 activations.
// The parity mode.  Input channels are staged in 4 chunks of 32 (128 B per row) to keep LDS small.
// ------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256, 2) void tcn_block_f32_kernel(TcnBlockArgs a) {
    constexpr int T = 256, R = T + 14 * P, MT = T / P;
    __shared__ __attribute__((aligned(16))) float smem[R * 32];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ln = lane & 31, h = lane >> 5;

    int tile = a.xcd_tiles > 0 ? (int)(blockIdx.x & 7) * a.xcd_tiles + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int mg = tile % a.tiles_step;
    tile /= a.tiles_step;
    const int pg = tile % a.tiles_phase;
    const int b = tile / a.tiles_phase;
    const int m0 = mg * MT, phi0 = pg * P;
    const float *xb = (const float *)a.x + (size_t)b * a.Lp * 128;
    float *yb = (float *)a.y + (size_t)b * a.Lp * 128;

    f32x16 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.0f;

    // A fragments: wpk[((j*4 + c)*4 + ksg)*256 + w*64 + lane] = 4 floats, element i = W'[32w+ln][32c + 2(4ksg+i) + h][j]
    const f32x4 *wp = (const f32x4 *)a.wpk + (w * 64 + lane);

    for (int c = 0; c < 4; ++c) {
        if (c) __syncthreads();   // every wave is done reading the previous chunk
        {
            // 8 lanes x 16 B per row; dword-granular XOR swizzle (dword i of row r lives at i ^ (r & 31)) so the
            // 32 rows read by one half-wave ds_read_b32 land in 32 distinct banks
            const int s = tid & 7;
#pragma unroll 2
            for (int r = tid >> 3; r < R; r += 32) {
                const long t = (long)(m0 + r / P - 7) * a.d + phi0 + (r % P);
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};          // (a zero-page form of this load measured 1.7 % slower per launch, round 3)
                if (t >= 0 && t < a.L) v = *(const f32x4 *)(xb + t * 128 + c * 32 + s * 4);
                const int rs = r & 31;
                f32x4 o1, o2;   // o2[i ^ (rs & 3)] = v[i] as two conditional swaps (no runtime vector indexing)
                o1[0] = (rs & 1) ? v[1] : v[0];
                o1[1] = (rs & 1) ? v[0] : v[1];
                o1[2] = (rs & 1) ? v[3] : v[2];
                o1[3] = (rs & 1) ? v[2] : v[3];
                o2[0] = (rs & 2) ? o1[2] : o1[0];
                o2[1] = (rs & 2) ? o1[3] : o1[1];
                o2[2] = (rs & 2) ? o1[0] : o1[2];
                o2[3] = (rs & 2) ? o1[1] : o1[3];
                *(f32x4 *)(smem + r * 32 + ((s ^ (rs >> 2)) << 2)) = o2;
            }
        }
        __syncthreads();

        for (int j = 0; j < 15; ++j) {
            const int rowbase = j * P + ln;
            const int sw = rowbase & 31;
            const float *rp = smem + rowbase * 32;
            f32x4 af[4];
#pragma unroll
            for (int ksg = 0; ksg < 4; ++ksg) af[ksg] = wp[((j * 4 + c) * 4 + ksg) * 256];
#pragma unroll
            for (int ksg = 0; ksg < 4; ++ksg) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int off = (2 * (4 * ksg + i) + h) ^ sw;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float bv = rp[q * 1024 + off];
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ksg][i], bv, acc[q], 0, 0, 0);
                    }
                }
            }
        }
    }

    const float *frow = a.film + (a.film_rows > 1 ? (size_t)b * 256 : 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int co0 = 32 * w + 8 * g + 4 * h;
        const f32x4 sh = *(const f32x4 *)(a.shift + co0);
        const f32x4 fr = *(const f32x4 *)(frow + co0);
        const f32x4 fb = *(const f32x4 *)(frow + 128 + co0);
        const f32x4 rs = *(const f32x4 *)(a.res + co0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int o = 32 * q + ln;
            const long t = (long)(m0 + o / P) * a.d + phi0 + (o % P);
            if (t < a.L) {
                const f32x4 xin = *(const f32x4 *)(xb + t * 128 + co0);
                f32x4 out;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = leaky_relu(acc[q][4 * g + i] + sh[i]);
                    v = fr[i] * v + fb[i];
                    out[i] = v + rs[i] * xin[i];
                }
                *(f32x4 *)(yb + t * 128 + co0) = out;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// block 0: 2 -> 128 channels, k = 15, dilation 1, input fp32 NCL [B][2][L] (the caller's tensor),
// output NLC.  30 MAC per output element: fp32 VALU (exact), register-tiled 8 channels x 4 time steps.
// res is the grouped 1x1 with groups = 2: out-channel co reads in-channel co / 64.
// ------------------------------------------------------------------------------------------------
struct TcnBlock0Args {
    const float *x;       // [B][2][L]
    void *y;              // [B][Lp][128]
    const float *w;       // [2][15][128] BN-folded
    const float *shift, *film, *res;
    int film_rows, B, L, Lp;
    const void *wpk16;    // bf16 MFMA form: A fragments [k-step s = 0, 1][wave][lane][8], k = ci * 15 + j (30 of 32 used)
};

// ------------------------------------------------------------------------------------------------
// block 0 on the matrix cores (bf16 mode).  The scalar kernel below runs at the fp32 FMA peak of the vector ALUs (32 GFLOP
// in 0.45 ms = 71 of 78.6 TFLOP/s; v_pk_fma_f32 issues at half rate and does not help), so the 2 x 15 taps become one
// K = 32 contraction: D[co][t] = sum_k W'[co][k] X[k][t], X[ci * 15 + j][t] = x[ci][t + j - 7].  The waveform is split
// x = hi + lo into two bf16 operands (16 significant bits), the folded weights are bf16 like every other block's.
// A workgroup owns 256 output times: wave w builds the B fragments of column tiles 2w, 2w+1 once for all four waves
// (through LDS), then every wave runs 32 MFMAs for its 32 channels and the usual fused epilogue; HBM-bound on the write.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void tcn_block0_mfma_kernel(TcnBlock0Args a) {
    constexpr int T = 256, XW = T + 14;
    __shared__ float xs[2][XW + 2];
    __shared__ __attribute__((aligned(16))) unsigned char buf[T * 256];       // B fragments (32 KB), later the transposed output tile
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ln = lane & 31, h = lane >> 5;
    const int tiles_t = (a.L + T - 1) / T;
    const int b = blockIdx.x / tiles_t;
    const int t0 = (blockIdx.x % tiles_t) * T;
    for (int i = tid; i < 2 * XW; i += 256) {
        const int ci = i / XW, k = i % XW;
        const long t = (long)t0 - 7 + k;
        xs[ci][k] = (t >= 0 && t < a.L) ? a.x[((size_t)b * 2 + ci) * a.L + t] : 0.0f;
    }
    const bf16x8 *wp = (const bf16x8 *)a.wpk16 + (w * 64 + lane);
    const bf16x8 af0 = wp[0], af1 = wp[256];
    __syncthreads();
    // B fragments: fragment (q, s, part) at buf + (((q * 2 + s) * 2 + part) * 64 + lane) * 16, part 0 = hi, 1 = lo
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
        const int q = 2 * w + qq;
#pragma unroll
        for (int sI = 0; sI < 2; ++sI) {
            bf16x8 hi, lo;
            tcn_block0_bfrag(xs[0], xs[1], 32 * q + ln, sI, h, hi, lo);
            *(bf16x8 *)(buf + (((q * 2 + sI) * 2 + 0) * 64 + lane) * 16) = hi;
            *(bf16x8 *)(buf + (((q * 2 + sI) * 2 + 1) * 64 + lane) * 16) = lo;
        }
    }
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 sh = *(const f32x4 *)(a.shift + 32 * w + 8 * g + 4 * h);
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[q][4 * g + i] = sh[i];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int sI = 0; sI < 2; ++sI) {
            const bf16x8 bh = *(const bf16x8 *)(buf + (((q * 2 + sI) * 2 + 0) * 64 + lane) * 16);
            const bf16x8 bl = *(const bf16x8 *)(buf + (((q * 2 + sI) * 2 + 1) * 64 + lane) * 16);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sI ? af1 : af0, bh, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sI ? af1 : af0, bl, acc[q], 0, 0, 0);
        }
    }
    __syncthreads();                                   // every wave has read its fragments: the buffer becomes the output tile
    const float *frow = a.film + (a.film_rows > 1 ? (size_t)b * 256 : 0);
    const int cin = w >> 1;                            // grouped residual: channels 0..63 read input 0, 64..127 input 1
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int co0 = 32 * w + 8 * g + 4 * h;
        const f32x4 fr = *(const f32x4 *)(frow + co0);
        const f32x4 fb = *(const f32x4 *)(frow + 128 + co0);
        const f32x4 rs = *(const f32x4 *)(a.res + co0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int o = 32 * q + ln;
            const float xres = xs[cin][o + 7];
            bf16x4 out;
#pragma unroll
            for (int i = 0; i < 4; ++i) out[i] = tcn_block0_out(acc[q][4 * g + i], fr[i], fb[i], rs[i], xres);
            *(bf16x4 *)(buf + o * 256 + (((co0 >> 3) ^ (o & 15)) << 4) + 8 * h) = out;
        }
    }
    __syncthreads();
    __bf16 *yb = (__bf16 *)a.y + (size_t)b * a.Lp * 128;
    const int slot = tid & 15;
#pragma unroll
    for (int i = 0; i < T / 16; ++i) {
        const int o = (tid >> 4) + 16 * i;
        const long t = (long)t0 + o;
        if (t < a.L) *(bf16x8 *)(yb + t * 128 + slot * 8) = *(const bf16x8 *)(buf + o * 256 + ((slot ^ (o & 15)) << 4));
    }
}

template <typename OutT>
__global__ __launch_bounds__(256) void tcn_block0_kernel(TcnBlock0Args a) {
    // exact-fp32 form (the parity mode; bf16 mode runs tcn_block0_mfma_kernel).  A workgroup stages the 15 KB of folded weights
    // once and walks NSUB consecutive 64-step tiles with them.  VALU-bound: 960 FMAs per thread and tile, 71 of 78.6 TFLOP/s.
    constexpr int TT = 64, XW = TT + 14, NSUB = 8;
    __shared__ __attribute__((aligned(16))) float ws[2 * 15 * 128];
    __shared__ float xs[2][2 * XW];
    const int tid = threadIdx.x;
    const int tiles_t = (a.L + TT * NSUB - 1) / (TT * NSUB);
    const int b = blockIdx.x / tiles_t;
    const int tbase = (blockIdx.x % tiles_t) * TT * NSUB;
    for (int i = tid; i < 2 * 15 * 128 / 4; i += 256) ((f32x4 *)ws)[i] = ((const f32x4 *)a.w)[i];
    auto stage_x = [&](int buf, int t0) {
        if (tid < 2 * XW) {
            const int ci = tid / XW, k = tid % XW;
            const long t = (long)t0 - 7 + k;
            xs[buf][tid] = (t >= 0 && t < a.L) ? a.x[((size_t)b * 2 + ci) * a.L + t] : 0.0f;
        }
    };
    stage_x(0, tbase);
    const int cg = tid & 15, tg = tid >> 4;
    const int co0 = cg * 8;
    const float *frow = a.film + (a.film_rows > 1 ? (size_t)b * 256 : 0);
    float sh[8], fr[8], fb[8], rs[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        sh[c] = a.shift[co0 + c];
        fr[c] = frow[co0 + c];
        fb[c] = frow[128 + co0 + c];
        rs[c] = a.res[co0 + c];
    }
    const int cin = co0 >> 6;
    OutT *yb = (OutT *)a.y + (size_t)b * a.Lp * 128;
    for (int sub = 0; sub < NSUB; ++sub) {
        const int t0 = tbase + sub * TT;
        if (t0 >= a.L) break;                              // uniform
        __syncthreads();                                   // this tile's samples (and, first time, the weights) are in LDS
        if (sub + 1 < NSUB) stage_x((sub + 1) & 1, t0 + TT);     // the next tile's samples travel during the arithmetic
        const float *xt = xs[sub & 1];
        float acc[4][8];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[tt][c] = 0.0f;
        for (int ci = 0; ci < 2; ++ci) {
            const float *xrow = xt + ci * XW + 4 * tg;
#pragma unroll 3
            for (int j = 0; j < 15; ++j) {
                const f32x4 w0 = *(const f32x4 *)(ws + (ci * 15 + j) * 128 + co0);
                const f32x4 w1 = *(const f32x4 *)(ws + (ci * 15 + j) * 128 + co0 + 4);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const float xv = xrow[tt + j];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[tt][c] = fmaf(w0[c], xv, acc[tt][c]);
                        acc[tt][c + 4] = fmaf(w1[c], xv, acc[tt][c + 4]);
                    }
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int t = t0 + 4 * tg + tt;
            if (t < a.L) {
                const float xin = xt[cin * XW + 4 * tg + tt + 7];
                float o8[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float v = leaky_relu(acc[tt][c] + sh[c]);
                    v = fr[c] * v + fb[c];
                    o8[c] = v + rs[c] * xin;
                }
                store8(yb + (size_t)t * 128 + co0, o8);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// output 1x1 conv (128 -> noutputs <= 2) + bias + clamp(-1, 1)  (architectures.py:133,145); NLC in,
// fp32 NCL [B][nout][L] out (the caller's tensor).  HBM-bound: one read of the last activation.
// ------------------------------------------------------------------------------------------------
struct TcnOutArgs {
    const void *x;     // [B][Lp][128]
    float *y;          // [B][nout][L]
    const float *w;    // [nout][128]
    const float *bias; // [nout]
    int nout, B, L, Lp;
};

template <typename InT>
__global__ __launch_bounds__(256) void tcn_output_kernel(TcnOutArgs a) {
    constexpr int TT = 64;
    __shared__ float outs[2][TT];
    const int tid = threadIdx.x, sl = tid & 15, rr = tid >> 4;
    const int tiles_t = (a.L + TT - 1) / TT;
    const int b = blockIdx.x / tiles_t;
    const int t0 = (blockIdx.x % tiles_t) * TT;
    float w0[8], w1[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        w0[c] = a.w[8 * sl + c];
        w1[c] = a.nout > 1 ? a.w[128 + 8 * sl + c] : 0.0f;
    }
    const InT *xb = (const InT *)a.x + (size_t)b * a.Lp * 128;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int i = pass * 16 + rr;
        const int t = t0 + i;
        float s0 = 0.0f, s1 = 0.0f;
        if (t < a.L) {
            float v8[8];
            load8(xb + (size_t)t * 128 + 8 * sl, v8);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                s0 = fmaf(w0[c], v8[c], s0);
                s1 = fmaf(w1[c], v8[c], s1);
            }
        }
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
            s0 += __shfl_xor(s0, m);
            s1 += __shfl_xor(s1, m);
        }
        if (sl == 0) {
            outs[0][i] = s0;
            outs[1][i] = s1;
        }
    }
    __syncthreads();
    if (tid < a.nout * TT) {
        const int o = tid / TT, i = tid % TT;
        const int t = t0 + i;
        if (t < a.L) {
            float v = outs[o][i] + a.bias[o];
            v = fminf(1.0f, fmaxf(-1.0f, v));
            a.y[((size_t)b * a.nout + o) * a.L + t] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// FiLM factor table: film[n][row][0..2C) = film_fc_n(cond[row])   (network_utils.py:180-181).
// One wave per output feature; 14 GEMVs of 256 x 2048 - once per embedding, not per segment.
// ------------------------------------------------------------------------------------------------
struct FilmArgs {
    const float *fw;     // [nblocks][2C][D]
    const float *fb;     // [nblocks][2C]
    const float *cond;   // [rows][D] (+ n*block_stride)
    float *film;         // [nblocks][rows][2C]
    int nblocks, two_c, D, rows;
    long block_stride;
};

__global__ __launch_bounds__(256) void tcn_film_kernel(FilmArgs a) {
    const int lane = threadIdx.x & 63;
    const int out = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (out >= a.nblocks * a.two_c) return;   // wave-uniform
    const int n = out / a.two_c, oc = out % a.two_c;
    const float *wrow = a.fw + ((size_t)n * a.two_c + oc) * a.D;
    for (int row = 0; row < a.rows; ++row) {
        const float *cv = a.cond + (size_t)n * a.block_stride + (size_t)row * a.D;
        float s = 0.0f;
        for (int i = lane; i < a.D; i += 64) s = fmaf(wrow[i], cv[i], s);
        s = wave_sum(s);
        if (lane == 0) a.film[((size_t)n * a.rows + row) * a.two_c + oc] = s + a.fb[n * a.two_c + oc];
    }
}

// FiLM.forward on its own (network_utils.py:181-182): y = r * x + b over NCL x [B][C][L], film[row][0..C) = r, [C..2C) = b
__global__ __launch_bounds__(256) void film_apply_kernel(const float *x, float *y, const float *film, int rows, int C, long L, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long bc = i / L;
    const int c = (int)(bc % C);
    const float *f = film + (rows > 1 ? (size_t)(bc / C) * 2 * C : 0);
    y[i] = f[c] * x[i] + f[C + c];
}

// NLC -> NCL fp32 copy of an intermediate activation (parity probe only, not on the hot path)
template <typename InT>
__global__ void tcn_unpack_kernel(const void *x, float *y, int B, int L, int Lp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * L * 128) return;
    const int c = i % 128;
    const size_t bt = i / 128;
    const int t = bt % L;
    const int b = bt / L;
    y[((size_t)b * 128 + c) * L + t] = (float)((const InT *)x)[((size_t)b * Lp + t) * 128 + c];
}

// ------------------------------------------------------------------------------------------------
// Calibration of the box (bench.py "roofline.calib_ms"): the BARE MAIN LOOP of tcn_block_bf16_duo_kernel - v_mfma_f32_16x16x32_bf16 with the
// product loop's operand traffic (class-major since round 4's last third: 128 weight fragments from L2 and 304 B fragments from LDS per
// 1920 MFMAs; before: 120 and 960 - the tap-major loop; same box 1.230 -> 1.145 ms, profiles/r04_micro_mainloop_reuse.txt, so calib_ms
// values of earlier records are 7 % higher for the same box), no staging, no epilogue, no store - on synthetic operands with realistic statistics (activations ~ N(0, 0.5^2), weights ~ N(0, 0.05^2): the chip's power limit
// depends on the operand bits).  512 workgroups x `rep` tiles of 256 times: rep = 32 is exactly the arithmetic of one dense TCN block
// launch at 32 x 131072 (2.06 TFLOP), so its duration is what this box lets the block kernel's main loop run at - boxes of the pool
// differ by +-4 %, the block kernel divided by THIS is comparable across boxes.  Thread 0 of workgroup 0 also reports the shader clock
// it ran at (s_memtime counts shader clocks, s_memrealtime 100 MHz).  tools/micro/tcn_mainloop_variants.hip::k_reuse16.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned calib_normalish_bf16(unsigned z, float sigma) {      // sum of four uniform bytes: ~ N(0, sigma^2), as bf16 bits
    z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16;
    const float u = ((z & 0xff) + ((z >> 8) & 0xff) + ((z >> 16) & 0xff) + (z >> 24)) / 255.0f - 2.0f;
    return (unsigned)(__builtin_bit_cast(unsigned, u * sigma * 1.74f) >> 16);
}
__global__ void tcn_calib_fill_kernel(unsigned *w, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) w[i] = calib_normalish_bf16((unsigned)i * 2654435761u + 777u, 0.05f) | (calib_normalish_bf16((unsigned)i * 747796405u + 2891336453u, 0.05f) << 16);
}
__global__ __launch_bounds__(256, 2) void tcn_calib_mainloop_kernel(const void *wpk, float *out, long long *clocks, int rep) {
    constexpr int P = 4, T = 256, R = T + 14 * P;
    __shared__ __attribute__((aligned(16))) unsigned char smem[R * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l16 = lane & 15, g = lane >> 4;
    long long c0 = 0, r0 = 0;
    if (blockIdx.x == 0 && tid == 0) {
        c0 = mst_clock();
        r0 = mst_realtime();
    }
    for (int i = tid; i < R * 64; i += 256) {
        const unsigned r = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        ((unsigned *)smem)[i] = calib_normalish_bf16(r, 0.5f) | (calib_normalish_bf16(r * 747796405u + 2891336453u, 0.5f) << 16);
    }
    __syncthreads();
    f32x4 acc[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[m][q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // the class-major loop of the duo kernel's four-phase tiles (tcn_reuse_class), tile after tile on the same LDS image
    const MstStream16 wst = mst_stream16(wpk, 60u * 2u * 4096u);
    const unsigned aoff = (unsigned)(w * 64 + lane) * 16u;
    bf16x8 A0[4][2], A1[4][2], ring[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int m = 0; m < 2; ++m) A0[u][m] = __builtin_bit_cast(bf16x8, mst_stream_load16(wst, aoff + m * 4096, (unsigned)(16 * u) * 8192u));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) ring[i] = *(const bf16x8 *)(smem + l16 * 256 + ((g ^ l16) << 4) + i * 4096);
    for (int r = 0; r < rep; ++r) {
#pragma unroll 1
        for (int c = 0; c < 3; ++c) tcn_reuse_class<P, 4, false, 4>(acc, A0, A1, ring, smem, wst, aoff, c, c + 1, l16, g);
        tcn_reuse_class<P, 3, false, 4>(acc, A0, A1, ring, smem, wst, aoff, 3, 0, l16, g);
    }
    float s = 0.0f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc[m][q][0] + acc[m][q][1] + acc[m][q][2] + acc[m][q][3];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) {
        clocks[0] = mst_clock() - c0;
        clocks[1] = mst_realtime() - r0;
    }
}
