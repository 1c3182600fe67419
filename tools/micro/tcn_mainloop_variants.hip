// Micro-benchmark (round 2): what the three per-MFMA overheads of the TCN bf16 main loop cost on REALISTIC operands
// (activations ~ N(0, 0.5^2), weights ~ N(0, 0.05^2); round 1's table used random bit patterns), and whether the two wave-tile
// shapes that trade them against each other pay:
//   base      32 channels x 256 times per wave (the product kernel): per k-step 1 A fragment from L2, 8 B fragments from LDS, 8 MFMAs
//   noA       the same with the A fragments loaded once per tile (as if the weight stream were free)
//   noB       the same with the B fragments loaded once per tile (as if the LDS reads were free)
//   noAB      bare MFMAs on these operands
//   w64x128   64 channels x 128 times per wave: per k-step 2 A fragments, 4 B fragments (each used twice), 8 MFMAs
//   bare16x16 bare v_mfma_f32_16x16x32_bf16 on these operands (the same FLOPs per wave in twice the instructions)
//   base16x16 the product loop's operand traffic with that shape (2 A fragments + 16 B fragments per 32 MFMAs)
//   noA16x16 / noB16x16  the same without the weight stream / without the LDS reads
//   w64_16x16 64 channels x 128 times per wave in that shape (4 A fragments + 8 B fragments per 32 MFMAs)
// Two 256-thread workgroups per CU, REP tiles per workgroup, no staging, no epilogue.
//   hipcc --offload-arch=gfx950 -O3 -o tcn_mainloop_variants tcn_mainloop_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__device__ inline void fill_lds(unsigned char *smem, int bytes, int tid) {
    for (int i = tid; i < bytes / 4; i += 256) {
        unsigned r = (unsigned)i * 2654435761u + blockIdx.x * 40503u; r ^= r >> 15; r *= 2246822519u; r ^= r >> 13;
        auto g = [&](unsigned z) { z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16;
                                   const float u = ((z & 0xff) + ((z >> 8) & 0xff) + ((z >> 16) & 0xff) + (z >> 24)) / 255.0f - 2.0f;
                                   return (unsigned)(__builtin_bit_cast(unsigned, u * 0.87f) >> 16); };
        ((unsigned *)smem)[i] = g(r) | (g(r * 747796405u + 2891336453u) << 16);
    }
}

// MODE 0 base, 1 noA, 2 noB, 3 noAB
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_base(const bf16x8 *wpk, float *out, int rep) {
    constexpr int P = 4, NQ = 8, T = 256, R = T + 14 * P;
    __shared__ __attribute__((aligned(16))) unsigned char smem[R * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ln = lane & 31, h = lane >> 5;
    fill_lds(smem, R * 256, tid);
    __syncthreads();
    f32x16 acc[NQ];
    for (int q = 0; q < NQ; ++q)
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.0f;
    const bf16x8 *wp = wpk + (w * 64 + lane);
    for (int r = 0; r < rep; ++r) {
        bf16x8 af[8], bf[NQ];
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) af[kc] = wp[kc * 256];
        {
            const unsigned char *rp0 = smem + ln * 256 + ((h ^ (ln & 15)) << 4);
#pragma unroll
            for (int q = 0; q < NQ; ++q) bf[q] = *(const bf16x8 *)(rp0 + q * 8192);
        }
        for (int j = 0; j < 15; ++j) {
            const int jn = j < 14 ? j + 1 : 14;
            const int rb0 = j * P + ln, rb1 = jn * P + ln;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                const int rbn = (kc == 7) ? rb1 : rb0;
                const int kcn = (kc + 1) & 7;
                const unsigned char *np = smem + rbn * 256 + (((2 * kcn + h) ^ (rbn & 15)) << 4);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kc], bf[q], acc[q], 0, 0, 0);
                    if (MODE == 0 || MODE == 1) {
                        bf[q] = *(const bf16x8 *)(np + q * 8192);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                if (MODE == 0 || MODE == 2) af[kc] = wp[(jn * 8 + kc) * 256];
            }
        }
    }
    float s = 0.0f;
    for (int q = 0; q < NQ; ++q)
        for (int i = 0; i < 16; ++i) s += acc[q][i];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

// 64 channels x 128 times per wave: wave w = (channel half w & 1, time half w >> 1)
__global__ __launch_bounds__(256, 2) void k_w64(const bf16x8 *wpk, float *out, int rep) {
    constexpr int P = 4, T = 256, R = T + 14 * P;
    __shared__ __attribute__((aligned(16))) unsigned char smem[R * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ln = lane & 31, h = lane >> 5;
    const int ch = w & 1, th = w >> 1;
    fill_lds(smem, R * 256, tid);
    __syncthreads();
    f32x16 acc[2][4];
    for (int m = 0; m < 2; ++m)
        for (int q = 0; q < 4; ++q)
            for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.0f;
    const bf16x8 *wp = wpk + ((2 * ch) * 64 + lane);             // channel blocks 2 ch, 2 ch + 1 of every k-step
    const unsigned char *tbase = smem + th * 4 * 8192;            // this wave's four column tiles
    for (int r = 0; r < rep; ++r) {
        bf16x8 a0[8], a1[8], bf[4];
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) { a0[kc] = wp[kc * 256]; a1[kc] = wp[kc * 256 + 64]; }
        {
            const unsigned char *rp0 = tbase + ln * 256 + ((h ^ (ln & 15)) << 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) bf[q] = *(const bf16x8 *)(rp0 + q * 8192);
        }
        for (int j = 0; j < 15; ++j) {
            const int jn = j < 14 ? j + 1 : 14;
            const int rb0 = j * P + ln, rb1 = jn * P + ln;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                const int rbn = (kc == 7) ? rb1 : rb0;
                const int kcn = (kc + 1) & 7;
                const unsigned char *np = tbase + rbn * 256 + (((2 * kcn + h) ^ (rbn & 15)) << 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[kc], bf[q], acc[0][q], 0, 0, 0);
                    acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[kc], bf[q], acc[1][q], 0, 0, 0);
                    bf[q] = *(const bf16x8 *)(np + q * 8192);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                a0[kc] = wp[(jn * 8 + kc) * 256];
                a1[kc] = wp[(jn * 8 + kc) * 256 + 64];
            }
        }
    }
    float s = 0.0f;
    for (int m = 0; m < 2; ++m)
        for (int q = 0; q < 4; ++q)
            for (int i = 0; i < 16; ++i) s += acc[m][q][i];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

// bare MFMAs of the other bf16 shape, v_mfma_f32_16x16x32_bf16 (same FLOPs per wave: 1920 instructions of 16384 FLOP on 32 accumulator
// tiles of 16 x 16): does the shape change what the power limit lets through?
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
__global__ __launch_bounds__(256, 2) void k_bare16(const bf16x8 *wpk, float *out, int rep) {
    constexpr int P = 4, T = 256, R = T + 14 * P;
    __shared__ __attribute__((aligned(16))) unsigned char smem[R * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    fill_lds(smem, R * 256, tid);
    __syncthreads();
    f32x4 acc[2][16];
    for (int m = 0; m < 2; ++m)
        for (int q = 0; q < 16; ++q)
            for (int i = 0; i < 4; ++i) acc[m][q][i] = 0.0f;
    bf16x8 af[2][4], bf[16];
    for (int m = 0; m < 2; ++m)
        for (int k = 0; k < 4; ++k) af[m][k] = wpk[(k * 2 + m) * 256 + w * 64 + lane];
    for (int q = 0; q < 16; ++q) bf[q] = *(const bf16x8 *)(smem + (16 * q + (lane & 15)) * 256 + ((lane >> 4) << 4));
    for (int r = 0; r < rep; ++r)
        for (int j = 0; j < 15; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][k], bf[q], acc[0][q], 0, 0, 0);
                    acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][k], bf[q], acc[1][q], 0, 0, 0);
                }
    float s = 0.0f;
    for (int m = 0; m < 2; ++m)
        for (int q = 0; q < 16; ++q)
            for (int i = 0; i < 4; ++i) s += acc[m][q][i];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

// the product loop's operand traffic with the 16 x 16 x 32 shape: per k-step of 32 two A fragments from L2 (row tiles of 16 channels),
// sixteen B fragments from LDS (16 rows x 4 consecutive 16-byte slots, the same swizzle), 32 MFMAs - every B fragment feeds two
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_base16(const bf16x8 *wpk, float *out, int rep) {
    constexpr int P = 4, T = 256, R = T + 14 * P;
    __shared__ __attribute__((aligned(16))) unsigned char smem[R * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l16 = lane & 15, g = lane >> 4;
    fill_lds(smem, R * 256, tid);
    __syncthreads();
    f32x4 acc[2][16];
    for (int m = 0; m < 2; ++m)
        for (int q = 0; q < 16; ++q)
            for (int i = 0; i < 4; ++i) acc[m][q][i] = 0.0f;
    const bf16x8 *wp = wpk + (w * 64 + lane);                     // [k32-step (4 per tap)][row tile][wave][lane]
    for (int r = 0; r < rep; ++r) {
        bf16x8 af[2][4], bf[16];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { af[0][kk] = wp[(kk * 2) * 256]; af[1][kk] = wp[(kk * 2 + 1) * 256]; }
        {
            const unsigned char *rp0 = smem + l16 * 256 + ((g ^ l16) << 4);
#pragma unroll
            for (int q = 0; q < 16; ++q) bf[q] = *(const bf16x8 *)(rp0 + q * 4096);
        }
        for (int j = 0; j < 15; ++j) {
            const int jn = j < 14 ? j + 1 : 14;
            const int rb0 = j * P + l16, rb1 = jn * P + l16;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int rbn = (kk == 3) ? rb1 : rb0;
                const int kn = (kk + 1) & 3;
                const unsigned char *np = smem + rbn * 256 + (((4 * kn + g) ^ (rbn & 15)) << 4);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][kk], bf[q], acc[0][q], 0, 0, 0);
                    acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][kk], bf[q], acc[1][q], 0, 0, 0);
                    if (MODE == 0 || MODE == 1) {
                        bf[q] = *(const bf16x8 *)(np + q * 4096);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                if (MODE == 0 || MODE == 2) {
                    af[0][kk] = wp[((jn * 4 + kk) * 2) * 256];
                    af[1][kk] = wp[((jn * 4 + kk) * 2 + 1) * 256];
                }
            }
        }
    }
    float s = 0.0f;
    for (int m = 0; m < 2; ++m)
        for (int q = 0; q < 16; ++q)
            for (int i = 0; i < 4; ++i) s += acc[m][q][i];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

// 64 channels x 128 times per wave with the 16 x 16 x 32 shape: per k-step of 32 four A fragments, eight B fragments, 32 MFMAs
__global__ __launch_bounds__(256, 2) void k_w64_16(const bf16x8 *wpk, float *out, int rep) {
    constexpr int P = 4, T = 256, R = T + 14 * P;
    __shared__ __attribute__((aligned(16))) unsigned char smem[R * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l16 = lane & 15, g = lane >> 4;
    const int ch = w & 1, th = w >> 1;
    fill_lds(smem, R * 256, tid);
    __syncthreads();
    f32x4 acc[4][8];
    for (int m = 0; m < 4; ++m)
        for (int q = 0; q < 8; ++q)
            for (int i = 0; i < 4; ++i) acc[m][q][i] = 0.0f;
    const bf16x8 *wp = wpk + (ch * 128 + lane);                   // row tiles 4 ch .. 4 ch + 3 of every k-step (synthetic addressing)
    const unsigned char *tbase = smem + th * 8 * 4096;            // this wave's eight column tiles
    for (int r = 0; r < rep; ++r) {
        bf16x8 af[4][4], bf[8];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m][kk] = wp[(kk * 2) * 256 + m * 32];
        {
            const unsigned char *rp0 = tbase + l16 * 256 + ((g ^ l16) << 4);
#pragma unroll
            for (int q = 0; q < 8; ++q) bf[q] = *(const bf16x8 *)(rp0 + q * 4096);
        }
        for (int j = 0; j < 15; ++j) {
            const int jn = j < 14 ? j + 1 : 14;
            const int rb0 = j * P + l16, rb1 = jn * P + l16;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int rbn = (kk == 3) ? rb1 : rb0;
                const int kn = (kk + 1) & 3;
                const unsigned char *np = tbase + rbn * 256 + (((4 * kn + g) ^ (rbn & 15)) << 4);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][kk], bf[q], acc[m][q], 0, 0, 0);
                    bf[q] = *(const bf16x8 *)(np + q * 4096);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) af[m][kk] = wp[((jn * 4 + kk) * 2) * 256 + m * 32];
            }
        }
    }
    float s = 0.0f;
    for (int m = 0; m < 4; ++m)
        for (int q = 0; q < 8; ++q)
            for (int i = 0; i < 4; ++i) s += acc[m][q][i];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}


// ------------------------------------------------------------------------------------------------
// round 4: B-FRAGMENT REUSE ACROSS TAPS.  With P = 4 phases per tile the B fragment of (tap j, column tile q) is rows 4 j + 16 q ..+15 of
// the LDS image - the same rows as (tap j + 4, column tile q - 1).  The product loop reads each of the 75 x 4 distinct fragments up to four
// times (960 ds_read_b128 per tile).  Class-major order: the taps fall into four classes c = j mod 4; for one class and one k-step (kk)
// the wave holds the A fragments of its <= 4 taps x 2 row tiles (double-buffered: 64 registers) and walks the 16 + 3 row windows
// i of that class once - window i feeds tap c + 4 u into column tile i - u: 304 reads per tile, up to 8 MFMAs per read.
// The A stream is unchanged (128 fragments per tile against 120).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned ru32x4;
struct RStream { __amdgpu_buffer_rsrc_t rsrc; unsigned voff; };
__device__ __forceinline__ bf16x8 rs_load(const RStream &s, unsigned soffset) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, (int)s.voff, (int)soffset, 0));
}
// MODE 0: the loop as built into the kernels; 1: the A fragments are not re-fetched (as if the weight stream were free); 2: no
// sched_barrier behind the windows (the compiler's own schedule)
template <int NU, int MODE>
__device__ __forceinline__ void reuse_class(f32x4 (&acc)[2][16], bf16x8 (&A0)[4][2], bf16x8 (&A1)[4][2], bf16x8 (&ring)[4],
                                            const unsigned char *smem, const RStream &wp, int c, int cn, int l16, int g) {
    constexpr int NW = 15 + NU;                                   // row windows of a class: column tile 0 of tap c .. tile 15 of its last tap
    const int rsw = (4 * c + l16) & 15, rswn = (4 * cn + l16) & 15;
    const unsigned char *rowb = smem + (4 * c + l16) * 256, *rowbn = smem + (4 * cn + l16) * 256;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        bf16x8 (&cur)[4][2] = (kk & 1) ? A1 : A0;
        bf16x8 (&nxt)[4][2] = (kk & 1) ? A0 : A1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {                             // the next phase's A fragments: a whole phase (~120 MFMAs) ahead
            const int cc = kk < 3 ? c : cn, kn = kk < 3 ? kk + 1 : 0;
            int j = cc + 4 * u;
            j = j < 15 ? j : 14;                                   // class 3 has three taps: the fourth slot loads a fragment nobody uses
            if (kk < 3 && u >= NU) continue;
            if (MODE == 1) continue;
            const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((j * 4 + kn) * 8192);
            nxt[u][0] = rs_load(wp, so);
            nxt[u][1] = rs_load(wp, so + 4096u);
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
            const int n2 = n + 4, k2 = n2 / NW, i2 = n2 % NW;      // the window four ahead: this class, or the next one's first k-step
            if (k2 < 4)
                ring[n & 3] = *(const bf16x8 *)(rowb + (((4 * k2 + g) ^ rsw) << 4) + i2 * 4096);
            else
                ring[n & 3] = *(const bf16x8 *)(rowbn + ((g ^ rswn) << 4) + i2 * 4096);
            if (MODE != 2) __builtin_amdgcn_sched_barrier(0);
        }
    }
}
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_reuse16(const bf16x8 *wpk, float *out, int rep) {
    constexpr int P = 4, T = 256, R = T + 14 * P;
    __shared__ __attribute__((aligned(16))) unsigned char smem[R * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l16 = lane & 15, g = lane >> 4;
    fill_lds(smem, R * 256, tid);
    __syncthreads();
    f32x4 acc[2][16];
    for (int m = 0; m < 2; ++m)
        for (int q = 0; q < 16; ++q)
            for (int i = 0; i < 4; ++i) acc[m][q][i] = 0.0f;
    const RStream wp{__builtin_amdgcn_make_buffer_rsrc((void *)wpk, 0, 120 * 4096, 0x00020000), (unsigned)(w * 64 + lane) * 16u};
    bf16x8 A0[4][2], A1[4][2], ring[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        A0[u][0] = rs_load(wp, (unsigned)(4 * u * 4) * 8192u);
        A0[u][1] = rs_load(wp, (unsigned)(4 * u * 4) * 8192u + 4096u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        A1[u][0] = A0[(u + 1) & 3][1];
        A1[u][1] = A0[(u + 2) & 3][0];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) ring[i] = *(const bf16x8 *)(smem + l16 * 256 + ((g ^ l16) << 4) + i * 4096);
    for (int r = 0; r < rep; ++r) {
#pragma unroll 1
        for (int c = 0; c < 3; ++c) reuse_class<4, MODE>(acc, A0, A1, ring, smem, wp, c, c + 1, l16, g);
        reuse_class<3, MODE>(acc, A0, A1, ring, smem, wp, 3, 0, l16, g);
    }
    float s = 0.0f;
    for (int m = 0; m < 2; ++m)
        for (int q = 0; q < 16; ++q)
            for (int i = 0; i < 4; ++i) s += acc[m][q][i];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

// sustained rate: NL launches back to back (~0.15 s), the second half timed - the chip's power controller needs tens of milliseconds
// to settle, a pair of launches measures the transient (1277 vs 1480 TFLOP/s for the same kernel in two consecutive pairs)
template <typename F> static void run(const char *name, F launch, int rep) {
    constexpr int NL = 100;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < NL / 2; ++i) launch();
    (void)hipEventRecord(e0);
    for (int i = 0; i < NL / 2; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= (NL / 2);
    const double mf = (double)rep * 960;                         // MFMAs per wave
    printf("%-10s %.3f ms  %.0f TFLOP/s\n", name, ms, 512.0 * 4 * mf * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
    bf16x8 *wa; float *out;
    const size_t wbytes = (size_t)120 * 256 * 16;
    (void)hipMalloc(&wa, wbytes);
    (void)hipMalloc(&out, 512 * 256 * 4);
    std::vector<unsigned short> hw(wbytes / 2);
    unsigned r = 777u;
    for (auto &x : hw) {
        r = r * 1664525u + 1013904223u;
        const float u = ((r & 0xff) + ((r >> 8) & 0xff) + ((r >> 16) & 0xff) + (r >> 24)) / 255.0f - 2.0f;
        const float f = u * 0.087f;
        unsigned bits; memcpy(&bits, &f, 4);
        x = (unsigned short)(bits >> 16);
    }
    (void)hipMemcpy(wa, hw.data(), wbytes, hipMemcpyHostToDevice);
    const int rep = 32;
    printf("operands: activations ~ N(0, 0.5^2), weights ~ N(0, 0.05^2); 512 workgroups (2 per CU), %d tiles each\n", rep);
    for (int pass = 0; pass < 2; ++pass) {
        run("base", [&] { k_base<0><<<512, 256>>>(wa, out, rep); }, rep);
        run("noA", [&] { k_base<1><<<512, 256>>>(wa, out, rep); }, rep);
        run("noB", [&] { k_base<2><<<512, 256>>>(wa, out, rep); }, rep);
        run("noAB", [&] { k_base<3><<<512, 256>>>(wa, out, rep); }, rep);
        run("w64x128", [&] { k_w64<<<512, 256>>>(wa, out, rep); }, rep);
        run("bare16x16", [&] { k_bare16<<<512, 256>>>(wa, out, rep); }, rep);
        run("base16x16", [&] { k_base16<0><<<512, 256>>>(wa, out, rep); }, rep);
        run("noA16x16", [&] { k_base16<1><<<512, 256>>>(wa, out, rep); }, rep);
        run("noB16x16", [&] { k_base16<2><<<512, 256>>>(wa, out, rep); }, rep);
        run("w64_16x16", [&] { k_w64_16<<<512, 256>>>(wa, out, rep); }, rep);
        run("reuse16x16", [&] { k_reuse16<0><<<512, 256>>>(wa, out, rep); }, rep);
        run("base16x16", [&] { k_base16<0><<<512, 256>>>(wa, out, rep); }, rep);
        run("reuse16x16", [&] { k_reuse16<0><<<512, 256>>>(wa, out, rep); }, rep);
        run("reuse_noA", [&] { k_reuse16<1><<<512, 256>>>(wa, out, rep); }, rep);
        // (k_reuse16<2>, no sched_barrier behind the windows: hipcc hoists the LDS reads and spills 391 registers - not worth a run)
    }
    return 0;
}
