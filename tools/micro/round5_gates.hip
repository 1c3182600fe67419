// Gates for two kernels that round 4 did not build (EXPERIMENTS.md §C.11) - each answers "can this pay?" in a few seconds of GPU time.
//
//   fuse0_*   Block 0 (2 -> 128 channels, 0.31 ms per step, HBM-bound on its 1.07 GB store) could be computed by the LOADER waves of block 1's
//             duo kernel straight into the LDS image (no store, no re-read).  What does that compute cost block 1?  One workgroup of eight
//             waves per CU, persistent: waves 0-3 run the class-major main loop (tcn_reuse_class, the product's), waves 4-7 either idle
//             (fuse0_idle) or do a stand-in for block 0's work on one 284-row tile per main-loop tile (fuse0_busy): per wave nine column
//             tiles of 32 times x 32 channels = 36 v_mfma_f32_32x32x16_bf16 (hi + lo, two k-steps) + the operand build (16 values split
//             into hi / lo per column tile and k-step) + the epilogue (16 elements per lane and column tile: leaky, FiLM, residual, convert)
//             + 36 ds_write_b64; one barrier per tile.  Fusion pays if  (fuse0_busy - fuse0_idle) per launch equivalent  <  0.31 ms.
//   x3_bare   The split-bf16 class-major main loop alone (tcn_reuse_class_x3, two workgroups of four waves per CU, 128-time two-phase
//             tiles, no staging, no epilogue): what a split-bf16 kernel with perfectly hidden staging / epilogue would run at
//             (the product kernel: 4.1 ms per launch at 32 x 131072).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../music_mixing_style_transfer_amd/csrc -o round5_gates round5_gates.hip
#include "tcn_kernels.h"

#include <vector>

__device__ __forceinline__ unsigned gate_bf16(unsigned z, float sigma) { return calib_normalish_bf16(z, sigma); }

// ---- fuse0: 256 workgroups x 512 threads, `rep` tiles each (rep = 64: the 16384 tiles of one dense launch at 32 x 131072)
template <bool BUSY>
__global__ __launch_bounds__(512, 1) void k_fuse0(const void *wpk, const float *wave_in, float *out, int rep) {
    constexpr int P = 4, T = 256, R = T + 14 * P;
    __shared__ __attribute__((aligned(16))) unsigned char smem[R * 256];
    __shared__ __attribute__((aligned(16))) unsigned char dump[4 * 9 * 4 * 512];      // where the stand-in writes its rows (73 KB)
    __shared__ float xs[2][320];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, w = wv & 3, l16 = lane & 15, g = lane >> 4;
    for (int i = tid; i < R * 64; i += 512) {
        const unsigned r = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        ((unsigned *)smem)[i] = gate_bf16(r, 0.5f) | (gate_bf16(r * 747796405u + 2891336453u, 0.5f) << 16);
    }
    for (int i = tid; i < 640; i += 512) xs[i / 320][i % 320] = wave_in[(blockIdx.x * 640 + i) & 0xffff];
    __syncthreads();
    if (wv < 4) {
        // ---------------------------------------------------------------- matrix waves: the product's class-major loop
        __builtin_amdgcn_s_setprio(2);
        f32x4 acc[2][16];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[m][q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const MstStream16 wst = mst_stream16(wpk, 60u * 2u * 4096u);
        const unsigned aoff = (unsigned)(w * 64 + lane) * 16u;
        bf16x8 A0[4][2], A1[4][2], ring[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int m = 0; m < 2; ++m) A0[u][m] = __builtin_bit_cast(bf16x8, mst_stream_load16(wst, aoff + m * 4096, (unsigned)(16 * u) * 8192u));
        for (int r = 0; r < rep; ++r) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ring[i] = *(const bf16x8 *)(smem + l16 * 256 + ((g ^ l16) << 4) + i * 4096);
#pragma unroll 1
            for (int c = 0; c < 3; ++c) tcn_reuse_class<P, 4, false, 4>(acc, A0, A1, ring, smem, wst, aoff, c, c + 1, l16, g);
            tcn_reuse_class<P, 3, true, 4>(acc, A0, A1, ring, smem, wst, aoff, 3, 0, l16, g);
            mst_dma_wait_barrier<63>();
        }
        float s = 0.0f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) s += acc[m][q][0] + acc[m][q][1] + acc[m][q][2] + acc[m][q][3];
        out[(size_t)blockIdx.x * 512 + tid] = s;
    } else {
        // ---------------------------------------------------------------- loader waves: idle, or a stand-in for block 0 on 284 rows
        const int ln = lane & 31, h = lane >> 5;
        const bf16x8 af0 = ((const bf16x8 *)wpk)[w * 64 + lane], af1 = ((const bf16x8 *)wpk)[256 + w * 64 + lane];
        float keep = 0.0f;
        for (int r = 0; r < rep; ++r) {
            if (BUSY) {
#pragma unroll 1
                for (int q = 0; q < 9; ++q) {                     // nine column tiles of 32 times
                    f32x16 acc;
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = 0.01f * (float)i;
#pragma unroll
                    for (int sI = 0; sI < 2; ++sI) {
                        bf16x8 hi, lo;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int k = 16 * sI + 8 * h + e;
                            const int ci = k >= 15 ? 1 : 0, j = k - 15 * ci;
                            const float v = k < 30 ? xs[ci][32 * q + ln + j] : 0.0f;
                            const __bf16 vh = (__bf16)v;
                            hi[e] = vh;
                            lo[e] = (__bf16)(v - (float)vh);
                        }
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sI ? af1 : af0, hi, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sI ? af1 : af0, lo, acc, 0, 0, 0);
                    }
                    const float xres = xs[w >> 1][32 * q + ln + 7];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        bf16x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float v = acc[4 * gq + i];
                            o[i] = (__bf16)(1.01f * mst_fmax(v, MST_LEAKY * v) + (0.02f + 0.5f * xres));
                        }
                        *(bf16x4 *)(dump + (((w * 9 + q) * 4 + gq) * 64 + lane) * 8) = o;
                    }
                }
                keep += (float)*(const __bf16 *)(dump + lane * 8);
            }
            mst_dma_wait_barrier<63>();
        }
        if (keep == 123.456f) out[tid] = keep;
    }
}

// ---- x3_bare: 512 workgroups x 256 threads (two per CU), `rep` 128-time tiles each (rep = 64: one dense launch)
__global__ __launch_bounds__(256, 2) void k_x3_bare(const void *wpk, float *out, int rep) {
    constexpr int P = 2, NC = 8, R = 128 + 14 * P;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * R * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l16 = lane & 15, g = lane >> 4;
    for (int i = tid; i < 2 * R * 64; i += 256) {
        const unsigned r = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        const float sg = i < R * 64 ? 0.5f : 0.5f / 256.0f;          // the lo image is 2^-8 of the hi image
        ((unsigned *)smem)[i] = gate_bf16(r, sg) | (gate_bf16(r * 747796405u + 2891336453u, sg) << 16);
    }
    __syncthreads();
    const unsigned char *sm_hi = smem, *sm_lo = smem + R * 256;
    f32x4 acc[2][NC];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < NC; ++q) acc[m][q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const unsigned char *wbase = (const unsigned char *)wpk;
    const unsigned aoff = (unsigned)(w * 64 + lane) * 16u;
    constexpr size_t LO_IMG = (size_t)120 * 4096;
    constexpr int NCLS = 16 / P, NUMAX = 2;
    bf16x8 H0[NUMAX][2], L0[NUMAX][2], H1[NUMAX][2], L1[NUMAX][2], rh[4], rl[4];
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int u = 0; u < NUMAX; ++u)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                H0[u][m] = *(const bf16x8 *)(wbase + (size_t)((NCLS * u * 4) * 2 + m) * 4096 + aoff);
                L0[u][m] = *(const bf16x8 *)(wbase + LO_IMG + (size_t)((NCLS * u * 4) * 2 + m) * 4096 + aoff);
            }
        const int o0 = l16 * 256 + ((g ^ l16) << 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rh[i] = *(const bf16x8 *)(sm_hi + o0 + i * 4096);
            rl[i] = *(const bf16x8 *)(sm_lo + o0 + i * 4096);
        }
#pragma unroll 1
        for (int c = 0; c < NCLS - 1; ++c)
            tcn_reuse_class_x3<P, NUMAX, false, NUMAX, NC>(acc, H0, L0, H1, L1, rh, rl, sm_hi, sm_lo, wbase, LO_IMG, aoff, c, c + 1, l16, g);
        tcn_reuse_class_x3<P, 15 / NCLS, true, NUMAX, NC>(acc, H0, L0, H1, L1, rh, rl, sm_hi, sm_lo, wbase, LO_IMG, aoff, NCLS - 1, 0, l16, g);
    }
    float s = 0.0f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < NC; ++q) s += acc[m][q][0] + acc[m][q][1] + acc[m][q][2] + acc[m][q][3];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <typename F> static float run(const char *name, F launch, const char *what) {
    constexpr int NL = 60;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < NL / 2; ++i) launch();
    (void)hipEventRecord(e0);
    for (int i = 0; i < NL / 2; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= (NL / 2);
    printf("%-12s %.3f ms per launch equivalent  (%s)\n", name, ms, what);
    return ms;
}

int main() {
    void *w; float *out, *wav;
    const size_t wbytes = (size_t)2 * 120 * 4096;               // hi and lo fragment images
    (void)hipMalloc(&w, wbytes);
    (void)hipMalloc(&out, (size_t)512 * 512 * 4);
    (void)hipMalloc(&wav, 65536 * 4);
    tcn_calib_fill_kernel<<<(unsigned)(wbytes / 4 + 255) / 256, 256>>>((unsigned *)w, (int)(wbytes / 4));
    std::vector<float> hw(65536);
    unsigned r = 99u;
    for (auto &x : hw) { r = r * 1664525u + 1013904223u; x = ((r >> 8) & 0xffff) / 32768.0f - 1.0f; }
    (void)hipMemcpy(wav, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; ++pass) {
        const float a = run("fuse0_idle", [&] { k_fuse0<false><<<256, 512>>>(w, wav, out, 64); }, "duo main loop, loader waves idle");
        const float b = run("fuse0_busy", [&] { k_fuse0<true><<<256, 512>>>(w, wav, out, 64); }, "loader waves compute a block-0 stand-in per tile");
        printf("   block 0 in the loader waves costs %.3f ms per launch of block 1; the separate block-0 kernel is 0.31 ms\n", b - a);
        run("x3_bare", [&] { k_x3_bare<<<512, 256>>>(w, out, 64); }, "split-bf16 class-major main loop alone; product kernel 4.1 ms");
    }
    return 0;
}
