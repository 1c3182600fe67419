// Micro-benchmark (round 4): the bandwidth gate of a transform-domain (overlap-save, N = 32) TCN block kernel.
// Its dominant phase is one complex 128 x 128 GEMM per frequency bin: Y[co][col] = sum_ci W_bin[co][ci] X_bin[ci][col], col = (time block, re | im).
// Every bin has its own weights (16 packed bins x 2 parts x 128 x 128 bf16 = 1 MB per TCN block) and a workgroup can only keep the spectra of
// NB * 8 time blocks in LDS (64 KB per 8 blocks of 16 outputs), so a weight fragment fetched from L2 feeds only NB MFMAs - against 16 in the
// direct kernel's main loop.  This measures what the weight stream lets through:
//   wg8     8 time blocks (128 outputs) per tile, 64 KB of LDS, two workgroups per CU, weights from L2      (1 MFMA per weight fragment)
//   wg16    16 time blocks (256 outputs) per tile, 128 KB of LDS, one workgroup per CU, weights from L2     (2 MFMAs per weight fragment)
//   wg32x   32 time blocks per tile (does NOT fit: the same 128 KB read twice) - what 4 MFMAs per fragment would give
//   noA8/16 the same loops with the weights loaded once (no weight stream): the matrix-pipe + LDS bound
// Reported: ms per LAUNCH EQUIVALENT = the time this phase would need for one TCN block of 32 x 131072 output steps (32768 / NB tiles), next to
// the direct kernel's 1.47-1.50 ms for the whole block.  The other two phases (DFT, inverse DFT: 16 % of the MFMAs) are not included.
//   hipcc --offload-arch=gfx950 -O3 -o fft_bin_gemm fft_bin_gemm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

__device__ inline void fill_lds(unsigned char *smem, int bytes, int tid) {
    for (int i = tid; i < bytes / 4; i += 256) {
        unsigned r = (unsigned)i * 2654435761u + blockIdx.x * 40503u; r ^= r >> 15; r *= 2246822519u; r ^= r >> 13;
        auto g = [&](unsigned z) { z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16;
                                   const float u = ((z & 0xff) + ((z >> 8) & 0xff) + ((z >> 16) & 0xff) + (z >> 24)) / 255.0f - 2.0f;
                                   return (unsigned)(__builtin_bit_cast(unsigned, u * 0.87f) >> 16); };
        ((unsigned *)smem)[i] = g(r) | (g(r * 747796405u + 2891336453u) << 16);
    }
}

// NBL: column tiles (of 8 time blocks x {re, im}) resident in LDS; NBX: column tiles computed per weight fragment (NBX > NBL re-reads LDS)
template <int NBL, int NBX, bool STREAM, int WGS>
__global__ __launch_bounds__(256, WGS) void k_bin(const bf16x8 *wpk, float *out, int rep) {
    constexpr int ROWS = 32 * 8 * NBL;                // (bin, part, column) rows of 128 input channels = 256 B
    __shared__ __attribute__((aligned(16))) unsigned char smem[ROWS * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l16 = lane & 15, g = lane >> 4;
    fill_lds(smem, ROWS * 256, tid);
    __syncthreads();
    f32x4 tot = {0.0f, 0.0f, 0.0f, 0.0f};
    const bf16x8 *wp = wpk + (w * 64 + lane);         // [bin][k-step][frag = (re | im) x row tile][wave][lane]
    const int part = l16 >> 3, col = l16 & 7;
    bf16x8 af[2][4];
#pragma unroll
    for (int f = 0; f < 4; ++f) af[0][f] = wp[f * 256];
    for (int r = 0; r < rep; ++r) {
        for (int bin = 0; bin < 16; ++bin) {
            f32x4 acc[4][NBX];
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int n = 0; n < NBX; ++n) acc[f][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (STREAM) {          // the fragments of the next k-step (of the next bin behind the last) travel while this one is computed on
                    const int nb = ks < 3 ? bin : (bin + 1) & 15, nk = (ks + 1) & 3;
#pragma unroll
                    for (int f = 0; f < 4; ++f) af[nxt][f] = wp[((nb * 4 + nk) * 4 + f) * 256];
                }
                bf16x8 bf[NBX];
#pragma unroll
                for (int n = 0; n < NBX; ++n) {
                    const int row = ((bin * 2 + part) * NBL + (n % NBL)) * 8 + col;
                    bf[n] = *(const bf16x8 *)(smem + row * 256 + (((4 * ks + g) ^ (row & 15)) << 4));
                }
#pragma unroll
                for (int n = 0; n < NBX; ++n)
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[STREAM ? cur : 0][f], bf[n], acc[f][n], 0, 0, 0);
            }
            // combine: Y_re = (W_re X)[re] - (W_im X)[im], Y_im = (W_re X)[im] + (W_im X)[re]: the partner column sits 8 lanes away
#pragma unroll
            for (int n = 0; n < NBX; ++n)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float o = __shfl_xor(acc[2 + m][n][i], 8);
                        tot[i] += acc[m][n][i] + (part ? o : -o);
                    }
        }
    }
    out[(size_t)blockIdx.x * 256 + tid] = tot[0] + tot[1] + tot[2] + tot[3];
}

template <typename F> static void run(const char *name, F launch, int grid, int rep, int nbx) {
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
    const double tiles = (double)grid * rep;                        // tiles of 128 * nbx output steps
    const double mfma = tiles * 4 /*waves*/ * 16 * 4 * 4 * nbx;     // instructions
    const double equiv = ms * (32.0 * 131072 / (128.0 * nbx)) / tiles;
    printf("%-8s %.3f ms  %7.0f TFLOP/s on the pipe  weight stream %6.2f TB/s  -> %.3f ms per launch equivalent (bin GEMMs only)\n", name, ms,
           mfma * 16384.0 / (ms * 1e-3) / 1e12, tiles * 1048576.0 / (ms * 1e-3) / 1e12, equiv);
}

int main() {
    bf16x8 *wa; float *out;
    const size_t wbytes = (size_t)16 * 4 * 4 * 256 * 16;          // 1 MB
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
    for (int pass = 0; pass < 2; ++pass) {
        run("wg8", [&] { k_bin<1, 1, true, 2><<<512, 256>>>(wa, out, 32); }, 512, 32, 1);
        run("noA8", [&] { k_bin<1, 1, false, 2><<<512, 256>>>(wa, out, 32); }, 512, 32, 1);
        run("wg16", [&] { k_bin<2, 2, true, 1><<<256, 256>>>(wa, out, 32); }, 256, 32, 2);
        run("noA16", [&] { k_bin<2, 2, false, 1><<<256, 256>>>(wa, out, 32); }, 256, 32, 2);
        run("wg32x", [&] { k_bin<2, 4, true, 1><<<256, 256>>>(wa, out, 16); }, 256, 16, 4);
        run("wg64x", [&] { k_bin<2, 8, true, 1><<<256, 256>>>(wa, out, 8); }, 256, 8, 8);
    }
    return 0;
}
