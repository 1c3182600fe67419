// Micro-benchmark: the EXACT main loop of tcn_block_bf16_kernel<4, 2, false, 8> (15 taps x 8 k-chunks x 8 column tiles, P = 4
// row addressing, 120 A fragments per wave from L2, B ring from a swizzled LDS tile), two 256-thread workgroups per CU,
// random bf16 data, repeated REP times per workgroup with no staging and no epilogue.  Separates what the main loop itself
// sustains from what the phases around it cost.   hipcc --offload-arch=gfx950 -O3 -o tcn_mainloop tcn_mainloop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int P, int NQ>
__global__ __launch_bounds__(256, (NQ > 8 ? 1 : 2)) void k(const bf16x8 *wpk, float *out, long long *clk, int rep, int gauss) {
    constexpr int T = 32 * NQ, R = T + 14 * P;
    __shared__ __attribute__((aligned(16))) unsigned char smem[R * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ln = lane & 31, h = lane >> 5;
    for (int i = tid; i < R * 256 / 4; i += 256) {
        unsigned r = (unsigned)i * 2654435761u + blockIdx.x * 40503u; r ^= r >> 15; r *= 2246822519u; r ^= r >> 13;
        unsigned v = (r & 0x807f80ffu) | 0x3f003e00u | ((r >> 3) & 0x007f0000u);
        if (gauss) {           // two bf16 values ~ N(0, 0.5^2): sum of four uniforms, like post-BN activations
            auto g = [&](unsigned z) { z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16;
                                       const float u = ((z & 0xff) + ((z >> 8) & 0xff) + ((z >> 16) & 0xff) + (z >> 24)) / 255.0f - 2.0f;
                                       return (unsigned)(__builtin_bit_cast(unsigned, u * 0.87f) >> 16); };
            v = g(r) | (g(r * 747796405u + 2891336453u) << 16);
        }
        ((unsigned *)smem)[i] = v;
    }
    __syncthreads();
    f32x16 acc[NQ];
    for (int q = 0; q < NQ; ++q)
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.0f;
    const bf16x8 *wp = wpk + (w * 64 + lane);
    const long long t0 = __builtin_readcyclecounter();
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
                    bf[q] = *(const bf16x8 *)(np + q * 8192);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                af[kc] = wp[(jn * 8 + kc) * 256];
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int q = 0; q < NQ; ++q)
        for (int i = 0; i < 16; ++i) s += acc[q][i];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (lane == 0) clk[(size_t)blockIdx.x * 4 + w] = t1 - t0;
}

int main() {
    bf16x8 *wa; float *out; long long *clk;
    const size_t wbytes = (size_t)120 * 256 * 16;
    (void)hipMalloc(&wa, wbytes);
    {
        std::vector<unsigned> hw(wbytes / 4);
        unsigned r = 12345u;
        for (auto &x : hw) { r = r * 1664525u + 1013904223u; x = (r & 0x807f807fu) | 0x3e003f00u | ((r >> 7) & 0x00800080u); }
        (void)hipMemcpy(wa, hw.data(), wbytes, hipMemcpyHostToDevice);
    }
    (void)hipMalloc(&out, 512 * 256 * 4); (void)hipMalloc(&clk, 512 * 4 * 8);
    const int rep = 32;
    for (int gauss = 0; gauss < 2; ++gauss) {
    printf("---- operands: %s\n", gauss ? "approximately normal (activations sigma 0.5, weights sigma 0.05)" : "random bit patterns");
    if (gauss) {
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
    }
    {   // 512-time tiles: 16 accumulator tiles per wave (AGPRs), one workgroup per CU, half the A loads per MFMA
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<4, 16><<<256, 256>>>(wa, out, clk, rep / 2, gauss);
        (void)hipEventRecord(e0);
        k<4, 16><<<256, 256>>>(wa, out, clk, rep / 2, gauss);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(256 * 4);
        (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
        double m = 0; for (auto v : h) m += v; m /= h.size();
        const double mf = (double)(rep / 2) * 1920;
        printf("512-time tiles (NQ = 16), 1 workgroup per CU: %.1f clk/MFMA per SIMD | %.3f ms  %.0f TFLOP/s\n", m / mf, ms,
               256.0 * 4 * mf * 32768.0 / (ms * 1e-3) / 1e12);
    }
    for (int wgs : {512, 256}) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<4, 8><<<wgs, 256>>>(wa, out, clk, rep, gauss);
        (void)hipEventRecord(e0);
        k<4, 8><<<wgs, 256>>>(wa, out, clk, rep, gauss);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(wgs * 4);
        (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
        double m = 0; for (auto v : h) m += v; m /= h.size();
        const double mf = (double)rep * 960;
        printf("TCN main loop only, %d workgroups (%d per CU): %.1f clk/MFMA per wave = %.1f per SIMD | %.3f ms  %.0f TFLOP/s\n", wgs, wgs / 256,
               m / mf, m / mf / (wgs / 256), ms, (double)wgs * 4 * mf * 32768.0 / (ms * 1e-3) / 1e12);
    }
    }
    return 0;
}
