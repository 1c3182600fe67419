// Micro-benchmark: how fast can ONE wave per SIMD issue v_mfma_f32_32x32x16_bf16 when each MFMA is paired with the
// ds_read_b128 that refills its B operand (the TCN main loop), compared with two waves per SIMD and with batched reads.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_issue tools/micro/mfma_issue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

// MODE 0: bare MFMAs   1: MFMA + ds_read each (ring 8, pinned)   2: 8 MFMAs then 8 ds_reads (double ring)
// MODE 3: as 1 plus one global A load per 8 MFMAs   4: as 1 but ring refilled 16 MFMAs ahead (two rings)
template <int MODE, int NT, int NV = 0, int RND = 0>
__global__ __launch_bounds__(NT, 1) void k(const bf16x8 *wa, float *out, long long *clk, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[80 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, ln = lane & 31, h = lane >> 5;
    for (int i = tid; i < 80 * 1024 / 4; i += NT) {
        unsigned v = 0x3c003c00u + (i & 7);
        if (RND) {             // two pseudo-random bf16 in (-1, 1): random sign + mantissa, exponent 0x3e/0x3f
            unsigned r = (unsigned)i * 2654435761u + blockIdx.x * 40503u; r ^= r >> 15; r *= 2246822519u; r ^= r >> 13;
            v = (r & 0x807f80ffu) | 0x3f003e00u | ((r >> 3) & 0x007f0000u);
        }
        ((unsigned *)smem)[i] = v;
    }
    __syncthreads();
    f32x16 acc[8];
    for (int q = 0; q < 8; ++q)
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.0f;
    bf16x8 af[8], bf[8], bg[8];
    float va[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    const bf16x8 *wp = wa + tid;
    for (int kc = 0; kc < 8; ++kc) af[kc] = wp[kc * 256];
    const unsigned char *rp0 = smem + ln * 256 + ((h ^ (ln & 15)) << 4);
    for (int q = 0; q < 8; ++q) { bf[q] = *(const bf16x8 *)(rp0 + q * 8192); bg[q] = bf[q]; }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int rb = (it & 7) * 4 + ln;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            const unsigned char *np = smem + rb * 256 + (((2 * ((kc + 1) & 7) + h) ^ (rb & 15)) << 4);
            if constexpr (MODE == 2) {
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kc], (kc & 1) ? bg[q] : bf[q], acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (kc & 1) bg[q] = *(const bf16x8 *)(np + q * 8192);
                    else bf[q] = *(const bf16x8 *)(np + q * 8192);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if constexpr (MODE == 4) {
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kc], (kc & 1) ? bg[q] : bf[q], acc[q], 0, 0, 0);
                        if (kc & 1) bg[q] = *(const bf16x8 *)(np + q * 8192);
                        else bf[q] = *(const bf16x8 *)(np + q * 8192);
                    } else {
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kc], bf[q], acc[q], 0, 0, 0);
                        if constexpr (MODE != 0) bf[q] = *(const bf16x8 *)(np + q * 8192);
                    }
                    if constexpr (NV > 0) {
#pragma unroll
                        for (int r = 0; r < NV; ++r) va[r & 7] = va[r & 7] * 1.0001f + 0.5f;
                    }
                    if constexpr (MODE != 0) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        if constexpr (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
                    }
                }
            }
            if constexpr (MODE == 3) af[kc] = wp[((it + 1) * 8 + kc) % 120 * 256];
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int q = 0; q < 8; ++q)
        for (int i = 0; i < 16; ++i) s += acc[q][i];
    for (int r = 0; r < 8; ++r) s += va[r];
    out[(size_t)blockIdx.x * NT + tid] = s;
    if (lane == 0) clk[(size_t)blockIdx.x * (NT / 64) + (tid >> 6)] = t1 - t0;
}

template <int MODE, int NT, int NV = 0, int RND = 0> void run(const char *name, const bf16x8 *wa, float *out, long long *clk, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NT, NV, RND><<<256, NT>>>(wa, out, clk, iters);
    hipEventRecord(e0);
    k<MODE, NT, NV, RND><<<256, NT>>>(wa, out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256 * (NT / 64));
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= h.size();
    const double mf = (double)iters * 64;
    printf("%-46s waves/SIMD %d  ticks/MFMA per wave %.1f  per SIMD %.1f  | %.3f ms  %.0f TFLOP/s\n", name, NT / 256, m / mf, m / mf / (NT / 256),
           ms, 256.0 * (NT / 64) * mf * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
    bf16x8 *wa; float *out; long long *clk;
    hipMalloc(&wa, 120 * 8 * 256 * 16 * 2); hipMemset(wa, 0x3c, 120 * 8 * 256 * 16 * 2);
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8 * 8);
    const int iters = 2000;
    run<0, 256>("bare MFMA", wa, out, clk, iters);
    run<1, 256>("MFMA + ds_read_b128 each (ring 8)", wa, out, clk, iters);
    run<4, 256>("MFMA + ds_read_b128 each (ring 16)", wa, out, clk, iters);
    run<2, 256>("8 MFMA then 8 ds_read_b128 (ring 16)", wa, out, clk, iters);
    run<3, 256>("MFMA + ds_read each + A global load / 8", wa, out, clk, iters);
    run<1, 256, 2>("MFMA + ds_read + 2 VALU fma each", wa, out, clk, iters);
    run<1, 256, 4>("MFMA + ds_read + 4 VALU fma each", wa, out, clk, iters);
    run<1, 256, 8>("MFMA + ds_read + 8 VALU fma each", wa, out, clk, iters);
    run<1, 256, 16>("MFMA + ds_read + 16 VALU fma each", wa, out, clk, iters);
    run<1, 512, 2>("MFMA + ds_read + 2 VALU fma each", wa, out, clk, iters);
    run<1, 512, 4>("MFMA + ds_read + 4 VALU fma each", wa, out, clk, iters);
    run<1, 512, 8>("MFMA + ds_read + 8 VALU fma each", wa, out, clk, iters);
    run<0, 512>("bare MFMA", wa, out, clk, iters);
    run<1, 512>("MFMA + ds_read_b128 each (ring 8)", wa, out, clk, iters);
    run<2, 512>("8 MFMA then 8 ds_read_b128 (ring 16)", wa, out, clk, iters);
    run<3, 512>("MFMA + ds_read each + A global load / 8", wa, out, clk, iters);
    {   // random A operands too
        std::vector<unsigned> hw(120 * 8 * 256 * 16 * 2 / 4);
        unsigned r = 12345u;
        for (auto &x : hw) { r = r * 1664525u + 1013904223u; x = (r & 0x807f807fu) | 0x3e003f00u | ((r >> 7) & 0x00800080u); }
        (void)hipMemcpy(wa, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    }
    run<3, 512, 0, 1>("random data: MFMA + ds_read + A load / 8", wa, out, clk, iters);
    run<1, 512, 0, 1>("random data: MFMA + ds_read_b128 each", wa, out, clk, iters);
    run<0, 512, 0, 1>("random data: bare MFMA", wa, out, clk, iters);
    run<3, 256, 0, 1>("random data: MFMA + ds_read + A load / 8", wa, out, clk, iters);
    return 0;
}
