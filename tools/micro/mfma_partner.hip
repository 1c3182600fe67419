// Micro-benchmark: one MFMA-streaming wave per SIMD (the TCN main loop: MFMA + ds_read_b128 per MFMA) next to a partner
// wave on the same SIMD doing VALU / LDS / VMEM work.  Reports clocks per MFMA of the streaming wave and clocks per
// partner operation, against each running alone.   Build: hipcc --offload-arch=gfx950 -O3 -o mfma_partner mfma_partner.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

// PARTNER 0: none (partner waves exit)  1: idle (s_sleep)  2: VALU fma  3: LDS ds_write_b64 + ds_read_b128 (private rows)
//         4: global_load_dwordx4 (L2 hits)  5: global_store_dwordx4  6: epilogue-like mix (VALU + ds_write_b64 + loads)
// MFMA_ON 0: the streaming set only spins on the flag (partner alone)
template <int PARTNER, int MFMA_ON, int PRIO = 0>
__global__ __launch_bounds__(512, 1) void k(const bf16x8 *wa, float *out, long long *clk, float *gbuf, int iters, int piters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[144 * 1024];
    __shared__ volatile int done;
    const int set = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, ln = lane & 31, h = lane >> 5, w = tid >> 6;
    for (int i = threadIdx.x; i < 144 * 1024 / 4; i += 512) ((unsigned *)smem)[i] = 0x3c003c00u + (i & 7);
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (set == 0) {
        long long t0 = 0, t1 = 0;
        float s = 0.0f;
        if (MFMA_ON) {
            f32x16 acc[8];
            for (int q = 0; q < 8; ++q)
                for (int i = 0; i < 16; ++i) acc[q][i] = 0.0f;
            bf16x8 af[8], bf[8];
            const bf16x8 *wp = wa + tid;
            for (int kc = 0; kc < 8; ++kc) af[kc] = wp[kc * 256];
            const unsigned char *rp0 = smem + ln * 256 + ((h ^ (ln & 15)) << 4);
            for (int q = 0; q < 8; ++q) bf[q] = *(const bf16x8 *)(rp0 + q * 8192);
            t0 = __builtin_readcyclecounter();
            for (int it = 0; it < iters; ++it) {
                const int rb = (it & 7) * 4 + ln;
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const unsigned char *np = smem + rb * 256 + (((2 * ((kc + 1) & 7) + h) ^ (rb & 15)) << 4);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kc], bf[q], acc[q], 0, 0, 0);
                        bf[q] = *(const bf16x8 *)(np + q * 8192);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    af[kc] = wp[(((it + 1) & 7) * 8 + kc) * 256];
                }
            }
            t1 = __builtin_readcyclecounter();
            for (int q = 0; q < 8; ++q)
                for (int i = 0; i < 16; ++i) s += acc[q][i];
        } else {
            t0 = __builtin_readcyclecounter();
            while (done < 4) __builtin_amdgcn_s_sleep(20);
            t1 = __builtin_readcyclecounter();
        }
        out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
        if (lane == 0) clk[(size_t)blockIdx.x * 8 + w] = t1 - t0;
        if (MFMA_ON && lane == 0) atomicAdd((int *)&done, 1);
    } else {
        if (PARTNER == 0) { if (lane == 0) clk[(size_t)blockIdx.x * 8 + 4 + w] = 0; return; }
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
        unsigned char *mine = smem + 80 * 1024 + w * 16384;          // private 16 KB of LDS per partner wave
        float *g = gbuf + ((size_t)blockIdx.x * 256 + tid) * 4;
        const size_t gstride = (size_t)256 * 256 * 4;                 // next "row" of the global scratch (stays in L2)
        f32x4 v0 = {1.0f, 2.0f, 3.0f, 4.0f}, v1 = v0, v2 = v0, v3 = v0;
        long long t0 = __builtin_readcyclecounter(), t1 = 0;
        int n = 0;
        for (;; ++n) {
            if (n == piters) {
                t1 = __builtin_readcyclecounter();
                if (!MFMA_ON) break;
            }
            if (MFMA_ON && n >= piters && done >= 4) break;
            if (PARTNER == 1) __builtin_amdgcn_s_sleep(10);
            if (PARTNER == 2) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    v0 = v0 * v1 + v2; v1 = v1 * v2 + v3; v2 = v2 * v3 + v0; v3 = v3 * v0 + v1;     // 16 scalar fma / pk_fma per r
                }
            }
            if (PARTNER == 3) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    *(bf16x4 *)(mine + (r * 64 + lane) * 8 + ((n & 3) << 12)) = __builtin_bit_cast(bf16x4, (double)v0[0]);
                    f32x4 t = *(const f32x4 *)(mine + ((r * 64 + lane) ^ 5) * 16);
                    v0 += t;
                }
            }
            if (PARTNER == 4) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v0 += *(const f32x4 *)(g + ((size_t)((n * 8 + r) & 15)) * gstride);
            }
            if (PARTNER == 5) {
#pragma unroll
                for (int r = 0; r < 8; ++r) *(f32x4 *)(g + ((size_t)((n * 8 + r) & 15)) * gstride) = v0;
            }
            if (PARTNER == 6) {
                const f32x4 p = *(const f32x4 *)(g + ((size_t)(n & 15)) * gstride);
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    f32x4 t = *(const f32x4 *)(mine + ((r * 64 + lane) ^ 5) * 16);
                    f32x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float a = t[i] + p[i];
                        a = fmaxf(a, 0.01f * a);
                        a = p[(i + 1) & 3] * a + p[(i + 2) & 3];
                        o[i] = a + v1[i] * t[(i + 1) & 3];
                    }
                    v0 += o;
                    bf16x4 ob = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
                    *(bf16x4 *)(mine + (r * 64 + lane) * 8 + ((n & 3) << 12)) = ob;
                }
            }
        }
        out[(size_t)blockIdx.x * 512 + threadIdx.x] = v0[0] + v1[1] + v2[2] + v3[3];
        if (lane == 0) clk[(size_t)blockIdx.x * 8 + 4 + w] = t1 - t0;
        if (!MFMA_ON && lane == 0) atomicAdd((int *)&done, 1);
    }
}

template <int PARTNER, int MFMA_ON, int PRIO = 0> void run(const char *name, const bf16x8 *wa, float *out, long long *clk, float *gbuf, int iters, int piters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<PARTNER, MFMA_ON, PRIO><<<256, 512>>>(wa, out, clk, gbuf, iters, piters);
    (void)hipEventRecord(e0);
    k<PARTNER, MFMA_ON, PRIO><<<256, 512>>>(wa, out, clk, gbuf, iters, piters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256 * 8);
    (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0, p = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 4; ++w) { m += h[b * 8 + w]; p += h[b * 8 + 4 + w]; }
    m /= 1024; p /= 1024;
    printf("%-44s mfma %s: %6.1f clk/MFMA   partner: %8.1f clk per iteration (8 ops)   kernel %.3f ms\n", name, MFMA_ON ? "on " : "off",
           MFMA_ON ? m / ((double)iters * 64) : 0.0, p / piters, ms);
}

int main() {
    bf16x8 *wa; float *out; long long *clk; float *gbuf;
    (void)hipMalloc(&wa, 120 * 8 * 256 * 16 * 2); (void)hipMemset(wa, 0x3c, 120 * 8 * 256 * 16 * 2);
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&clk, 256 * 8 * 8);
    (void)hipMalloc(&gbuf, (size_t)16 * 256 * 256 * 4 * 4); (void)hipMemset(gbuf, 0, (size_t)16 * 256 * 256 * 4 * 4);
    const int iters = 1000, pit = 300;
    run<0, 1>("no partner", wa, out, clk, gbuf, iters, pit);
    run<1, 1>("idle partner (s_sleep)", wa, out, clk, gbuf, iters, pit);
    run<2, 0>("VALU fma x32", wa, out, clk, gbuf, iters, pit);
    run<2, 1>("VALU fma x32", wa, out, clk, gbuf, iters, pit);
    run<2, 1, 1>("VALU fma x32, partner s_setprio 3", wa, out, clk, gbuf, iters, pit);
    run<6, 1, 1>("epilogue-like mix, partner s_setprio 3", wa, out, clk, gbuf, iters, pit);
    run<3, 0>("LDS 8x(ds_write_b64 + ds_read_b128)", wa, out, clk, gbuf, iters, pit);
    run<3, 1>("LDS 8x(ds_write_b64 + ds_read_b128)", wa, out, clk, gbuf, iters, pit);
    run<4, 0>("8x global_load_dwordx4 (L2)", wa, out, clk, gbuf, iters, pit);
    run<4, 1>("8x global_load_dwordx4 (L2)", wa, out, clk, gbuf, iters, pit);
    run<5, 0>("8x global_store_dwordx4", wa, out, clk, gbuf, iters, pit);
    run<5, 1>("8x global_store_dwordx4", wa, out, clk, gbuf, iters, pit);
    run<6, 0>("epilogue-like mix", wa, out, clk, gbuf, iters, pit);
    run<6, 1>("epilogue-like mix", wa, out, clk, gbuf, iters, pit);
    return 0;
}
