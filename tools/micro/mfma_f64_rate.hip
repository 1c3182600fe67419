// Issue rate of v_mfma_f64_16x16x4_f64 (and of v_fma_f64 for comparison): clocks per instruction per SIMD, one and two waves per SIMD,
// 8 independent accumulators.   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_f64_rate tools/micro/mfma_f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC> __global__ __launch_bounds__(256) void k_mfma(double *out, double a, double b, int iters) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
    const double av = a + threadIdx.x * 1e-9, bv = b + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int wgs_per_cu) {
    double *out;
    const int blocks = 256 * wgs_per_cu;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
    const int iters = 500;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(256), 0, 0, out, 0.5, 0.25, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(256), 0, 0, out, 0.5, 0.25, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_instr = (double)iters * 4 * NACC * wgs_per_cu;      // per SIMD (a workgroup = one wave per SIMD)
    const double flops = (double)blocks * 4 * iters * 4 * NACC * 2048.0;
    printf("v_mfma_f64_16x16x4_f64, %d accumulators, %d wave(s) per SIMD: %.3f ms -> %.1f ns = %.0f clocks at 2.4 GHz per instruction per SIMD; %.1f TFLOP/s\n",
           NACC, wgs_per_cu, ms, ms * 1e6 / n_instr, ms * 1e6 / n_instr * 2.4, flops / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}
int main() {
    run<8>(1); run<8>(2); run<4>(1); run<2>(1); run<1>(1);
    return 0;
}
