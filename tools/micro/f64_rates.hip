// Micro-benchmark: issue rate of the float64 VALU operations the FX kernels are made of, on all SIMDs of the chip (4 waves per SIMD,
// 8 independent chains per lane): clocks per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o f64_rates f64_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP> __global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters) {
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i * 1e-3 + threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) v[i] = fma(v[i], a, b);
                if (OP == 1) v[i] = fmax(v[i] * 1.0000001, b);       // mul + max
                if (OP == 2) v[i] = v[i] + b;
                if (OP == 3) v[i] = v[i] * a;
                if (OP == 4) v[i] = v[i] > b ? v[i] - a : v[i] + a;   // add + add + cmp + 2 cndmask
                if (OP == 5) asm volatile("v_max_f64 %0, %0, %1" : "+v"(v[i]) : "v"(b));
                if (OP == 6) asm volatile("v_min_f64 %0, %0, %1" : "+v"(v[i]) : "v"(b));
                if (OP == 7) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                if (OP == 8) { float f = (float)v[i]; asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f)); v[i] = f; }
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP> void run(const char *name, int n_instr_per_iter) {
    double *out;
    const int blocks = 256 * 4;                       // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipMalloc(&out, blocks * 256 * 8);
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 0.999, 0.5, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 0.999, 0.5, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instr_per_simd = 4.0 * iters * 64.0 * n_instr_per_iter;      // 4 waves x iters x 64 statements x instructions
    printf("%-34s %7.3f ms  -> %6.2f ns per wave-instruction per SIMD (= %5.1f clocks at 2.4 GHz)\n", name, ms,
           ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
    hipFree(out);
}

int main() {
    run<7>("v_fma_f64 (asm)", 1);
    run<5>("v_max_f64 (asm)", 1);
    run<6>("v_min_f64 (asm)", 1);
    run<2>("v_add_f64", 1);
    run<3>("v_mul_f64", 1);
    run<0>("fma (compiler)", 1);
    run<1>("mul + max (compiler, 2 instr)", 2);
    run<4>("cmp + 2 add + 2 cndmask (5 instr)", 5);
    run<8>("cvt + v_fma_f32 + cvt (3 instr)", 3);
    return 0;
}
