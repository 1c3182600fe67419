// Device-side vocabulary shared by the gfx950 kernels (wave64, MFMA fragment types, small helpers).
#pragma once
#include <mst_rt.h>

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define MST_WAVE 64
#define MST_LEAKY 0.01f   // torch.nn.LeakyReLU default slope (reference architectures.py:215)

// C/D fragment row of accumulator register `reg` for the 32x32 MFMA shapes (col = lane & 31).
__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// slope < 1: max(v, slope*v) == (v > 0 ? v : slope*v), one multiply + one max (packs into v_pk_mul / v_pk_max)
__device__ __forceinline__ float leaky_relu(float v) { return fmaxf(v, MST_LEAKY * v); }

// TCN block epilogue for 4 consecutive channels of one output time: y = r * LeakyReLU(v) + (b + s * x) rounded to bf16, where v is
// the accumulator (dilated conv + BN shift), r/b the FiLM pair and s the residual scale (reference architectures.py:225-233).
// Written on float pairs so that it compiles to v_pk_mul_f32 / v_pk_fma_f32: 4 VALU instructions per element.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x4 tcn_epilogue4(const float *v, f32x4 fr, f32x4 fb, f32x4 rs, bf16x4 xin) {
    bf16x4 out;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        f32x2 vv = {v[2 * p], v[2 * p + 1]};
        const f32x2 t = vv * MST_LEAKY;
        vv = f32x2{mst_fmax(vv.x, t.x), mst_fmax(vv.y, t.y)};
        const f32x2 xx = {(float)xin[2 * p], (float)xin[2 * p + 1]};
        const f32x2 c = f32x2{rs[2 * p], rs[2 * p + 1]} * xx + f32x2{fb[2 * p], fb[2 * p + 1]};
        const f32x2 y = f32x2{fr[2 * p], fr[2 * p + 1]} * vv + c;
        out[2 * p] = (__bf16)y.x;
        out[2 * p + 1] = (__bf16)y.y;
    }
    return out;
}

__device__ __forceinline__ long long mst_clock() { return (long long)__builtin_readcyclecounter(); }      // s_memtime: shader clocks

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// 8 consecutive channels of one NLC row (16-byte aligned): vector load / store with fp32 <-> storage conversion
__device__ __forceinline__ void load8(const float *p, float (&v)[8]) {
    const f32x4 a = *(const f32x4 *)p, b = *(const f32x4 *)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[i + 4] = b[i]; }
}
__device__ __forceinline__ void load8(const __bf16 *p, float (&v)[8]) {
    const bf16x8 a = *(const bf16x8 *)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void store8(float *p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[i + 4]; }
    *(f32x4 *)p = a;
    *(f32x4 *)(p + 4) = b;
}
__device__ __forceinline__ void store8(__bf16 *p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)v[i];
    *(bf16x8 *)p = a;
}
