// Device-side vocabulary shared by the gfx950 kernels (wave64, MFMA fragment types, small helpers).
#pragma once
#include <mst_rt.h>

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define MST_WAVE 64
#define MST_LEAKY 0.01f   // torch.nn.LeakyReLU default slope (reference architectures.py:215)

// C/D fragment row of accumulator register `reg` for the 32x32 MFMA shapes (col = lane & 31).
__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// slope < 1: max(v, slope*v) == (v > 0 ? v : slope*v), one multiply + one max (packs into v_pk_mul / v_pk_max)
__device__ __forceinline__ float leaky_relu(float v) { return fmaxf(v, MST_LEAKY * v); }

__device__ __forceinline__ long long mst_clock() { return (long long)__builtin_readcyclecounter(); }

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
