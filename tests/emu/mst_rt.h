// TEST INFRASTRUCTURE ONLY - a CPU SIMT emulator standing in for <hip/hip_runtime.h>.
//
// The product sources under music_mixing_style_transfer_amd/csrc include "mst_rt.h"; the product build
// (hipcc, gfx950) finds csrc/mst_rt.h, the emulator build (host clang++) puts THIS directory first on
// the include path.  Kernels then run unmodified on the host: every HIP thread is a fiber, 64 fibers
// form a wavefront, __syncthreads / MFMA / shuffles are fiber rendezvous points.  This exists because
// the build container has no GPU and GPU box minutes are scarce: index math, MFMA fragment layouts,
// LDS swizzles and the C-ABI host logic are all checked here first (tests/test_emu_*.py).
// It is never loaded by the product (music_mixing_style_transfer_amd/_lib.py loads libmst_hip.so only).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define MST_EMULATED 1
#define MST_HEAVY_UNROLL __attribute__((optnone))      // host clang needs > 10 minutes at -O2 for the unrolled compressor map kernel
#define MST_WAVES_PER_SIMD(n)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };
extern thread_local emu_uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

typedef int hipError_t;
typedef void *hipStream_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
static inline hipError_t hipMalloc(void **p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256); return *p ? 0 : 2; }
static inline hipError_t hipFree(void *p) { free(p); return 0; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
typedef struct emu_event { double t; } *hipEvent_t;
double emu_now_ms();
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event{0.0}; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = emu_now_ms(); return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
// streams are synchronous here: a side stream is the same thing as the caller's, waiting for an event is a no-op
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
static inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return 0; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = nullptr; return 0; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new emu_event{0.0}; return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return 0; }

static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
struct float2 { float x, y; };
struct double2 { double x, y; };
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float mst_fmax(float a, float b) { return fmaxf(a, b); }
#define MST_NO_CONTRACT(v) asm volatile("" : "+x"(v))      // value barrier: no fp contraction across it

namespace emu {
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
void block_barrier();
// wave rendezvous with a 64 x 64-byte exchange area (double buffered internally)
unsigned char *wave_publish(const void *src, size_t bytes);   // returns pointer to slot[lane 0]; stride 64 B
int lane_id();
}  // namespace emu

#define MST_LAUNCH(kern, grid, block, stream, ...) \
    emu::launch((grid), (block), [&]() { kern(__VA_ARGS__); })

static inline void __syncthreads() { emu::block_barrier(); }
static inline void __threadfence() {}          // one OS thread runs the workgroups one after the other: every store is visible to the next

// ---- wave collectives ------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float emu_f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 emu_bf16x8;

// v_mfma_f32_32x32x2_f32: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
// D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5); exact k-ordered fmaf chain (guide section 3).
static inline emu_f32x16 emu_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c) {
    struct AB { float a, b; } me = {a, b};
    const unsigned char *base = emu::wave_publish(&me, sizeof(me));
    const int l = emu::lane_id(), col = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            const AB *pa = (const AB *)(base + 64 * (row + 32 * k));
            const AB *pb = (const AB *)(base + 64 * (col + 32 * k));
            acc = fmaf(pa->a, pb->b, acc);
        }
        c[r] = acc;
    }
    return c;
}

// v_mfma_f32_32x32x16_bf16: lane l holds 8 consecutive k of A row (l&31) / B column (l&31),
// k group = l>>5; fp32 accumulate.
static inline emu_f32x16 emu_mfma_f32_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c) {
    struct AB { emu_bf16x8 a, b; } me = {a, b};
    const unsigned char *base = emu::wave_publish(&me, sizeof(me));
    const int l = emu::lane_id(), col = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        double acc = 0.0;
        for (int g = 0; g < 2; ++g) {
            const AB *pa = (const AB *)(base + 64 * (row + 32 * g));
            const AB *pb = (const AB *)(base + 64 * (col + 32 * g));
            for (int e = 0; e < 8; ++e) acc += (double)(float)pa->a[e] * (double)(float)pb->b[e];
        }
        c[r] = (float)((double)c[r] + acc);
    }
    return c;
}
// v_mfma_f32_16x16x32_bf16: lane l holds 8 consecutive k of A row (l&15) / B column (l&15), k group = l>>4;
// D: col = l&15, row = 4*(l>>4) + reg; fp32 accumulate.
typedef __attribute__((ext_vector_type(4))) float emu_f32x4;
static inline emu_f32x4 emu_mfma_f32_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c) {
    struct AB { emu_bf16x8 a, b; } me = {a, b};
    const unsigned char *base = emu::wave_publish(&me, sizeof(me));
    const int l = emu::lane_id(), col = l & 15, hi = l >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * hi + r;
        double acc = 0.0;
        for (int kg = 0; kg < 4; ++kg) {
            const AB *pa = (const AB *)(base + 64 * (row + 16 * kg));
            const AB *pb = (const AB *)(base + 64 * (col + 16 * kg));
            for (int e = 0; e < 8; ++e) acc += (double)(float)pa->a[e] * (double)(float)pb->b[e];
        }
        c[r] = (float)((double)c[r] + acc);
    }
    return c;
}
// v_mfma_f32_16x16x4_f32: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; D: col = l&15, row = 4*(l>>4) + reg; exact k-ordered fmaf chain
static inline emu_f32x4 emu_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c) {
    struct AB { float a, b; } me = {a, b};
    const unsigned char *base = emu::wave_publish(&me, sizeof(me));
    const int l = emu::lane_id(), col = l & 15, hi = l >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * hi + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            const AB *pa = (const AB *)(base + 64 * (row + 16 * k));
            const AB *pb = (const AB *)(base + 64 * (col + 16 * k));
            acc = fmaf(pa->a, pb->b, acc);
        }
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu_mfma_f32_16x16x4f32((a), (b), (c))
// v_mfma_f64_16x16x4_f64: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; D: col = l&15, row = (l>>4) + 4*reg (NOT the f32 map); k-ordered fma chain
typedef __attribute__((ext_vector_type(4))) double emu_f64x4;
static inline emu_f64x4 emu_mfma_f64_16x16x4f64(double a, double b, emu_f64x4 c) {
    struct AB { double a, b; } me = {a, b};
    const unsigned char *base = emu::wave_publish(&me, sizeof(me));
    const int l = emu::lane_id(), col = l & 15, hi = l >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = hi + 4 * r;
        double acc = c[r];
        for (int k = 0; k < 4; ++k) {
            const AB *pa = (const AB *)(base + 64 * (row + 16 * k));
            const AB *pb = (const AB *)(base + 64 * (col + 16 * k));
            acc = fma(pa->a, pb->b, acc);
        }
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) emu_mfma_f64_16x16x4f64((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu_mfma_f32_16x16x32_bf16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma_f32_32x32x2f32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu_mfma_f32_32x32x16_bf16((a), (b), (c))

template <typename T> static inline T emu_shfl(T v, int src) {
    const unsigned char *base = emu::wave_publish(&v, sizeof(T));
    T out;
    memcpy(&out, base + 64 * (src & 63), sizeof(T));
    return out;
}
template <typename T> static inline T __shfl_xor(T v, int mask, int = 64) { return emu_shfl(v, emu::lane_id() ^ mask); }
static inline unsigned long long mst_wave_ballot(bool p) {
    unsigned long long m = 0;
    const int mine = p ? 1 : 0;
    const unsigned char *base = emu::wave_publish(&mine, sizeof(int));
    for (int l = 0; l < 64; ++l) { int v; memcpy(&v, base + 64 * l, sizeof(int)); m |= (unsigned long long)(v & 1) << l; }
    return m;
}
struct MstUniformF64 {
    double v;
    double value() const { return v; }
};
static inline MstUniformF64 mst_wave_read_u64(double v, int src) { return MstUniformF64{emu_shfl(v, src)}; }
static inline MstUniformF64 mst_wave_uniform(MstUniformF64 y) { return y; }
static inline float mst_lane_swap(float v) { return emu_shfl(v, emu::lane_id() ^ 1); }
static inline double mst_half_prefix_sum_f64(double v) {
    const int l = emu::lane_id();
    const unsigned char *base = emu::wave_publish(&v, sizeof(double));      // one rendezvous, then every lane sums its own prefix
    double s = 0.0;
    for (int j = l & 32; j <= l; ++j) { double t; memcpy(&t, base + 64 * j, sizeof(double)); s += t; }
    return s;
}
static inline MstUniformF64 mst_wave_first_ge(MstUniformF64 y, double u, double r) {
    const unsigned long long m = mst_wave_ballot(y.v >= u);
    return MstUniformF64{emu_shfl(r, m ? __builtin_ctzll(m) : 0)};
}
template <int DST> static inline double mst_wave_park_f64(double keep, MstUniformF64 y) { return emu::lane_id() == DST ? y.v : keep; }
template <typename T> static inline T __shfl_down(T v, int d, int = 64) {
    const int l = emu::lane_id();
    return emu_shfl(v, l + d < 64 ? l + d : l);
}
template <typename T> static inline T __shfl(T v, int src, int = 64) { return emu_shfl(v, src); }
template <typename T> static inline T __shfl_up(T v, int d, int = 64) {
    const int l = emu::lane_id();
    return emu_shfl(v, l - d >= 0 ? l - d : l);
}
// LDS-DMA stand-ins: the copy happens at issue (the emulator runs one fiber at a time), the wait is the workgroup barrier
// (the wave meets first: on the GPU a wave's earlier LDS reads have been issued by ALL its lanes before the copy is - the emulator runs the lanes one by one)
static inline void mst_dma16(const void *gsrc, void *lds_wave_base) {
    (void)emu_shfl(0, 0);
    memcpy((unsigned char *)lds_wave_base + 16 * emu::lane_id(), gsrc, 16);
}
template <int N> static inline void mst_dma_wait_barrier() { __syncthreads(); }
static inline void mst_wave_lds_fence() { (void)emu_shfl(0, 0); }      // the lanes of the wave meet
static inline void mst_row_swap(unsigned &a, unsigned &b) {      // v_permlane16_swap_b32: rows 1, 3 of a <-> rows 0, 2 of b
    struct AB { unsigned a, b; } me = {a, b};
    const unsigned char *base = emu::wave_publish(&me, sizeof(me));
    const int l = emu::lane_id();
    if ((l >> 4) & 1) a = ((const AB *)(base + 64 * (l - 16)))->b;
    else b = ((const AB *)(base + 64 * (l + 16)))->a;
}
typedef __attribute__((ext_vector_type(4))) unsigned mst_u32x4;
struct MstStream16 { const unsigned char *base; unsigned bytes; };
static inline MstStream16 mst_stream16(const void *base, unsigned bytes) { return MstStream16{(const unsigned char *)base, bytes}; }
static inline mst_u32x4 mst_stream_load16(MstStream16 s, unsigned voffset, unsigned soffset) {
    mst_u32x4 v = {0u, 0u, 0u, 0u};          // a raw buffer load beyond num_records returns zeros
    const unsigned long off = (unsigned long)voffset + soffset;
    if (off + 16 <= s.bytes) memcpy(&v, s.base + off, 16);
    return v;
}
static inline unsigned mst_stream_load4(MstStream16 s, unsigned voffset, unsigned soffset) {
    unsigned v = 0u;
    const unsigned long off = (unsigned long)voffset + soffset;
    if (off + 4 <= s.bytes) memcpy(&v, s.base + off, 4);
    return v;
}
static inline int mst_mul24(int a, int b) { return a * b; }
static inline long long mst_realtime() { return (long long)(emu_now_ms() * 1e5); }      // 100 MHz ticks
static inline int mst_current_device() { return 0; }
static inline int mst_num_cus() { return 4; }      // a small persistent grid: every workgroup walks several tiles
#define __builtin_amdgcn_readfirstlane(v) emu_shfl((v), 0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_s_barrier() __syncthreads()
#define __builtin_amdgcn_wave_barrier() ((void)emu_shfl(0, 0))       // lanes of a wave meet (the emulator runs them one by one)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)

static inline float atomicAdd(float *p, float v) {
    uint32_t o = __atomic_load_n((uint32_t *)p, __ATOMIC_RELAXED), n;
    float old, neu;
    do {
        memcpy(&old, &o, 4);
        neu = old + v;
        memcpy(&n, &neu, 4);
    } while (!__atomic_compare_exchange_n((uint32_t *)p, &o, n, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return old;
}
static inline double atomicAdd(double *p, double v) {
    uint64_t o = __atomic_load_n((uint64_t *)p, __ATOMIC_RELAXED), n;
    double old, neu;
    do {
        memcpy(&old, &o, 8);
        neu = old + v;
        memcpy(&n, &neu, 8);
    } while (!__atomic_compare_exchange_n((uint64_t *)p, &o, n, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return old;
}
static inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i); return r; }
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
