// HIP runtime glue for the gfx950 build (hipcc).  The test-only CPU emulator substitutes its own file of
// the same name (tests/emu/mst_rt.h) earlier on the include path; product sources carry no #ifdefs (the timing probes of
// tools/micro are patched into a generated copy by tools/micro/probe_patch.py).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// marks a kernel whose fully unrolled body a HOST compiler cannot optimise in reasonable time (empty here; the emulator's mst_rt.h
// turns it into an attribute for its own build)
#define MST_HEAVY_UNROLL
// register budget of a kernel: exactly n waves per SIMD (empty in the emulator)
#define MST_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))

#define MST_LAUNCH(kern, grid, block, stream, ...) \
    hipLaunchKernelGGL(kern, (grid), (block), 0, (hipStream_t)(stream), __VA_ARGS__)

// value barrier: the compiler may not fuse the operation that produced `v` with the one that consumes it (fp contraction)
#define MST_NO_CONTRACT(v) asm volatile("" : "+v"(v))

// plain v_max_f32 (fmaxf() on a raw MFMA result costs a second v_max that only quiets NaNs)
__device__ __forceinline__ float mst_fmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- LDS-DMA (global_load_lds_dwordx4): the 64 lanes of the calling wave copy 16 bytes each from their OWN global address straight into
// LDS at lds_wave_base + 16 * lane (the destination is lane-linear: a swizzled LDS image is made by permuting the SOURCE addresses).
// No VGPR round trip, no ds_write; the copy is asynchronous and counted on vmcnt, in order with the wave's other loads.  hipcc does not
// know about it (inline asm): its own counted waits stay counted (never a vmcnt(0) drain), and the data is retired by
// mst_dma_wait_barrier<N>() - N = loads the wave has issued AFTER the copy that may still be in flight - followed by the barrier every
// reader has to pass.  M0 (the LDS base register of the instruction) is saved and restored inside the statement.
__device__ __forceinline__ void mst_dma16(const void *gsrc, void *lds_wave_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);      // low 32 bits of a flat LDS address = LDS offset
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}
// wait until at most N of this wave's vector-memory operations are outstanding (in-order counter: everything older has landed) and all
// its LDS operations have returned, then the workgroup barrier - a raw s_barrier: __syncthreads() would drain vmcnt to 0
template <int N> __device__ __forceinline__ void mst_dma_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// this wave's LDS stores have completed and all its lanes have passed this point: what one lane wrote, another lane of the wave may read
__device__ __forceinline__ void mst_wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// v_permlane16_swap_b32: rows (16 lanes) 1 and 3 of `a` trade places with rows 0 and 2 of `b`
typedef __attribute__((ext_vector_type(2))) unsigned mst_u32x2;
__device__ __forceinline__ void mst_row_swap(unsigned &a, unsigned &b) {
    const mst_u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
// a read-only array streamed with buffer loads: address = base + voffset (per lane, VGPR) + soffset (wave-uniform, SGPR) - the uniform part
// of the address never occupies vector registers (hipcc otherwise keeps one 64-bit VGPR address per unrolled load and spills them)
typedef __attribute__((ext_vector_type(4))) unsigned mst_u32x4;
struct MstStream16 {
    __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ MstStream16 mst_stream16(const void *base, unsigned bytes) {
    return MstStream16{__builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000)};
}
__device__ __forceinline__ mst_u32x4 mst_stream_load16(MstStream16 s, unsigned voffset, unsigned soffset) {
    return __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, (int)voffset, (int)soffset, 0);
}
__device__ __forceinline__ unsigned mst_stream_load4(MstStream16 s, unsigned voffset, unsigned soffset) {      // one dword, zeros beyond the descriptor
    return __builtin_amdgcn_raw_buffer_load_b32(s.rsrc, (int)voffset, (int)soffset, 0);
}
// the constant 100 MHz real-time counter (s_memrealtime) - against mst_clock() (shader clocks) it gives the clock a kernel ran at
__device__ __forceinline__ long long mst_realtime() { return (long long)__builtin_amdgcn_s_memrealtime(); }
// a * b for operands known to fit 24 bits: one full-rate instruction (v_mul_lo_u32 runs at a quarter of the rate)
__device__ __forceinline__ int mst_mul24(int a, int b) { return __mul24(a, b); }
// the device the calling thread is bound to (per-device constants), -1 on error
static inline int mst_current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess ? dev : -1;
}
// compute units of the current device (grid size of the persistent kernels)
static inline int mst_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

// a wave-uniform double as the two SGPR halves v_readlane returns (kept apart so that v_writelane can take them without a copy)
struct MstUniformF64 {
    int lo, hi;
    __device__ __forceinline__ double value() const { return __hiloint2double(hi, lo); }
};
__device__ __forceinline__ MstUniformF64 mst_wave_uniform(MstUniformF64 y) {      // re-assert uniformity after divergent control flow
    return MstUniformF64{__builtin_amdgcn_readfirstlane(y.lo), __builtin_amdgcn_readfirstlane(y.hi)};
}
__device__ __forceinline__ MstUniformF64 mst_wave_read_u64(double v, int src) {
    return MstUniformF64{__builtin_amdgcn_readlane(__double2loint(v), src), __builtin_amdgcn_readlane(__double2hiint(v), src)};
}
// the value of the neighbouring lane (lane ^ 1): one DPP move (quad_perm [1, 0, 3, 2])
__device__ __forceinline__ float mst_lane_swap(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
}
// inclusive prefix sum over each 32-lane half of the wave (lanes 0..31 and 32..63 separately): four row_shr steps inside the 16-lane
// rows, then row_bcast:15 carries the first row's total into the second row of each half.  DPP moves: VALU speed, no LDS crossbar.
__device__ __forceinline__ double mst_half_prefix_sum_f64(double v) {
#define MST_SCAN_STEP(CTRL, ROWS)                                                                          \
    {                                                                                                      \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, false);          \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xf, false);          \
        v += __hiloint2double(hi, lo);                       /* lanes without a source add 0 */           \
    }
    MST_SCAN_STEP(0x111, 0xf)      // row_shr:1
    MST_SCAN_STEP(0x112, 0xf)      // row_shr:2
    MST_SCAN_STEP(0x114, 0xf)      // row_shr:4
    MST_SCAN_STEP(0x118, 0xf)      // row_shr:8
    MST_SCAN_STEP(0x142, 0xa)      // row_bcast:15 into rows 1 and 3
#undef MST_SCAN_STEP
    return v;
}
// r of the LOWEST lane whose u <= y (y wave-uniform; at least one lane must pass; ALL 64 lanes of the calling wave must be active - EXEC
// is put back to all ones): v_cmpx writes the compare straight into EXEC and v_readfirstlane picks the first lane left - no trip through
// the scalar unit (v_cmp -> s_bcnt1 -> v_readlane) on a dependent chain.  Wait states by hand (the hazard recogniser does not look
// inside): VALU-written EXEC -> v_readfirstlane 4; VALU-written SGPR -> VALU read 2 (the trailing s_nop; callers keep two instructions
// between two calls of a chain, which the fma that produces r and the bookkeeping around it always are).
__device__ __forceinline__ MstUniformF64 mst_wave_first_ge(MstUniformF64 y, double u, double r) {
    MstUniformF64 o;
    const double yd = y.value();
    const int rlo = __double2loint(r), rhi = __double2hiint(r);
    asm volatile("v_cmpx_ge_f64_e32 vcc, %[y], %[u]\n\t"
                 "s_nop 3\n\t"
                 "v_readfirstlane_b32 %[olo], %[rlo]\n\t"
                 "v_readfirstlane_b32 %[ohi], %[rhi]\n\t"
                 "s_mov_b64 exec, -1\n\t"
                 "s_nop 0"
                 : [olo] "=&s"(o.lo), [ohi] "=&s"(o.hi)
                 : [y] "s"(yd), [u] "v"(u), [rlo] "v"(rlo), [rhi] "v"(rhi)
                 : "vcc");
    return o;
}
// the wave-uniform y written into lane `dst` of keep (v_writelane_b32 x 2: no compare, no exec change)
template <int DST> __device__ __forceinline__ double mst_wave_park_f64(double keep, MstUniformF64 y) {
    int lo = __double2loint(keep), hi = __double2hiint(keep);      // the lane select is an inline constant: one SGPR per instruction
    asm("v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"(y.lo), "n"(DST));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"(y.hi), "n"(DST));
    return __hiloint2double(hi, lo);
}
