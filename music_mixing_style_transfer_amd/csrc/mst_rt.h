// HIP runtime glue for the gfx950 build (hipcc).  The test-only CPU emulator substitutes its own file of
// the same name (tests/emu/mst_rt.h) earlier on the include path; product sources carry no #ifdefs.
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

// hipFFT entry points bound at first use (no link-time dependency); returns false when the library cannot be loaded
#include <dlfcn.h>
static inline bool mst_fft_bind(void **plan_many, void **set_stream, void **exec_r2c, void **exec_c2r, void **destroy) {
    void *lib = nullptr;
    for (const char *name : {"libhipfft.so.0", "libhipfft.so", "/opt/rocm/lib/libhipfft.so"})
        if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) return false;
    *plan_many = dlsym(lib, "hipfftPlanMany");
    *set_stream = dlsym(lib, "hipfftSetStream");
    *exec_r2c = dlsym(lib, "hipfftExecR2C");
    *exec_c2r = dlsym(lib, "hipfftExecC2R");
    *destroy = dlsym(lib, "hipfftDestroy");
    return *plan_many && *set_stream && *exec_r2c && *exec_c2r && *destroy;
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
