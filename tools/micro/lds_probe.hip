// Probe (round 6): does a co-resident workgroup of ANOTHER kernel ever change this workgroup's LDS or registers?  Every workgroup fills its
// LDS with a pattern of its own, keeps a few registers of known content, and re-checks both `iters` times; the first mismatches go to `out`.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/_ab/lds_probe.so tools/micro/lds_probe.hip
#include <hip/hip_runtime.h>
template <int WORDS>
__global__ __launch_bounds__(256) void lds_probe_kernel(unsigned *out, int iters) {
    __shared__ unsigned lds[WORDS];
    const unsigned pat = 0xA5000000u | (blockIdx.x << 8);
    for (int i = threadIdx.x; i < WORDS; i += 256) lds[i] = pat ^ (unsigned)i;
    unsigned r0 = pat + threadIdx.x, r1 = ~r0, r2 = r0 * 3u;
    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2));
    __syncthreads();
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    for (int it = 0; it < iters; ++it) {
        // rewrite the pattern with 128-bit stores, read it back with 128-bit and 64-bit loads (the block kernels' LDS instructions)
        for (int i = threadIdx.x * 4; i < WORDS; i += 1024) {
            u4 w = {pat ^ (unsigned)i, pat ^ (unsigned)(i + 1), pat ^ (unsigned)(i + 2), pat ^ (unsigned)(i + 3)};
            *(u4 *)(lds + i) = w;
        }
        __syncthreads();
        for (int i = ((threadIdx.x + 7 * it) & 255) * 4; i < WORDS; i += 1024) {
            const u4 w = *(const u4 *)(lds + i);
            const u2 h = *(const u2 *)(lds + i + 2);
            if (w[0] != (pat ^ (unsigned)i) || w[1] != (pat ^ (unsigned)(i + 1)) || w[2] != (pat ^ (unsigned)(i + 2)) || w[3] != (pat ^ (unsigned)(i + 3)) ||
                h[0] != w[2] || h[1] != w[3])
                atomicAdd(out + 2, 1u);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < WORDS; i += 256) {
            const unsigned v = lds[i];
            if (v != (pat ^ (unsigned)i)) {
                const unsigned slot = atomicAdd(out, 1u);
                if (slot < 200) { out[4 + 4 * slot] = blockIdx.x; out[5 + 4 * slot] = (unsigned)i; out[6 + 4 * slot] = v; out[7 + 4 * slot] = (unsigned)it; }
                lds[i] = pat ^ (unsigned)i;
            }
        }
        asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2));
        if (r0 != pat + threadIdx.x || r1 != ~r0 || r2 != r0 * 3u) atomicAdd(out + 1, 1u);
        __builtin_amdgcn_s_sleep(20);
    }
}
extern "C" int lds_probe_launch(void *stream, int n_wg, int iters, unsigned *out_dev, int kb) {
    if (kb == 16) hipLaunchKernelGGL(lds_probe_kernel<4096>, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, out_dev, iters);
    else if (kb == 40) hipLaunchKernelGGL(lds_probe_kernel<10240>, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, out_dev, iters);
    else hipLaunchKernelGGL(lds_probe_kernel<2048>, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, out_dev, iters);
    return (int)hipGetLastError();
}
