// The equaliser's apply pass (fx_biquad_chunk_kernel<true, 5>, the product's kernel) at several chunk lengths, and a copy whose 25 coefficients sit in
// VECTOR registers instead of scalar pairs: is the pass bound by the issue of float64 instructions with 64-bit scalar operands?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I music_mixing_style_transfer_amd/csrc -o tools/micro/eq_apply_variants tools/micro/eq_apply_variants.hip
#include "mst_dev.h"
#include "fx_kernels.h"

#include <vector>

template <int NBANDS>
__global__ __launch_bounds__(64) void eq_apply_vgpr_kernel(BiquadChunkArgs a) {          // stereo only, full batches only (M a multiple of 16)
    const long gid = (long)blockIdx.x * 64 + threadIdx.x;
    if (gid >= (long)a.n_seq * a.nchunks) return;
    const int c = (int)(gid % 2);
    const int k = (int)((gid / 2) % a.nchunks);
    const int item = (int)(gid / (2L * a.nchunks));
    const int seq = item * 2 + c;
    const long n_lo = (long)k * a.M;
    const float *xp = a.x + (size_t)item * a.L * 2 + c;
    double z1[NBANDS], z2[NBANDS], cf[NBANDS][5];
#pragma unroll
    for (int b = 0; b < NBANDS; ++b) {
        const double2 zz = *(const double2 *)(a.starts + ((size_t)seq * a.nchunks + k) * (2 * MST_MAX_BANDS) + 2 * b);
        z1[b] = zz.x;
        z2[b] = zz.y;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            cf[b][i] = a.coef[b][i];
            asm volatile("" : "+v"(cf[b][i]));
        }
    }
    double ss = 0.0;
    constexpr int NB = 16;
    for (long bt = 0; bt < a.M / NB; ++bt) {
        float xin[NB], o[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) xin[i] = xp[(n_lo + bt * NB + i) * 2];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            double v = (double)xin[i];
#pragma unroll
            for (int b = 0; b < NBANDS; ++b) {
                const double yn = cf[b][0] * v + z1[b];
                z1[b] = cf[b][1] * v - cf[b][3] * yn + z2[b];
                z2[b] = cf[b][2] * v - cf[b][4] * yn;
                v = yn;
            }
            o[i] = (float)v;
            ss += (double)o[i] * (double)o[i];
        }
        const bool odd = c != 0;
        float *fp = a.y + ((size_t)item * a.L + n_lo + bt * NB) * 2;
#pragma unroll
        for (int f = 0; f < NB; f += 4) {
            const float ta = mst_lane_swap(odd ? o[f] : o[f + 2]), tb = mst_lane_swap(odd ? o[f + 1] : o[f + 3]);
            const float4 v = odd ? make_float4(ta, o[f + 2], tb, o[f + 3]) : make_float4(o[f], ta, o[f + 1], tb);
            *(float4 *)(fp + (f + (odd ? 2 : 0)) * 2) = v;
        }
    }
    if (a.out_sumsq) atomicAdd(&a.out_sumsq[item * MST_SUMSQ_SLOTS + (k & (MST_SUMSQ_SLOTS - 1))], ss);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int items = 64;
    const long L = 131072;
    float *x, *y;
    double *st, *ssq;
    if (hipMalloc(&x, items * L * 2 * 4 + 4096) != hipSuccess || hipMalloc(&y, items * L * 2 * 4 + 4096) != hipSuccess) { printf("malloc failed\n"); return 1; } (void)hipMalloc(&ssq, items * MST_SUMSQ_SLOTS * 8);
    if (hipMalloc(&st, (size_t)items * 2 * 4096 * 16 * 8) != hipSuccess) { printf("malloc failed\n"); return 1; }
    (void)hipMemset(x, 0, items * L * 2 * 4); (void)hipMemset(st, 0, (size_t)items * 2 * 4096 * 16 * 8); (void)hipMemset(ssq, 0, items * MST_SUMSQ_SLOTS * 8);
    {
        std::vector<float> h((size_t)items * L * 2);
        unsigned z = 1u;
        for (auto &v : h) { z = z * 1664525u + 1013904223u; v = ((z >> 8) & 0xffff) / 65536.0f - 0.5f; }
        (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int M : {1024, 512, 272, 256, 144, 128, 64}) {
        BiquadChunkArgs a;
        a.x = x; a.y = y; a.ends = nullptr; a.starts = st; a.n_seq = items * 2; a.C = 2; a.M = M; a.L = L; a.nchunks = (int)((L + M - 1) / M); a.n_bands = 5;
        for (int b = 0; b < MST_MAX_BANDS; ++b) { a.coef[b][0] = 0.9 + 0.01 * b; a.coef[b][1] = -1.7; a.coef[b][2] = 0.8; a.coef[b][3] = -1.8 + 0.01 * b; a.coef[b][4] = 0.85; }
        a.out_sumsq = ssq;
        const long lanes = (long)a.n_seq * a.nchunks;
        const dim3 g((unsigned)((lanes + 63) / 64));
        for (int var = 0; var < 3; ++var) {
            float best = 1e9f;
            for (int r = 0; r < 5; ++r) {
                (void)hipEventRecord(e0);
                if (var == 0) hipLaunchKernelGGL((fx_biquad_chunk_kernel<true, 5>), g, dim3(64), 0, 0, a);
                else if (var == 1) hipLaunchKernelGGL((eq_apply_vgpr_kernel<5>), g, dim3(64), 0, 0, a);
                else hipLaunchKernelGGL((fx_biquad_stereo_apply_kernel<5>), dim3((unsigned)((lanes / 2 + 127) / 128)), dim3(256), 0, 0, a);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("M = %4d (%6ld lanes, %5.2f waves per SIMD)  %s  %7.1f us\n", M, lanes, lanes / 64.0 / 1024.0, var == 2 ? "slabs through LDS      " : var ? "coefficients in VGPRs " : "lane per chunk (SGPRs)", best * 1000.0f);
        }
    }
    return 0;
}
