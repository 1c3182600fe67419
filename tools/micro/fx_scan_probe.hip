// Timing probe for fx_biquad_scan_kernel: the whole kernel against its load / store shell (MST_SCAN_PROBE=1 skips the levels).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I_gen -I../../music_mixing_style_transfer_amd/csrc [-DMST_SCAN_PROBE=1] -o fx_scan_probe fx_scan_probe.hip
#include "fx_kernels.h"

#include <vector>

int main() {
    const int n_seq = 128, nchunks = 482, SM = 2 * MST_MAX_BANDS;
    double *ends, *starts, *pm;
    hipMalloc(&ends, (size_t)nchunks * SM * n_seq * 8);
    hipMalloc(&starts, (size_t)nchunks * SM * n_seq * 8);
    hipMalloc(&pm, MST_BIQUAD_LEVELS * 256 * 8);
    hipMemset(ends, 0, (size_t)nchunks * SM * n_seq * 8);
    std::vector<double> p(MST_BIQUAD_LEVELS * 256, 0.001);
    hipMemcpy(pm, p.data(), p.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i)
            hipLaunchKernelGGL((fx_biquad_scan_kernel<5, 512>), dim3(n_seq), dim3(512), 0, 0, (const double *)ends, starts, (const double *)pm, n_seq, nchunks);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("probe %d: fx_biquad_scan_kernel<5, 512>, 128 sequences x 482 chunks: %.1f us per launch\n",
#ifdef MST_SCAN_PROBE
               MST_SCAN_PROBE,
#else
               0,
#endif
               ms * 100.0f);
    }
    return 0;
}
