// Micro-benchmark (round 2): who sets the pace of the compressor's chain kernel (fx_comp_chain_kernel) - the walker wave or the wave
// that brings the records into LDS.  Runs the product's own map kernel on 128 sequences x 131072 samples of noise whose level
// wanders around the threshold, then times the map kernel and the chain kernel.  Compile three times:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I_gen -I../../music_mixing_style_transfer_amd/csrc -DMST_CHAIN_PROBE=0 -o fx_chain_p0 fx_chain_variants.hip
//   ... -DMST_CHAIN_PROBE=1 (helper waves alone) ... -DMST_CHAIN_PROBE=2 (walker alone, on the first two batches' entries)
#include "fx_kernels.h"

#include <vector>

template <typename F> float time_us(F launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1000.0f;
}

int main() {
    const int n_seq = 128, C = 2;
    const long L = 131072, nchunks = L / MST_COMP_T;
    std::vector<float> x((size_t)L * n_seq);                // [64 items][L][2]: noise whose level wanders around the threshold
    unsigned s = 12345u;
    for (size_t i = 0; i < x.size(); ++i) {
        s = s * 1664525u + 1013904223u;
        const float u = (s >> 8) / 16777216.0f - 0.5f;
        const float env = 0.02f + 0.4f * (0.5f + 0.5f * sinf((float)((i / 2) % L) * 3.0e-4f));
        x[i] = u * env;
    }
    CompArgs ca;
    CompMapArgs m;
    float *dx;
    double *dmaps, *dys, *dtab;
    hipMalloc(&dx, x.size() * 4);
    hipMalloc(&dmaps, (size_t)n_seq * nchunks * MST_COMP_REC * 8);
    hipMalloc(&dys, (size_t)n_seq * nchunks * 8);
    hipMalloc(&dtab, 256 * 8);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(fx_log10_table_kernel, dim3(1), dim3(128), 0, 0, dtab);
    ca.x = dx; ca.y = nullptr; ca.n_seq = n_seq; ca.C = C; ca.L = L; ca.threshold = -20.0; ca.ratio = 4.0; ca.makeup = 0.0;
    m.aA = exp(-1.0 / (0.005 * 44100.0)); m.aR = exp(-1.0 / (0.05 * 44100.0)); m.use_min = 0;
    ca.alpha_att = m.aA; ca.alpha_rel = m.aR;
    m.log_tab = dtab; m.maps = dmaps; m.ystart = dys; m.n_seq = n_seq; m.nchunks = (int)nchunks; m.L = L;
    for (int p = 0; p < MST_COMP_NP; ++p) {
        m.slope[0][p] = m.slope[1][p] = pow(m.aA, MST_COMP_T - p) * pow(m.aR, p);
        m.inv_slope[0][p] = m.inv_slope[1][p] = 1.0 / m.slope[0][p];
    }
    const dim3 cg((unsigned)nchunks, (n_seq + 63) / 64);
    const float t_map = time_us([&] { hipLaunchKernelGGL(fx_comp_map_kernel<false>, cg, dim3(64), 0, 0, m, ca); }, 5);
    const float t_chain = time_us([&] { hipLaunchKernelGGL(fx_comp_chain_kernel, dim3(n_seq), dim3(MST_CHAIN_THREADS), 0, 0, m); }, 5);
    printf("probe %d  (0 = product, 1 = helper waves alone, 2 = walker alone), 128 sequences x %ld chunks: map kernel %.1f us, chain kernel %.1f us\n",
           MST_CHAIN_PROBE, nchunks, t_map, t_chain);
    std::vector<double> ys(8);
    hipMemcpy(ys.data(), dys + (size_t)(nchunks - 1) * n_seq, 64, hipMemcpyDeviceToHost);
    printf("  last chunk start values: %.9f %.9f %.9f\n", ys[0], ys[1], ys[2]);
    return 0;
}
