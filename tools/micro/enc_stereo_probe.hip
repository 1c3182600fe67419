// Phase clocks of the encoder's fused stereo-block kernel (32 x 2 x 131072 -> 32 x 32768 x 16): thread 0 of every workgroup stamps s_memtime behind
// staging, the first conv, the second conv and the stores (generated copy of csrc/enc_kernels.h: tools/micro/enc_stereo_probe.py).
//   python tools/micro/enc_stereo_probe.py && hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tools/micro/_gen -I music_mixing_style_transfer_amd/csrc -o tools/micro/enc_stereo_probe tools/micro/enc_stereo_probe.hip
#include "enc_kernels_probe.h"

#include <vector>

int main() {
    const int B = 32, L = 131072, Lout = L / 4;
    std::vector<float> hx((size_t)B * 2 * L), hw0(100), hw1(800), hs(16, 0.01f), f0(16 * 64), f1(14 * 64);
    unsigned z = 12345u;
    auto rnd = [&]() { z = z * 1664525u + 1013904223u; return ((z >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto &v : hx) v = rnd();
    for (auto &v : hw0) v = 0.2f * rnd();
    for (auto &v : hw1) v = 0.2f * rnd();
    enc_stereo_pack_a0(hw0.data(), f0.data());
    enc_stereo_pack_a1(hw1.data(), f1.data());
    float *x, *w0, *w1, *sh;
    void *y;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&w0, 4096); hipMalloc(&w1, 3584); hipMalloc(&sh, 64); hipMalloc(&y, (size_t)B * Lout * 16 * 2);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w0, f0.data(), 4096, hipMemcpyHostToDevice);
    hipMemcpy(w1, f1.data(), 3584, hipMemcpyHostToDevice); hipMemcpy(sh, hs.data(), 64, hipMemcpyHostToDevice);
    EncStereoArgs a;
    a.x = x; a.y = y; a.ylo = nullptr; a.a0 = w0; a.shift0 = sh; a.a1 = w1; a.shift1 = sh; a.B = B; a.L = L; a.Lout = Lout;
    a.tiles = (Lout + ENC_STEREO_TO - 1) / ENC_STEREO_TO; a.slope0 = a.slope1 = 0.0f;
    const int grid = B * a.tiles, reps = 20;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(enc_stereo_block_kernel, dim3(grid), dim3(256), 0, 0, a);
    unsigned long long zero[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(stereo_probe), zero, sizeof(zero));
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(enc_stereo_block_kernel, dim3(grid), dim3(256), 0, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long pr[8];
    hipMemcpyFromSymbol(pr, HIP_SYMBOL(stereo_probe), sizeof(pr));
    printf("enc_stereo_block_kernel: %.1f us per launch (with stamps), %d workgroups\n", 1000.0f * ms / reps, grid);
    const char *names[4] = {"staging", "first conv", "mirror fix-up", "second conv + stores"};
    double tot = 0;
    for (int i = 0; i < 4; ++i) tot += (double)pr[i];
    for (int i = 0; i < 4; ++i) printf("  %-22s %9.0f clocks per workgroup (%.0f %%)\n", names[i], (double)pr[i] / ((double)grid * reps), 100.0 * pr[i] / tot);
    printf("  workgroup lifetime %.0f clocks; %d workgroups / 256 CUs = %.1f per CU\n", tot / ((double)grid * reps), grid, grid / 256.0);
    return 0;
}
