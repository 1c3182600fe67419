// Phase clocks of the encoder's fused block-1 kernel (32 x 32768 x 16 -> 32 x 8192 x 32, bf16): thread 0 of every workgroup stamps s_memtime
// (generated copy of csrc/enc_kernels.h: tools/micro/enc_block1_probe.py).
//   python tools/micro/enc_block1_probe.py && hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tools/micro/_gen -I music_mixing_style_transfer_amd/csrc -o tools/micro/enc_block1_probe tools/micro/enc_block1_probe.hip
#include "enc_kernels_probe_b1.h"

#include <vector>

int main() {
    const int B = 32, L = 32768, Lout = L / 4;
    std::vector<__bf16> hx((size_t)B * L * 16), f0(13 * 64 * 8), f1(2 * 13 * 64 * 8);
    std::vector<float> w0(16 * 16 * 25), w1(32 * 16 * 25), hs(32, 0.01f);
    unsigned z = 12345u;
    auto rnd = [&]() { z = z * 1664525u + 1013904223u; return ((z >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto &v : hx) v = (__bf16)rnd();
    for (auto &v : w0) v = 0.1f * rnd();
    for (auto &v : w1) v = 0.1f * rnd();
    enc_block1_pack(w0.data(), 0, 16, 25, f0.data());
    enc_block1_pack(w1.data(), 0, 16, 25, f1.data());
    enc_block1_pack(w1.data(), 16, 16, 25, f1.data() + 13 * 64 * 8);
    void *x, *y, *a0, *a1, *zr;
    float *sh;
    (void)hipMalloc(&x, hx.size() * 2); (void)hipMalloc(&y, (size_t)B * Lout * 32 * 2); (void)hipMalloc(&a0, f0.size() * 2); (void)hipMalloc(&a1, f1.size() * 2);
    (void)hipMalloc(&sh, 128); (void)hipMalloc(&zr, 256); (void)hipMemset(zr, 0, 256);
    (void)hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(a0, f0.data(), f0.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(a1, f1.data(), f1.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(sh, hs.data(), 128, hipMemcpyHostToDevice);
    EncBlock1Args a;
    a.x = (const __bf16 *)x; a.y = (__bf16 *)y; a.a0 = a0; a.a1 = a1; a.shift0 = sh; a.shift1 = sh; a.B = B; a.L = L; a.Lout = Lout;
    a.tiles = (Lout + ENC_B1_TO - 1) / ENC_B1_TO; a.slope0 = a.slope1 = 0.0f; a.zeros = zr;
    const int grid = B * a.tiles, reps = 20;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((enc_block1_fused_kernel<16, 25, 4, 66>), dim3(grid), dim3(256), 0, 0, a);
    unsigned long long zero[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(block1_probe), zero, sizeof(zero));
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((enc_block1_fused_kernel<16, 25, 4, 66>), dim3(grid), dim3(256), 0, 0, a);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long pr[8];
    (void)hipMemcpyFromSymbol(pr, HIP_SYMBOL(block1_probe), sizeof(pr));
    printf("enc_block1_fused_kernel: %.1f us per launch (with stamps), %d workgroups\n", 1000.0f * ms / reps, grid);
    const char *names[4] = {"staging (LDS-DMA)", "first conv", "A1 loads + barrier", "second conv + stores"};
    double tot = 0;
    for (int i = 0; i < 4; ++i) tot += (double)pr[i];
    for (int i = 0; i < 4; ++i) printf("  %-22s %9.0f clocks per workgroup (%.0f %%)\n", names[i], (double)pr[i] / ((double)grid * reps), 100.0 * pr[i] / tot);
    printf("  workgroup lifetime %.0f clocks; %d workgroups / 256 CUs = %.1f per CU\n", tot / ((double)grid * reps), grid, grid / 256.0);
    return 0;
}
