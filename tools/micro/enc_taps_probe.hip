// Phase clocks of enc_conv_taps_kernel<5, 1> on the encoder's 2048 -> 2048 layer (32 items x 32 steps = 1024 columns, k = 5, 16 channel tiles, 4 k-slices = 256 workgroups):
// clocks of matrix wave 0 inside its taps against clocks at the per-block barrier (generated copy: tools/micro/enc_taps_probe.py).
#include "enc_kernels_probe_taps.h"
#include <vector>
int main() {
    const int B = 32, L = 32, C = 2048, K = 5, nblk = C / 64, nchunks = K * nblk, S = 4;
    const size_t wfr = (size_t)(C / 128) * nchunks * 2 * 8 * 64 * 8;
    std::vector<__bf16> hx((size_t)B * L * C), hw(wfr);
    unsigned z = 7u;
    auto rnd = [&]() { z = z * 1664525u + 1013904223u; return ((z >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto &v : hx) v = (__bf16)rnd();
    for (auto &v : hw) v = (__bf16)(0.02f * rnd());
    void *x, *y, *wp, *part, *zr;
    float *sh;
    (void)hipMalloc(&x, hx.size() * 2); (void)hipMalloc(&y, hx.size() * 2); (void)hipMalloc(&wp, hw.size() * 2); (void)hipMalloc(&part, (size_t)S * B * L * C * 4);
    (void)hipMalloc(&sh, C * 4); (void)hipMalloc(&zr, 256); (void)hipMemset(zr, 0, 256); (void)hipMemset(sh, 0, C * 4);
    (void)hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(wp, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    EncTapsArgs a;
    a.x = (const __bf16 *)x; a.y = (__bf16 *)y; a.part = (float *)part; a.wpk = wp; a.shift = sh; a.B = B; a.Cin = C; a.Lin = L; a.Cout = C; a.Lout = L; a.stride = 1; a.ksz = K;
    a.pad_l = 2; a.nchunks = nchunks; a.residual = 0; a.S = S; a.Ntot = (long)B * L; a.slope = 0.0f; a.zeros = zr;
    const dim3 grid((unsigned)((a.Ntot + 255) / 256), C / 128, S);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 20;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((enc_conv_taps_kernel<5, 1>), grid, dim3(512), 0, 0, a);
    unsigned long long zero[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(taps_probe), zero, sizeof(zero));
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((enc_conv_taps_kernel<5, 1>), grid, dim3(512), 0, 0, a);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long pr[8];
    (void)hipMemcpyFromSymbol(pr, HIP_SYMBOL(taps_probe), sizeof(pr));
    const double nwg = (double)grid.x * grid.y * grid.z * reps;
    printf("enc_conv_taps_kernel<5, 1>, 2048 -> 2048, 1024 columns: %.1f us per launch (with stamps), %d workgroups, %d blocks x 5 taps each\n", 1000.0f * ms / reps, (int)(grid.x * grid.y * grid.z), nblk / S);
    printf("  inside the taps      %9.0f clocks per workgroup (%d taps x 64 MFMAs x 16 clocks = %d of them)\n", (double)pr[0] / nwg, 5 * nblk / S, 5 * nblk / S * 64 * 16);
    printf("  at the block barrier %9.0f clocks per workgroup\n", (double)pr[1] / nwg);
    printf("  epilogue             %9.0f clocks per workgroup\n", (double)pr[2] / nwg);
    return 0;
}
