// Host-side internals shared by the translation units of libmst_hip.so (round 6: the C ABI is built from one TU per kernel family -
// mst_api.hip (errors, version), mst_tcn.hip, mst_enc.hip, mst_fx.hip - compiled in parallel).  Not part of the ABI: include/mst_hip.h is.
#pragma once
#include "../../include/mst_hip.h"

#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mst_dev.h"

#define MST_INTERNAL __attribute__((visibility("hidden")))

// records the message mst_last_error() returns (thread-local) and hands the code back
MST_INTERNAL int mst_fail(int code, const std::string &msg);
static inline int fail(int code, const std::string &msg) { return mst_fail(code, msg); }

#define MST_HIP_TRY(expr)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(MST_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define MST_CHECK_LAUNCH(name)                                                                     \
    do {                                                                                           \
        hipError_t e_ = hipGetLastError();                                                         \
        if (e_ != hipSuccess) return fail(MST_ERR_HIP, std::string(name) + ": " + hipGetErrorString(e_)); \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <typename T> int upload(T **dev, const std::vector<T> &host) {
    if (*dev == nullptr) MST_HIP_TRY(hipMalloc((void **)dev, host.size() * sizeof(T)));
    MST_HIP_TRY(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return MST_OK;
}

// host-side weight packing of the large encoder layers (81 M parameters, five images): the independent tiles of an image on a few threads
template <typename F> void host_parallel_for(int n, F fn) {
    const int nt = std::max(1, std::min({n, 16, (int)std::thread::hardware_concurrency()}));
    if (nt == 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([=]() {
            for (int i = t; i < n; i += nt) fn(i);
        });
    for (auto &t : th) t.join();
}

// eval-mode BatchNorm as y = x*scale + shift
static inline void bn_fold(const float *w, const float *b, const float *mean, const float *var, float eps, int c,
                           std::vector<float> &scale, std::vector<float> &shift) {
    scale.resize(c);
    shift.resize(c);
    for (int i = 0; i < c; ++i) {
        scale[i] = w[i] / std::sqrt(var[i] + eps);
        shift[i] = b[i] - mean[i] * scale[i];
    }
}

// one convolution layer as packed for enc_conv_kernel / enc_conv_bf16_kernel / the NLC pipeline
struct MstEncConv {
    float *wpk = nullptr, *shift = nullptr;
    __bf16 *wpk16 = nullptr;
    int *ktab = nullptr;
    float *w_direct = nullptr;   // [Cout][Cin][ksz] folded fp32 (layers with Cin < 8: direct kernel)
    __bf16 *w_taps = nullptr;    // 128-channel layers (Cin a multiple of 64, k = 5 / 10): enc_conv_taps_kernel's A fragments (enc_taps_pack)
    __bf16 *w_frag16 = nullptr;  // blocks 1 / 2 of the default encoder (Cin = 16, k = 25 / Cin = 32, k = 15): the fused kernel's bf16 A fragments, one 16-row tile after the other (enc_block1_pack)
    float *w_frag = nullptr;     // stereo block (Cin = 2, k = 25): the fused kernel's fp32 MFMA A fragments (enc_stereo_pack_a0 / _a1)
    __bf16 *wpk_nlc = nullptr;   // NLC pipeline A fragments, k = j*Cin + ci, K-chunk 64
    float slope = 0.0f;             // activation slope for negative values: 0 ReLU, 0.01 LeakyReLU, 1 none (MstEncDesc.act_slope)
    __bf16 *wpk_nlc_lo = nullptr;   // split mode: bf16(W' - bf16(W')) in the same fragment order
    int *stab = nullptr;         // NLC pipeline slot table
    int nchunks64 = 0;
    int cin = 0, cout = 0, ksz = 0, stride = 1, dil = 1, pad_l = 0, pad_r = 0, nchunks = 0, nchunks32 = 0, mw = 4;
    bool loaded = false;
};

// defined in mst_enc.hip (the implicit-GEMM conv kernel's packing and launch; the generic TCN configuration runs on it too)
MST_INTERNAL int pack_conv_f32(MstEncConv &c, const float *w, const std::vector<float> &scale);
MST_INTERNAL void conv_geometry(MstEncConv &c, int cin, int cout, int ksz, int stride, int dil, int pad_l, int pad_r);
MST_INTERNAL int conv_buf32(int mw, int B, int cin, long Lin, long Lout);
MST_INTERNAL int tcn_launch_generic(const MstEncConv &c, const float *x, float *y, int B, int L, int epi, const float *film, int film_rows,
                                    const float *res, int res_div, void *stream);
