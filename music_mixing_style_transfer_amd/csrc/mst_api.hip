// libmst_hip.so - C ABI implementation (host side): opaque handles, BatchNorm folding + weight packing into
// MFMA fragment order, tile geometry and kernel launches.  See include/mst_hip.h for the contract.
// One translation unit per kernel family (round 6): this file (version, errors), mst_tcn.hip (MixFXcloner), mst_enc.hip (FXencoder),
// mst_fx.hip (FX processors, FFT convolution, STFT); shared host-side internals in mst_host.h.
#include "mst_host.h"

namespace {
thread_local std::string g_err;
}

int mst_fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

extern "C" int mst_version(void) { return 100; }
extern "C" const char *mst_last_error(void) { return g_err.c_str(); }
