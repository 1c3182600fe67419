#!/usr/bin/env python
"""Generates tools/micro/_gen/enc_kernels_probe.h: csrc/enc_kernels.h with s_memtime phase stamps in thread 0 of every workgroup of
enc_stereo_block_kernel (staging | first conv | mirror fix-up + second conv | epilogue + stores), summed into stereo_probe[].
The product sources carry no probe code.   python tools/micro/enc_stereo_probe.py && hipcc ... enc_stereo_probe.hip"""
import os

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(R, "music_mixing_style_transfer_amd", "csrc", "enc_kernels.h")).read()
lo = s.index("void enc_stereo_block_kernel(EncStereoArgs a) {")
hi = s.index("struct EncNlcArgs {")
k = s[lo:hi]


def patch(k, old, new):
    assert k.count(old) == 1, old[:60]
    return k.replace(old, new)


k = patch(k, "    const int b = blockIdx.x / a.tiles, t0", "    long long pt_ = mst_clock();\n"
          "#define PROBE(i) do { if (tid == 0) { const long long n_ = mst_clock(); atomicAdd(&stereo_probe[i], (unsigned long long)(n_ - pt_)); pt_ = n_; } } while (0)\n    const int b = blockIdx.x / a.tiles, t0")
k = patch(k, "    __syncthreads();\n    // ---- first conv", "    __syncthreads();\n    PROBE(0);\n    // ---- first conv")
k = patch(k, "    __syncthreads();\n    // ---- the second conv's reflection padding", "    __syncthreads();\n    PROBE(1);\n    // ---- the second conv's reflection padding")
k = patch(k, "    // ---- second conv: wave w owns", "    PROBE(2);\n    // ---- second conv: wave w owns")
k = k.rstrip()
assert k.endswith("}")
k = k[:-1] + "    PROBE(3);\n}\n\n"
out = s[:lo].replace("struct EncStereoArgs {", "__device__ unsigned long long stereo_probe[8];\nstruct EncStereoArgs {") + k + s[hi:]
os.makedirs(os.path.join(R, "tools", "micro", "_gen"), exist_ok=True)
open(os.path.join(R, "tools", "micro", "_gen", "enc_kernels_probe.h"), "w").write(out)
print("ok")
