#!/usr/bin/env python
"""Generates tools/micro/_gen/enc_kernels_probe_b1.h: csrc/enc_kernels.h with s_memtime phase stamps in thread 0 of every workgroup of
enc_block1_fused_kernel (staging | first conv | weights of the second conv + barrier (+ mirror fix-up) | second conv + stores), summed into
block1_probe[].  The product sources carry no probe code."""
import os

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(R, "music_mixing_style_transfer_amd", "csrc", "enc_kernels.h")).read()
lo = s.index("void enc_block1_fused_kernel(EncBlock1Args a) {")
hi = s.index("// The 128-channel x 128-column tile with its four waves 2 x 2")
hi = s.rindex("// ----", lo, hi)
k = s[lo:hi]


def patch(k, old, new):
    assert k.count(old) == 1, old[:60]
    return k.replace(old, new)


k = patch(k, "    const int b = blockIdx.x / a.tiles, t0", "    long long pt_ = mst_clock();\n"
          "#define PROBE(i) do { if (tid == 0) { const long long n_ = mst_clock(); atomicAdd(&block1_probe[i], (unsigned long long)(n_ - pt_)); pt_ = n_; } } while (0)\n"
          "    const int b = blockIdx.x / a.tiles, t0")
k = patch(k, "    mst_dma_wait_barrier<0>();\n", "    mst_dma_wait_barrier<0>();\n    PROBE(0);\n")
k = patch(k, "    bf16x8 A1[2][KS];\n", "    PROBE(1);\n    bf16x8 A1[2][KS];\n")
k = patch(k, "    // ---- second conv: wave w owns", "    PROBE(2);\n    // ---- second conv: wave w owns")
k = k.rstrip()
assert k.endswith("}")
k = k[:-1] + "    PROBE(3);\n}\n\n"
out = s[:lo].replace("struct EncBlock1Args {", "__device__ unsigned long long block1_probe[8];\nstruct EncBlock1Args {") + k + s[hi:]
os.makedirs(os.path.join(R, "tools", "micro", "_gen"), exist_ok=True)
open(os.path.join(R, "tools", "micro", "_gen", "enc_kernels_probe_b1.h"), "w").write(out)
print("ok")
