#!/usr/bin/env python
"""Generates tools/micro/_gen/enc_kernels_probe_taps.h: csrc/enc_kernels.h with s_memtime stamps in matrix wave 0 (lane 0) of every workgroup of
enc_conv_taps_kernel: clocks inside the taps (MFMAs + their operand reads / waits) against clocks waiting at the per-block barrier, and the epilogue."""
import os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(R, "music_mixing_style_transfer_amd", "csrc", "enc_kernels.h")).read()
lo = s.index("void enc_conv_taps_kernel(EncTapsArgs a) {")
hi = s.index("// Res_ConvBlocks 1 and 2 of the default encoder in ONE launch each")
hi = s.rindex("// ----", lo, hi)
k = s[lo:hi]
def patch(k, old, new):
    assert k.count(old) == 1, old[:60]
    return k.replace(old, new)
k = patch(k, "    mst_dma_wait_barrier<63>();                                          // (P)\n", "    mst_dma_wait_barrier<63>();                                          // (P)\n    long long pt_ = mst_clock();\n"
          "#define PROBE(i) do { if (w == 0 && lane == 0) { const long long n_ = mst_clock(); atomicAdd(&taps_probe[i], (unsigned long long)(n_ - pt_)); pt_ = n_; } } while (0)\n")
k = patch(k, "            ++blk;\n            mst_dma_wait_barrier<63>();", "            ++blk;\n            PROBE(0);\n            mst_dma_wait_barrier<63>();\n            PROBE(1);")
k = patch(k, "}\n// host side: the A fragments of enc_conv_taps_kernel", "    PROBE(2);\n}\n// host side: the A fragments of enc_conv_taps_kernel")
import sys
var = sys.argv[1] if len(sys.argv) > 1 else ""
if var == "noA":          # the A fragments are fetched once (no weight stream)
    k = patch(k, "        fetch_a(Anext, t + 1);\n", "        if (t == 0) fetch_a(Anext, t + 1);\n")
if var == "noB":          # no B fragment reads (whatever the registers hold)
    k = patch(k, "            B[c] = *(const bf16x8 *)(bt + R * 128 + (((4 * kh + kk) ^ sw) << 4));", "            if (t < 0) B[c] = *(const bf16x8 *)(bt + R * 128 + (((4 * kh + kk) ^ sw) << 4));")
out = s[:lo].replace("struct EncTapsArgs {", "__device__ unsigned long long taps_probe[8];\nstruct EncTapsArgs {") + k + s[hi:]
open(os.path.join(R, "tools", "micro", "_gen", "enc_kernels_probe_taps.h"), "w").write(out)
print("ok")
