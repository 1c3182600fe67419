// HIP runtime glue for the gfx950 build (hipcc).  The test-only CPU emulator substitutes its own file of
// the same name (tests/emu/mst_rt.h) earlier on the include path; product sources carry no #ifdefs.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define MST_LAUNCH(kern, grid, block, stream, ...) \
    hipLaunchKernelGGL(kern, (grid), (block), 0, (hipStream_t)(stream), __VA_ARGS__)
