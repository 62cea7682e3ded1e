// Explicit instantiation of the single-end go() kernel for LINEAR indexes (see h2g_go_kernels.h).
#include "h2g_go_kernels.h"
template __global__ void k_align<H2G_LINEAR_WAVES, false>(DGfm, DRef, DLocalSet, DReads, AlnParams, const char*, const uint32_t*, AlignWS*, ReadOut*, h2g_alnres*,
        unsigned long long*, const uint32_t*, unsigned long long*, uint8_t*, size_t, GraphArgs);
// per-lane workspace size of THIS translation unit's layout (AL_MAX_GHITS differs between the linear and graph units)
extern "C" size_t h2g_ws_bytes_linear_se() { return sizeof(AlignWS); }
