// Explicit instantiation of the single-end go() kernel for LINEAR indexes (see h2g_go_kernels.h).
#include "h2g_go_kernels.h"
template __global__ void k_align<4, false>(DGfm, DRef, DLocalSet, DReads, AlnParams, const char*, const uint32_t*, AlignWS*, ReadOut*, h2g_alnres*,
        unsigned long long*, const uint32_t*, unsigned long long*, uint8_t*, size_t, GraphArgs);
