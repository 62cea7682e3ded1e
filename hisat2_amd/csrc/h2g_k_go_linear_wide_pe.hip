// Explicit instantiation of the paired-end go() kernel for LINEAR indexes with the wide genome-hit list (see h2g_k_go_linear_wide.hip).
#include "h2g_go_kernels.h"
template __global__ void k_align_pairs<false, 1>(DGfm, DRef, DLocalSet, DReads, DReads, AlnParams, const char*, const uint32_t*, const char*,
        const uint32_t*, AlignWS*, PairOut*, h2g_alnres*, h2g_alnres*, unsigned long long*, uint8_t*, size_t, GraphArgs);
extern "C" size_t h2g_ws_bytes_linear_wide_pe() { return sizeof(AlignWS); }
