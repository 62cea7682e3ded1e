// Explicit instantiation of the paired-end go() kernel for GRAPH (SNP) indexes (see h2g_go_kernels.h).
#include "h2g_go_kernels.h"
template __global__ void k_align_pairs<true>(DGfm, DRef, DLocalSet, DReads, DReads, AlnParams, const char*, const uint32_t*, const char*,
        const uint32_t*, AlignWS*, PairOut*, h2g_alnres*, h2g_alnres*, unsigned long long*, uint8_t*, size_t, GraphArgs);
// per-lane workspace size of THIS translation unit's layout (AL_MAX_GHITS differs between the linear and graph units)
extern "C" size_t h2g_ws_bytes_graph_pe() { return sizeof(AlignWS); }
