// Explicit instantiation of the single-end go() kernel for GRAPH (SNP) indexes (see h2g_go_kernels.h).
#define H2G_INLINE_GLF 1
#include "h2g_go_kernels.h"
template __global__ void k_align<H2G_GRAPH_WAVES, true>(DGfm, DRef, DLocalSet, DReads, AlnParams, const char*, const uint32_t*, AlignWS*, ReadOut*, h2g_alnres*,
        unsigned long long*, const uint32_t*, unsigned long long*, uint8_t*, size_t, GraphArgs);
// per-lane workspace size of THIS translation unit's layout (AL_MAX_GHITS differs between the linear and graph units)
extern "C" size_t h2g_ws_bytes_graph_se() { return sizeof(AlignWS); }
