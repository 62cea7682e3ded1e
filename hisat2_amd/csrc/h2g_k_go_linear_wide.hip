// Explicit instantiation of the single-end go() kernel for LINEAR indexes with the wide genome-hit list (AL_MAX_GHITS = 20:
// -k up to 10 / --max-seeds up to 20, e.g. --sensitive).  <3, false> only tags this build; see h2g_go_kernels.h.
#include "h2g_go_kernels.h"
template __global__ void k_align<3, false>(DGfm, DRef, DLocalSet, DReads, AlnParams, const char*, const uint32_t*, AlignWS*, ReadOut*, h2g_alnres*,
        unsigned long long*, const uint32_t*, unsigned long long*, uint8_t*, size_t, GraphArgs);
extern "C" size_t h2g_ws_bytes_linear_wide_se() { return sizeof(AlignWS); }
