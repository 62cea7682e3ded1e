// go() kernel for SPLICED alignment on LINEAR indexes with the default capacities (see h2g_go_kernels.h): -k <= 5 / --max-seeds <= 10.
#define AL_MAX_GHITS 10   // max(khits, kseeds) of the default option set on a linear index (hisat2.cpp:3174-3176, 3903-3906)
#define H2G_SPLICE_DB 1   // spliced alignment: the machine with the splice-site database joins
#define H2G_HAPLOTYPE 0    // haplotypes belong to graph indexes
#include "h2g_go_kernels.h"
#ifndef H2G_LINEAR_WAVES
#define H2G_LINEAR_WAVES 2
#endif
H2G_GO_UNIT(linear_spl, false, H2G_LINEAR_WAVES, 4)
