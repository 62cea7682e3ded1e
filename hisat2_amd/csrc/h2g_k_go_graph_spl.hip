// go() kernel for SPLICED alignment on GRAPH (SNP) indexes with the default capacities (see h2g_go_kernels.h): -k <= 10 / --max-seeds <= 20.
#define AL_MAX_GHITS 20
#define H2G_SPLICE_DB 1   // spliced alignment: the machine with the splice-site database joins
#include "h2g_go_kernels.h"
#ifndef H2G_GRAPH_WAVES
#define H2G_GRAPH_WAVES 2
#endif
H2G_GO_UNIT(graph_spl, true, H2G_GRAPH_WAVES, 6)
