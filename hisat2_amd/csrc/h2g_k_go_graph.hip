// go() kernel for GRAPH (SNP) indexes with the default capacities (see h2g_go_kernels.h): -k <= 10 / --max-seeds <= 20.
#define AL_MAX_GHITS 20
#define H2G_SPLICE_DB 0   // unspliced kernels: no splice-site database joins (h2g_machine.h)
#include "h2g_go_kernels.h"
#ifndef H2G_GRAPH_WAVES
#define H2G_GRAPH_WAVES 2
#endif
H2G_GO_UNIT(graph, true, H2G_GRAPH_WAVES, 1)
