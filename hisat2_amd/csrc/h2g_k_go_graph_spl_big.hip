// go() kernel for GRAPH (SNP) indexes with the large workspace (see h2g_go_big.h).
#include "h2g_go_big.h"
#define H2G_SPLICE_DB 1   // spliced alignment: the machine with the splice-site database joins
#include "h2g_go_kernels.h"
H2G_GO_UNIT(graph_spl_big, true, 2, 7)
