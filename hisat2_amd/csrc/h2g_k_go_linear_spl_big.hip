// go() kernel for LINEAR indexes with the large workspace (see h2g_go_big.h).
#include "h2g_go_big.h"
#define H2G_SPLICE_DB 1   // spliced alignment: the machine with the splice-site database joins
#define H2G_HAPLOTYPE 0    // haplotypes belong to graph indexes
#include "h2g_go_kernels.h"
H2G_GO_UNIT(linear_spl_big, false, 2, 5)
