// go() kernel for LINEAR indexes with the large workspace (see h2g_go_big.h).
#include "h2g_go_big.h"
#define H2G_SPLICE_DB 0   // unspliced kernels: no splice-site database joins (h2g_machine.h)
#include "h2g_go_kernels.h"
H2G_GO_UNIT(linear_big, false, 2, 2)
