// go() fast pass over a GRAPH index: h2g_k_go_fast.hip compiled with the graph form of the compact state (h2g_fast.h, FG_GRAPH = 1).
#define FG_GRAPH 1
#define FG_KERNEL   k_go_fast_graph
#define FG_LAUNCH   h2g_go_fast_graph_launch
#define FG_GEOMETRY h2g_go_fast_graph_geometry
#include "h2g_k_go_fast.hip"
