// go() fast pass over a GRAPH index: h2g_k_go_fast.hip compiled with the graph form of the compact state (h2g_fast.h, FG_GRAPH = 1).
#define FG_GRAPH 1
// Capacities of the graph primitives' scratch in THIS unit (h2g_graph.h, include/h2g.h; the general machine's units keep theirs).  A hit of the fast path holds
// four edits and a resolution five coordinates, every capacity hit is flagged by the primitive and the read handed on: smaller lists change which reads the
// pass completes, never what it writes.
#define H2G_GHIT_EDITS 12
#define H2G_IEDGE_CAP  4       // a read of this pass keeps in-edge lists of at most two entries (fg_ie_pack); a longer one is counted, not stored
#define H2G_NEW_EDITS  8
#define H2G_GW_MAXELT  8
#define H2G_GW_MAXST   12
#define H2G_GW_MAXROWS 64      // fixed: the row masks of the group walk are 64-bit words
#define H2G_AWA_CAND   4
#define H2G_AWA_DEPTH  12
#define H2G_HAPLOTYPE  0       // --haplotype runs on the graph_spl units (go_run: `spl`), never through this pass
#define H2G_INLINE_GLF         // the graph LF leaf functions inline in this unit's search loops (lease C: 75.8 -> 74.2 ms per million pairs)
#define FG_KERNEL   k_go_fast_graph
#define FG_LAUNCH   h2g_go_fast_graph_launch
#define FG_GEOMETRY h2g_go_fast_graph_geometry
#include "h2g_k_go_fast.hip"
