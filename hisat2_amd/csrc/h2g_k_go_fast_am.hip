// go() fast pass with alignMate (hi_aligner.h:5579) restated in the pass: h2g_k_go_fast.hip compiled with FG_ALIGN_MATE = 1 (h2g_fast.h).
// Pairs without a concordant alignment stay in the pass instead of being handed on; which of the two kernels runs is go_run's choice
// (h2g_stream_tune "align_mate"; profiles/r04_NOTES.md has the measurements behind the default).
#define FG_ALIGN_MATE 1
#define FG_KERNEL   k_go_fast_am
#define FG_LAUNCH   h2g_go_fast_am_launch
#define FG_GEOMETRY h2g_go_fast_am_geometry
#include "h2g_k_go_fast.hip"
