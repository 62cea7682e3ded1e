// h2g_go_args.h — the one kernel-argument block of every go() kernel build, and the extern "C" face of the translation
// units that instantiate them.  Nothing in here depends on a unit's AL_MAX_* capacities: the per-lane workspaces are passed
// as byte pools with strides, the results as fixed-layout records.
#pragma once
#include <hip/hip_runtime.h>
#include "h2g_align.h"
#include "h2g_graph.h"
#include "h2g_fast.h"

__device__ __forceinline__ void wave_add(unsigned long long* dst, unsigned long long v) {
	for(int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	if((threadIdx.x & 63) == 0 && v) atomicAdd(dst, v);
}

struct GoArgs {
	h2g::DGfm g; h2g::DRef ref; h2g::DLocalSet ls; h2g::DAlts alts; h2g::DSpliceDB ssdb;
	h2g::DReads rd1, rd2;                 // rd2 only when paired
	h2g::AlnParams P;
	const char* names1; const uint32_t* noffs1;
	const char* names2; const uint32_t* noffs2;
	uint8_t* pool; size_t ws_stride;      // per read in flight (slot): AlignWS, then GoSlot at slot_off, then GraphSlot at gsl_off
	size_t slot_off, gsl_off;
	uint8_t* gws_base; size_t gws_stride; // GraphWS per lane (graph indexes): scratch of one primitive
	uint8_t* sw_base; size_t sw_stride;   // Smith-Waterman scratch per lane (bowtie2_dp != 0)
	uint8_t* sc_base;                     // combineWith temp_scores per lane: 2 x H2G_COMBINE_MAXLEN int64
	h2g::MachOut O;
	unsigned long long* counters;
	uint32_t* work;                       // next unclaimed position of the read list (device counter, zeroed before the launch)
	const uint32_t* list;                 // nullptr = every read of the batch; else the read ids to process ...
	const uint32_t* nlist;                // ... and how many (device memory: the second pass is launched without a host sync)
	uint32_t paired;
	uint32_t dbg_read; uint32_t* dbg_buf; // development hook: trace of one read id (H2G_GO_DBG_READ): [0] = words used, then 8 words per primitive request
	uint32_t rdid_base;                   // Read::rdid of read 0 of the batch (splice-site visibility window)
	uint32_t defer_overflow;              // 1: a second pass follows; overflowed reads are not counted as aligned here
};
#define H2G_PK_LANE_WORDS_HOST (H2G_PK_WORDS + H2G_PK_WORDS / 2)

#define H2G_GO_DECLARE(NAME) \
	extern "C" size_t h2g_go_ws_bytes_##NAME(); extern "C" size_t h2g_go_gws_bytes_##NAME(); extern "C" int h2g_go_waves_##NAME(); \
	extern "C" size_t h2g_go_slot_off_##NAME(); extern "C" size_t h2g_go_gsl_off_##NAME(); extern "C" void h2g_go_geometry_##NAME(uint32_t*); extern "C" size_t h2g_go_sw_bytes_##NAME(uint32_t, int); \
	extern "C" void h2g_go_caps_##NAME(uint32_t*); extern "C" int h2g_go_launch_##NAME(const GoArgs*, unsigned, hipStream_t);
H2G_GO_DECLARE(linear) H2G_GO_DECLARE(graph) H2G_GO_DECLARE(linear_big) H2G_GO_DECLARE(graph_big)
H2G_GO_DECLARE(linear_spl) H2G_GO_DECLARE(linear_spl_big) H2G_GO_DECLARE(graph_spl) H2G_GO_DECLARE(graph_spl_big)   // spliced alignment (linear indexes): with the splice-site database joins

// ---- the fast pass (h2g_fast.h / h2g_k_go_fast.hip): one read / pair per lane, state on chip; what it cannot hold goes to `bail_list`
struct FastArgs {
	h2g::DGfm g; h2g::DRef ref; h2g::DLocalSet ls;
	h2g::DReads rd1, rd2;
	h2g::AlnParams P;
	const char* names1; const uint32_t* noffs1;
	const char* names2; const uint32_t* noffs2;
	uint32_t* slots;                      // FG_SLOT_WORDS words per read in flight: packed state, hot words, packed reads, cold words
	h2g::FastOut O;
	unsigned long long* counters;         // [120] rank calls [121] sides [122] SA steps [123] aligned [6] completed [7] bailed, [96 + why] bails by reason
	uint32_t* work;                       // next unclaimed read (zeroed before the launch)
	uint32_t* bail_list; uint32_t* bail_count;
	uint32_t total, paired;
	// graph indexes only (h2g_k_go_fast_graph.hip): the ALT database and the per-LANE scratch of one primitive
	h2g::DAlts alts;
	uint8_t* gws_base; size_t gws_stride;     // GraphWS per lane (group walk, ALT-aware extension)
	uint8_t* sc_base;                         // combineWith temp_scores per lane: 2 x H2G_COMBINE_MAXLEN int64, lane-interleaved per wave
	uint32_t tail;                            // > 0: a workgroup hands its last `tail` reads in flight on to the general machine once the batch is exhausted
	uint32_t dbg_read; uint32_t* dbg_buf;     // development hook (-DFG_DBG_TRACE builds, h2g_stream_tune "dbg_read"): the trips of one read id, 10 words each behind a word count
	// the end of a batch (h2g_k_go_fast.hip, fk_loop): a launch with orphan_T > 0 lists the slots (workgroup x slots per workgroup + slot) of the reads its workgroups still held
	// when they could fetch no more and had thinned out to orphan_T reads; the drain launch (..._launch_drain) resumes the reads of `adopt_list` out of `adopt_slots`
	uint32_t orphan_T; uint32_t* orphan_list; uint32_t* orphan_count;
	const uint32_t* adopt_list; const uint32_t* adopt_count; const uint32_t* adopt_slots;
	uint32_t cnt_off;                         // this launch's rank / side / step / aligned counters are counters[120 + cnt_off ..]
	uint32_t adopt_slot_words;                // words per slot of `adopt_slots` (the launch that left them may be another build of the pass: same layout, a shorter cold tail)
	uint32_t mate_handover;                   // 1 (with orphan_T > 0, a build without alignMate): pairs that need alignMate are parked in their slots and listed for the drain launch — the alignMate build's
};
extern "C" int h2g_go_fast_launch(const FastArgs*, unsigned grid, hipStream_t);
extern "C" int h2g_go_fast_launch_drain(const FastArgs*, unsigned grid, hipStream_t);
extern "C" int h2g_go_fast_am_launch_drain(const FastArgs*, unsigned grid, hipStream_t);
extern "C" int h2g_go_fast_graph_launch_drain(const FastArgs*, unsigned grid, hipStream_t);
extern "C" void h2g_go_fast_geometry(uint32_t* g);   // [0] threads per workgroup [1] LDS bytes per workgroup [2] slots per workgroup [3] bytes per slot
extern "C" int h2g_go_fast_am_launch(const FastArgs*, unsigned grid, hipStream_t);         // h2g_k_go_fast_am.hip: alignMate in the pass (FG_ALIGN_MATE = 1)
extern "C" void h2g_go_fast_am_geometry(uint32_t* g);
extern "C" int h2g_go_fast_graph_launch(const FastArgs*, unsigned grid, hipStream_t);
extern "C" void h2g_go_fast_graph_geometry(uint32_t* g);   // ... [4] bytes of GraphWS per lane
