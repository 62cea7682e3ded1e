// h2g_fast.h — HI_Aligner::go for the DOMINANT traces with a compact per-read state.
//
// The general machine (h2g_machine.h) keeps a read's state in a ~115 KB workspace in HBM and pays for it: 47 KB of HBM traffic per read,
// control bound by scattered workspace lines.  On the benchmark's reads 99.5 % of the pairs only ever walk
//   partialSearch* -> getGenomeCoords -> extend -> [localGFMSearch -> getGenomeCoords_local -> extend -> combineWith | globalGFMSearch ..] -> report
// with at most two genome hits, two reported alignments per mate, four edits per hit and a recursion of depth <= 5.  This file restates exactly that
// part of go() (same reference lines as h2g_machine.h, cited per state) over a state of 40 words of bit-fields (FState: its own stored form), 50 hot
// words staged in LDS during a trip and the cold words of a ~1 KB slot in HBM, and gives up ("bail") the moment a read leaves it: anything it cannot
// hold, the random sub-sampling of a repeat's rows, mate rescue (restated too, behind FG_ALIGN_MATE: h2g_k_go_fast_am.hip), an insertion or deletion
// that survives combineWith's gap budget.  A bailed read is re-run FROM SCRATCH by the general machine, so the only obligation of this file is: a read
// it completes has exactly the machine's results (PairOut / ReadOut incl. the PRNG state and the work counters, every record).  tests/ hold it to
// that on the host (the same source, one lane at a time, every read through both) and on the device (pass on == pass off, queued-run stress).  The
// kernel (h2g_k_go_fast.hip) is the slot-queue scheme: a workgroup owns 1024 slots and one queue of slot ids per primitive; a wave pops <= 64 slots
// of the longest queue, loads them in one go, runs THAT primitive at one code site, lets each lane run its control flow to the next request.
//
// Built for: --no-spliced-alignment, no --secondary, --bowtie2-dp 0, default pair policy (--fr, -I 0), reads of 32..128 bases without N.
// Everything else never enters (go_run) or bails at the first state.
//
// FG_GRAPH = 1 (round 4) is the same state machine over a GRAPH index (SNP / indel ALTs): a partial hit carries its node range and in-edge
// list, a coordinate is re-seated by adjustWithALT before it becomes a hit, and a hit's edits carry ALT ids.  The graph primitives are the
// general machine's item functions (h2g_graph.h: mapGLF searches, the group walk, alignWithALTs) run on per-LANE scratch; what is compact is
// the state a read keeps between two of them.  One translation unit sees one setting (the kernels h2g_k_go_fast.hip / h2g_k_go_fast_graph.hip,
// the test libraries tests/emul/libh2gemu.so / libh2gemu_g.so).
#pragma once
#include "h2g_align.h"
#ifndef FG_GRAPH
#define FG_GRAPH 0
#endif
#if FG_GRAPH
#include "h2g_graph.h"
#endif

namespace h2g {

#define FG_FE     4                     // edits kept per hit (more: bail), 16 bits each
#define FG_HW     (6 + FG_FE / 2 + FG_GRAPH * FG_FE)   // words of a stored hit (graph: + the ALT id of each edit)
#define FG_LW     3                     // words of a long partial hit in the pool (graph: + 3 cold words at FW_LONGX: node range, in-edge list)
#define FG_NLONG  3                     // partial hits longer than minK + 2 waiting for getAnchorHits: the hot ones ...
#define FG_NLONGC 3                     // ... and the cold ones behind them (repeat-rich reads; FState::pool_x says which are taken, so the common read never looks)
#define FG_NLONGT (FG_NLONG + FG_NLONGC)
#define FG_NCO    5                     // coordinates per SA resolution (the reference resolves at most 5 outside getAnchorHits)
#define FG_NRES   2                     // reported alignments per mate
#define FG_NSRCH  7                     // hybridSearch_recur roots per mate (_hits_searched: hash + hit)
#define FG_NPAIR  4                     // concordant pairs
#define FG_FRS    5                     // scalar words of a saved frame
#define FG_NFRAME 5
#define FG_NLOCAL 2                     // _local_genomeHits kept per frame
#define FG_NGH    4                     // genome hits of one strand (getAnchorHits); the first is hot, the others cold
#ifndef FG_ALIGN_MATE
#define FG_ALIGN_MATE 0                 // 1: alignMate (hi_aligner.h:5579) in the fast path; 0: such pairs are handed on (the shipped build: DESIGN.md §3.1)
#endif
// word store of one read in flight: [0, FW_HOT) is staged in LDS while a wave works on it, [FW_HOT, FW_TOTAL) stays in HBM (coordinate lists,
// the searched list, everything only reads with a mismatch touch)
#define FW_LONG   0
#define FW_G0     (FW_LONG + FG_LW * FG_NLONG)
#define FW_FR0    (FW_G0 + FG_HW)                  // frame 0: scalars + hit
#if FG_GRAPH
// (the graph form of a hit is 12 words: to keep 8 waves per workgroup inside the LDS the result summaries and the pool's graph words are cold)
#define FW_T1     (FW_FR0 + FG_FRS + FG_HW)        // the scratch hit of the extension branches (every read with a mismatch works through it)
#define FW_HOT    (FW_T1 + FG_HW)
#define FW_RES    FW_HOT
#define FW_LONGX  (FW_RES + 2 * FG_NRES * 3)       // node range + in-edge list of each long partial hit
#define FW_LONGC  (FW_LONGX + 3 * FG_NLONGT)       // the cold pool entries
#define FW_G1     (FW_LONGC + FG_LW * FG_NLONGC)
#else
#define FW_RES    (FW_FR0 + FG_FRS + FG_HW)
#define FW_T1     (FW_RES + 2 * FG_NRES * 3)       // the scratch hit of the extension branches (every read with a mismatch works through it)
#define FW_HOT    (FW_T1 + FG_HW)
#define FW_LONGC  FW_HOT                           // the cold pool entries
#define FW_G1     (FW_LONGC + FG_LW * FG_NLONGC)
#endif
#define FW_SRCH   (FW_G1 + (FG_NGH - 1) * FG_HW)
#define FW_CO     (FW_SRCH + 2 * FG_NSRCH * (1 + FG_HW))   // the coordinate list of every frame (frame 0's doubles as getAnchorHits'); before it: (hash, hit) per searched root
#define FW_FRX    (FW_CO + FG_NFRAME * 3 * FG_NCO)  // frames 1 .. FG_NFRAME - 1: scalars + hit
#define FW_LH     (FW_FRX + (FG_NFRAME - 1) * (FG_FRS + FG_HW))   // _local_genomeHits of every frame
#define FW_RH     (FW_LH + FG_NFRAME * FG_NLOCAL * FG_HW)   // the reported alignments themselves (hits), [mate][k]: their records are written once, by FPC_FINISH
#define FW_AM      (FW_RH + 2 * FG_NRES * FG_HW)             // alignMate's loop state (4 words; only pairs without a concordant alignment get there)
#define FW_TOTAL  (FW_AM + 4 * FG_ALIGN_MATE)
#define FW_COLD   (FW_TOTAL - FW_HOT)

enum : uint32_t { FOP_NONE = 0, FOP_PSEARCH, FOP_GCOORDS, FOP_EXTEND, FOP_LSEARCH, FOP_LCOORDS, FOP_COMBINE, FOP_GSEARCH, FOP_ADJUST, FOP_ADJMEMBER, FOP_COUNT };
// not a primitive: the read stands at a state only the alignMate build of this file can run (FCtx::mate_handover) — the kernel parks its slot for that build's drain launch
#define FOP_HANDOVER 15u
// ... and not one either: the trip of a queue whose lanes wait for different primitives (FQ_WALK_SLOW) runs each lane's own S.op
#define FOP_PER_LANE 14u
enum : uint32_t {
	FPC_DONE = 0, FPC_BAIL,
	FPC_GO_INIT, FPC_NB_PICK, FPC_NB_AFTER_PS, FPC_ALIGN, FPC_AFTER_ALIGN, FPC_PAIR_READS, FPC_AFTER_LOOP, FPC_FINISH,
	FPC_GAH_LOOP, FPC_GAH_FULL_AFTER, FPC_GAH_END, FPC_HS_EXT_LOOP, FPC_HS_EXT_AFTER, FPC_HS_LOOP, FPC_HS_AFTER_REC1,
	FPC_RC_ENTRY, FPC_RC_ENTRY_LX, FPC_RC_ENTRY_L2, FPC_RC_ENTRY_L3, FPC_RC_ENTRY_RX, FPC_RC_ENTRY_R2, FPC_RC_ENTRY_R3,
	FPC_L_WHILE, FPC_L_LS_LOOP, FPC_L_LS_AFTER, FPC_L_LS_DONE, FPC_L_LC_AFTER, FPC_L_FOR_RI, FPC_L_RI_B, FPC_L_RI_B2, FPC_L_RI_C, FPC_L_R1,
	FPC_L_AFTER_FOR, FPC_L_FOR_TI, FPC_L_R2, FPC_L_AFTER_WHILE, FPC_L_GS_AFTER, FPC_L_GC_AFTER, FPC_L_FOR_G, FPC_L_G_B, FPC_L_G_C, FPC_L_R3, FPC_L_TRIM, FPC_L_R4, FPC_L_EXT, FPC_L_EXT_A, FPC_L_R5,
	FPC_R_WHILE, FPC_R_LS_LOOP, FPC_R_LS_AFTER, FPC_R_LS_DONE, FPC_R_LC_AFTER, FPC_R_FOR_RI, FPC_R_RI_B, FPC_R_RI_C, FPC_R_R1,
	FPC_R_AFTER_FOR, FPC_R_FOR_TI, FPC_R_R2, FPC_R_AFTER_WHILE, FPC_R_GS_AFTER, FPC_R_GC_AFTER, FPC_R_FOR_G, FPC_R_G_B, FPC_R_G_C, FPC_R_R3, FPC_R_TRIM, FPC_R_R4, FPC_R_EXT, FPC_R_EXT_A, FPC_R_R5,
	FPC_GAH_K_LOOP, FPC_GAH_K_AFTER, FPC_L_RI_A, FPC_L_G_A, FPC_R_RI_A, FPC_R_G_A,     // graph: adjustWithALT of an anchor's coordinates / of a local hit
	FPC_MP_LOOP, FPC_AM_WHILE, FPC_AM_INNER, FPC_AM_AFTER_LS, FPC_AM_AFTER_LC, FPC_AM_RI_LOOP, FPC_AM_ADV, FPC_AM_EXT_LOOP, FPC_AM_EXT_AFTER, FPC_AM_REC_AFTER
};

// why a read left the fast path (statistics only)
enum : uint32_t {
	FB_NONE = 0, FB_INPUT, FB_LONGPOOL, FB_SUBSAMPLE, FB_COORDS, FB_NGHITS, FB_EDITS, FB_DEPTH, FB_LOCALHITS, FB_GSEARCH, FB_NRES,
	FB_SEARCHED, FB_REDUNDANT, FB_MATE, FB_NPAIRS, FB_PARTIAL, FB_STRADDLE, FB_OTHER, FB_INDEL, FB_TAIL, FB_IEDGES, FB_GWALK, FB_COUNT
};

#if defined(__HIP_DEVICE_COMPILE__)
#define FG_LDS  __attribute__((address_space(3)))   // ds_read / ds_write instead of flat accesses
#define FG_PRIV __attribute__((address_space(5)))
#else
#define FG_LDS
#define FG_PRIV
#endif
struct FWords {                          // the lane's word store
	FG_LDS uint32_t* hot; uint32_t hot_stride;  // LDS, lane-interleaved (host: stride 1)
	uint32_t* cold;                             // the slot's cold words in HBM (reads with a mismatch only)
	H2G_HD uint32_t ld(uint32_t i) const { return i < FW_HOT ? hot[i * hot_stride] : cold[i - FW_HOT]; }
	H2G_HD void st(uint32_t i, uint32_t v) const { if(i < FW_HOT) hot[i * hot_stride] = v; else cold[i - FW_HOT] = v; }
	// N consecutive words of one object (objects never straddle the hot / cold boundary): ONE branch, then N independent accesses —
	// a cold object costs one HBM latency instead of N
	template <int N> H2G_HD void ldv(uint32_t i, uint32_t* o) const {
		if(i < FW_HOT) {
#pragma unroll
			for(int k = 0; k < N; k++) o[k] = hot[(i + k) * hot_stride];
		} else {
			const uint32_t* c = cold + (i - FW_HOT);
#pragma unroll
			for(int k = 0; k < N; k++) o[k] = c[k];
		}
	}
	template <int N> H2G_HD void stv(uint32_t i, const uint32_t* v) const {
		if(i < FW_HOT) {
#pragma unroll
			for(int k = 0; k < N; k++) hot[(i + k) * hot_stride] = v[k];
		} else {
			uint32_t* c = cold + (i - FW_HOT);
#pragma unroll
			for(int k = 0; k < N; k++) c[k] = v[k];
		}
	}
};

struct FastOut {                         // where a completed read leaves its results (MachOut of h2g_machine.h)
	ReadOut*    rout; h2g_alnres* aln; uint32_t aln_slots;
	PairOut*    pout; h2g_alnres* paln[2]; uint32_t pair_slots;
};

// The state of one read between (and during) two trips: FS_WORDS 32-bit words of bit-fields.  It is the slot's stored form as it
// is (no packing step), and in a kernel it costs FS_WORDS registers instead of one per field — the control code of go() touches a
// handful of fields per state, and a field access is a bit-field extract / insert.
struct FState {
	uint32_t pc : 8, op : 4, bail : 5, paired : 1, nm : 2, found : 4, rb_done : 4, rb_nonempty : 4;                                          // 0
	uint32_t rl0 : 8, rl1 : 8, sel_r : 1, sel_f : 1, nb_rdi : 1, nb_fwi : 1, sv_rdi : 1, sv_fw : 1, nres0 : 2, nres1 : 2, nsearched0 : 3, nsearched1 : 3;   // 1
	uint32_t a0, a1, a2, a3, a4, a5;                                                                                                   // 2-7
	uint32_t read, ro0, ro1, rnd;                                                                                                      // 8-11
	uint32_t rb_cur, rb_nps, rb_nus, rb_np;          // ReadBWTHit x 4 (hi_aligner.h:216), 8 bits each, index = rdi * 2 + fwi          // 12-15
	uint32_t rb_sumsq0, rb_sumsq1, rb_sumsq2, rb_sumsq3;                                                                               // 16-19
	int32_t  bestUnp0 : 16, bestUnp1 : 16;           // the sink's per-mate bests; F_SMIN16 = none                                       // 20
	int32_t  best2Unp0 : 16, best2Unp1 : 16;                                                                                            // 21
	int32_t  minsc0 : 16, minsc1 : 16;               // F_SMAX16 = the mate is not there                                                // 22
	uint32_t npairs : 3, pairs : 16, insp_i : 2, insp_j : 2, hs_found : 1, rc_mate : 1, pool_x : 3, pad23_ : 4;                   // pairs: 4 bits per pair (i | j << 2)          // 23
	int32_t  bestPair, best2Pair;                                                                                                      // 24, 25
	uint32_t nrank : 16, nside : 16;                                                                                                    // 26
	uint32_t nsteps : 16, nframes_max : 8, pad27_ : 8;                                                                                  // 27
	uint32_t nghits : 3, ghit_done : 4, gh_hi : 5, gh_hj : 3, gh_nco : 3, gh_rdoff : 8, hs_hi : 3, hs_hj : 3;                 // 28
	uint32_t localindexatts : 16, max_localindexatts : 16;                                                                              // 29
	int32_t  sp : 4; uint32_t rc_ret_pc : 8, pr_ret_pc : 8, pad30_ : 12;                                                                // 30
	int32_t  rc_minsc, ret;                                                                                                            // 31, 32
	// the CURRENT frame of hybridSearch_recur (saved into the word store across a nested call)
	uint32_t f_hitoff : 8, f_hitlen : 8, f_extoff : 8, f_extlen : 8;                                                                     // 33
	uint32_t f_lidx;                                                                                                                   // 34
	uint32_t f_state : 8, f_count : 2, f_ncoords : 3; int32_t f_ri : 4; uint32_t f_success : 1, f_first : 1, f_uselocal : 1, f_unique : 1, f_noext : 1, f_nlocal : 2, f_ti : 2, pad35_ : 6;   // 35
	int32_t  f_maxsc, f_prev;                                                                                                          // 36, 37
	uint32_t f_top : 16, f_bot : 16;                 // rows of a local index (16-bit words); 0xffff stands for "none"                   // 38
	uint32_t f_nelt : 16, f_maxHitLen : 16;                                                                                             // 39
#if FG_GRAPH
	uint32_t a6, a7, a8;                             // node range + packed in-edge list of the last search (BWTHit::_node_top/_node_bot/_node_iedge_count)   // 40-42
	uint32_t gh_k : 3, gh_gsize : 3, pad43_ : 26;    // getAnchorHits' loop over an anchor's coordinates (each goes through adjustWithALT)                  // 43
	uint32_t f_ntop : 16, f_nbot : 16;               // node range of the frame's local search                                                           // 44
	uint32_t f_ie;                                   // ... and its in-edge list                                                                        // 45
	uint32_t pad46_, pad47_;                                                                                                                         // 46, 47
#endif
};
#define FS_WORDS (40 + 8 * FG_GRAPH)
static_assert(sizeof(FState) == FS_WORDS * 4, "FState is FS_WORDS words");
#define F_SMIN16 (-32768)
#define F_SMAX16 32767
// per-mate / per-strand fields by index
H2G_HD uint32_t fs_rl(const FState& S, uint32_t m) { return m ? S.rl1 : S.rl0; }
H2G_HD uint32_t fs_ro(const FState& S, uint32_t m) { return m ? S.ro1 : S.ro0; }
H2G_HD uint32_t fs_nres(const FState& S, uint32_t m) { return m ? S.nres1 : S.nres0; }
H2G_HD uint32_t fs_nsearched(const FState& S, uint32_t m) { return m ? S.nsearched1 : S.nsearched0; }
H2G_HD int32_t fs_bestUnp(const FState& S, uint32_t m) { return m ? S.bestUnp1 : S.bestUnp0; }
H2G_HD int32_t fs_minsc(const FState& S, uint32_t m) { return m ? S.minsc1 : S.minsc0; }
H2G_HD uint32_t fs_b4(uint32_t w, uint32_t x) { return (w >> (8 * x)) & 0xffu; }
H2G_HD uint32_t fs_b4_set(uint32_t w, uint32_t x, uint32_t v) { return (w & ~(0xffu << (8 * x))) | ((v & 0xffu) << (8 * x)); }
H2G_HD uint32_t fs_sumsq(const FState& S, uint32_t x) { return x == 0 ? S.rb_sumsq0 : x == 1 ? S.rb_sumsq1 : x == 2 ? S.rb_sumsq2 : S.rb_sumsq3; }
H2G_HD void fs_sumsq_add(FState& S, uint32_t x, uint32_t v) { if(x == 0) S.rb_sumsq0 += v; else if(x == 1) S.rb_sumsq1 += v; else if(x == 2) S.rb_sumsq2 += v; else S.rb_sumsq3 += v; }
#define F_SMIN INT32_MIN

struct FCtx {
	const DGfm* g; const DRef* ref; const DLocalSet* ls; const AlnParams* P;
	DReads rd[2];
	const uint32_t* pk[2]; uint32_t pk_stride;     // this lane's packed reads (2-bit words only: reads with an N never enter)
	const char* name[2]; uint32_t namelen[2];
	int64_t* sc; uint32_t sc_stride;               // combineWith temp_scores of this lane
	FastOut O;
	// 1 (graph indexes, the queued kernel): a primitive whose read turns out to need its slow form — the ALT-aware extension over the unpacked hit, the node-based group walk on the
	// lane's scratch: 300-900 us where the common form takes 30-170 — marks its arguments and asks to be queued again, on a queue of such requests (FQ_EXTEND_SLOW / FQ_WALK_SLOW):
	// a wave waits for its slowest lane, and with a third of the reads near a variant every trip had one (lease P: E:HS 308 us per trip against 29 on a linear index)
	uint32_t defer_slow = 0;
	uint32_t mate_handover = 0;                    // 1 (a build without alignMate): a pair that needs alignMate stops at FPC_AFTER_LOOP with FOP_HANDOVER instead of leaving the fast path
#if FG_GRAPH
	const DAlts* alts;                             // the ALT database
	GraphWS* gws;                                  // this LANE's scratch of one primitive (group walk, ALT-aware extension): nothing in it outlives a trip
#endif
};
#if FG_GRAPH
// BWTHit::_node_iedge_count in one word: n (2 bits), then (node index, extra in-edges) 7 + 7 bits each, at most two entries; a list that does
// not fit is FG_IE_NOFIT (the read leaves the fast path the moment that list is needed)
#define FG_IE_NOFIT 0xffffffffu
H2G_HD uint32_t fg_ie_pack(const IEdges& ie) {
	if(ie.n > 2) return FG_IE_NOFIT;
	uint32_t w = ie.n;
	if(ie.n > 0) { if(ie.e[0][0] > 127 || ie.e[0][1] > 127) return FG_IE_NOFIT; w |= (ie.e[0][0] << 2) | (ie.e[0][1] << 9); }
	if(ie.n > 1) { if(ie.e[1][0] > 127 || ie.e[1][1] > 127) return FG_IE_NOFIT; w |= (ie.e[1][0] << 16) | (ie.e[1][1] << 23); }
	return w;
}
H2G_HD void fg_ie_unpack(uint32_t w, IEdges* ie) {
	ie->n = w & 3u;
	ie->e[0][0] = (w >> 2) & 127u; ie->e[0][1] = (w >> 9) & 127u;
	ie->e[1][0] = (w >> 16) & 127u; ie->e[1][1] = (w >> 23) & 127u;
}
#endif

// ---------------------------------------------------------------------------------------- stored hits
H2G_HD uint32_t fg_pool(uint32_t k) { return k < FG_NLONG ? (uint32_t)FW_LONG + FG_LW * k : (uint32_t)FW_LONGC + FG_LW * (k - FG_NLONG); }   // long partial hit k of the pool
H2G_HD uint32_t fg_gbase(uint32_t k) { return k == 0 ? (uint32_t)FW_G0 : (uint32_t)FW_G1 + (k - 1) * FG_HW; }      // genome hit k of the strand
H2G_HD uint32_t fg_frame_base(int sp) { return sp == 0 ? (uint32_t)FW_FR0 : (uint32_t)FW_FRX + (uint32_t)(sp - 1) * (FG_FRS + FG_HW); }
H2G_HD uint32_t fg_frame_hit(int sp) { return fg_frame_base(sp) + FG_FRS; }
H2G_HD uint32_t fg_frame_lh(int sp, uint32_t k) { return (uint32_t)FW_LH + ((uint32_t)sp * FG_NLOCAL + k) * FG_HW; }
H2G_HD uint32_t fg_frame_co(int sp) { return (uint32_t)FW_CO + (uint32_t)sp * 3 * FG_NCO; }

// hit words: tidx, toff, joinedOff, score, rdoff | len << 8 | trim5 << 16 | trim3 << 24, fw | nedits << 1 | hitcount << 8, edits
H2G_HD void fg_hit_init(const FWords& W, uint32_t hb, bool fw, uint32_t rdoff, uint32_t len, uint32_t tidx, uint32_t toff, uint32_t joff) {
	const uint32_t v[6] = {tidx, toff, joff, 0u, rdoff | (len << 8), (fw ? 1u : 0u) | (1u << 8)};
	W.stv<6>(hb, v);
}
H2G_HD void fg_hit_copy(const FWords& W, uint32_t dst, uint32_t src) {
	if(dst == src) return;
	uint32_t v[FG_HW];
	W.ldv<FG_HW>(src, v);
	W.stv<FG_HW>(dst, v);
}
// A hit in REGISTERS: what GenomeHit holds for an alignment with at most FG_FE mismatch / gap edits on a linear index (no ALT ids,
// no splices).  Every loop over its edits is unrolled over FG_FE selects: nothing of it is ever indexed in private memory.
struct FHit {
	uint32_t tidx, toff, joff; int32_t score;
	uint32_t rdoff, len, trim5, trim3, fw, nedits, hitcount;
	uint64_t e;                           // the edits, 16 bits each, in read order: pos | ref base << 8 | read base << 11 | type << 14 (bases 0..4 = ACGTN, 5 = '-')
#if FG_GRAPH
	uint32_t snp[FG_FE];                  // Edit::snpID of each edit: index into the ALT list, H2G_MAX = not through a known variant
#endif
	uint32_t bad;                         // it stopped fitting (more than FG_FE edits): the read leaves the fast path
};
static_assert(FG_FE == 4, "FHit holds four 16-bit edits in one 64-bit word");
#define FE_GET(H, K) ((uint32_t)((H).e >> (16u * (uint32_t)(K))) & 0xffffu)
#define FE_SET(H, K, V) do { const uint32_t sh_ = 16u * (uint32_t)(K); (H).e = ((H).e & ~(0xffffull << sh_)) | ((uint64_t)((V) & 0xffffu) << sh_); } while(0)
#define FE_POS(E)  ((E) & 0xffu)
#define FE_CCODE(E) (((E) >> 8) & 7u)
#define FE_QCODE(E) (((E) >> 11) & 7u)
#define FE_DEC(K)  ((uint32_t)(0x2d4e54474341ull >> (8u * (K))) & 0xffu)      // "ACGTN-"
#define FE_CHR(E)  FE_DEC(FE_CCODE(E))
#define FE_QCHR(E) FE_DEC(FE_QCODE(E))
#define FE_TYPE(E) (((E) >> 14) & 3u)
#define FE_MAKE_BP(POS, RFBP, RDBP, TYPE) ((uint32_t)(POS) | ((uint32_t)(RFBP) << 8) | ((uint32_t)(RDBP) << 11) | ((uint32_t)(TYPE) << 14))
static_assert(H2G_EDIT_MM < 4 && H2G_EDIT_READ_GAP < 4 && H2G_EDIT_REF_GAP < 4, "edit types in two bits");
H2G_HD bool fe_is_gap(uint32_t e) { const uint32_t t = FE_TYPE(e); return t == H2G_EDIT_READ_GAP || t == H2G_EDIT_REF_GAP; }
// the edits getLeft / getRight stop at (is_stop_edit of h2g_core.h; hi_aligner.h:937-940, :981-984): gaps, and on a graph index every edit
// through a known variant
#if FG_GRAPH
#define FH_STOP(H, K) (fe_is_gap(FE_GET(H, K)) || (H).snp[K] != H2G_MAX)
#define FH_ALT(H, K)  ((H).snp[K] != H2G_MAX)
#else
#define FH_STOP(H, K) fe_is_gap(FE_GET(H, K))
#define FH_ALT(H, K)  false
#endif

H2G_HD FHit fh_load(const FWords& W, uint32_t hb) {
	uint32_t v[FG_HW];
	W.ldv<FG_HW>(hb, v);
	FHit h;
	h.tidx = v[0]; h.toff = v[1]; h.joff = v[2]; h.score = (int32_t)v[3];
	h.rdoff = v[4] & 0xffu; h.len = (v[4] >> 8) & 0xffu; h.trim5 = (v[4] >> 16) & 0xffu; h.trim3 = v[4] >> 24;
	h.fw = v[5] & 1u; h.nedits = (v[5] >> 1) & 7u; h.hitcount = v[5] >> 8;
	h.e = (uint64_t)v[6] | (uint64_t)v[7] << 32;
	h.e &= h.nedits >= 4 ? ~0ull : (1ull << (16u * h.nedits)) - 1ull;
#if FG_GRAPH
#pragma unroll
	for(uint32_t k = 0; k < FG_FE; k++) h.snp[k] = k < h.nedits ? v[8 + k] : H2G_MAX;
#endif
	h.bad = 0;
	return h;
}
// false: the hit does not fit the stored form (the caller bails)
H2G_HD bool fh_store(const FWords& W, uint32_t hb, const FHit& h) {
	if(h.bad || h.nedits > FG_FE || h.score < -(1 << 30) || h.score > (1 << 30) || h.hitcount > 0xffffu) return false;
	if(h.rdoff > 255 || h.len > 255 || h.trim5 > 255 || h.trim3 > 255) return false;
	const uint32_t v[FG_HW] = {h.tidx, h.toff, h.joff, (uint32_t)h.score, h.rdoff | (h.len << 8) | (h.trim5 << 16) | (h.trim3 << 24),
	                           (h.fw ? 1u : 0u) | (h.nedits << 1) | (h.hitcount << 8), (uint32_t)h.e, (uint32_t)(h.e >> 32)
#if FG_GRAPH
	                           , h.snp[0], h.snp[1], h.snp[2], h.snp[3]
#endif
	};
	W.stv<FG_HW>(hb, v);
	return true;
}
// a hash of exactly what GenomeHit::operator== compares (hit_equal, hi_aligner.h:1156): equal hits => equal hashes
H2G_HD uint32_t fh_hash(const FHit& h) {
	uint32_t x = 0x9e3779b9u;
#define FG_MIX(V) do { x ^= (uint32_t)(V); x *= 0x85ebca6bu; x ^= x >> 13; } while(0)
	FG_MIX(h.fw); FG_MIX(h.rdoff); FG_MIX(h.len); FG_MIX(h.tidx); FG_MIX(h.toff); FG_MIX(h.trim5); FG_MIX(h.trim3); FG_MIX(h.nedits);
#pragma unroll
	for(uint32_t i = 0; i < FG_FE; i++) if(i < h.nedits) {
		const uint32_t e = FE_GET(h, i);
		if(fe_is_gap(e)) FG_MIX(FE_TYPE(e)); else FG_MIX(e);
	}
#undef FG_MIX
	return x;
}
// operator== hi_aligner.h:1156-1183 (hit_equal)
H2G_HD bool fh_equal(const FHit& a, const FHit& b) {
	if(a.fw != b.fw || a.rdoff != b.rdoff || a.len != b.len || a.tidx != b.tidx || a.toff != b.toff || a.trim5 != b.trim5 || a.trim3 != b.trim3) return false;
	if(a.nedits != b.nedits) return false;
	bool eq = true;
#pragma unroll
	for(uint32_t i = 0; i < FG_FE; i++) if(i < a.nedits) {
		const uint32_t e = FE_GET(a, i), o = FE_GET(b, i);
		if(fe_is_gap(e)) { if(FE_TYPE(o) != FE_TYPE(e)) eq = false; }
		else if(e != o) eq = false;
	}
	return eq;
}
// getRight hi_aligner.h:962-1000 (hit_get_right): the part behind the last gap
H2G_HD void fh_get_right(const FHit& h, uint32_t* rdoff, uint32_t* len, uint32_t* toff) {
	*rdoff = h.rdoff; *len = h.len; *toff = h.toff;
	int last = -1;
#pragma unroll
	for(int i = 0; i < FG_FE; i++) if((uint32_t)i < h.nedits && FH_STOP(h, i)) last = i;
	if(last < 0) return;
	const uint32_t e = FE_GET(h, last);
	*rdoff = h.rdoff + FE_POS(e); *len = h.len - FE_POS(e);
	if(FE_TYPE(e) == H2G_EDIT_REF_GAP || FE_TYPE(e) == H2G_EDIT_MM) { (*rdoff)++; (*len)--; }     // (a mismatch stops the part on a graph index only: hit_get_right)
	uint32_t roff = h.toff + h.len;
#pragma unroll
	for(uint32_t k = 0; k < FG_FE; k++) if(k < h.nedits) {
		const uint32_t t = FE_TYPE(FE_GET(h, k));
		if(t == H2G_EDIT_READ_GAP) roff++; else if(t == H2G_EDIT_REF_GAP) roff--;
	}
	*toff = roff - *len;
}
// getLeft :919-958 (hit_get_left): the part in front of the first gap
H2G_HD void fh_get_left(const FHit& h, uint32_t* rdoff, uint32_t* len, uint32_t* toff) {
	*toff = h.toff; *rdoff = h.rdoff; *len = h.len;
	bool stop = false;
#pragma unroll
	for(uint32_t i = 0; i < FG_FE; i++) if(i < h.nedits && !stop && FH_STOP(h, i)) { *len = FE_POS(FE_GET(h, i)); stop = true; }
}
// compatibleWith :1375-1413 (hit_compatible) without spliced alignment
H2G_HD bool fh_compatible(const FHit& a, const FHit& b) {
	if(a.fw != b.fw || a.tidx != b.tidx) return false;
	if(a.rdoff > b.rdoff) return false;
	if(a.rdoff + a.len > b.rdoff + b.len) return false;
	if(a.toff > b.toff) return false;
	uint32_t ar, al, at, br, bl, bt;
	fh_get_right(a, &ar, &al, &at);
	fh_get_left(b, &br, &bl, &bt);
	if(ar > br) return false;
	if(ar + al > br + bl) return false;
	if(at > bt) return false;
	return true;
}
// calculateScore :3711-3891 (calculate_score) for mismatch / gap edits and soft trims
H2G_HD void fh_calc_score(const DScoring& sc, const SeqView& seq, FHit& h) {
	int64_t score = 0;
	uint32_t mm = 0, prev = 0;
#pragma unroll
	for(uint32_t i = 0; i < FG_FE; i++) if(i < h.nedits) {
		const uint32_t e = FE_GET(h, i), t = FE_TYPE(e);
		if(FH_ALT(h, i)) { prev = e; continue; }           // edits through known variants cost nothing (calculate_score of h2g_core.h; hi_aligner.h:3737, :3846, :3858)
		if(t == H2G_EDIT_MM) {
			const int q = seq.qual(h.rdoff + FE_POS(e)) - 33;
			if(FE_QCODE(e) == 4) score -= sc.nPen;
			else if(FE_CCODE(e) == 4) score += sc.matchBonus;
			else score -= mm_penalty(sc, q);
			mm++;
		} else if(t == H2G_EDIT_READ_GAP) {
			const bool open = !(i > 0 && FE_TYPE(prev) == H2G_EDIT_READ_GAP && FE_POS(prev) == FE_POS(e));
			score -= open ? (sc.rdGapConst + sc.rdGapLinear) : sc.rdGapLinear;
		} else if(t == H2G_EDIT_REF_GAP) {
			const bool open = !(i > 0 && FE_TYPE(prev) == H2G_EDIT_REF_GAP && FE_POS(prev) + 1 == FE_POS(e));
			score -= open ? (sc.rfGapConst + sc.rfGapLinear) : sc.rfGapLinear;
		}
		prev = e;
	}
	for(uint32_t i = 0; i < h.trim5; i++) score -= sc_penalty(sc, seq.qual(i));
	for(uint32_t i = 0; i < h.trim3; i++) score -= sc_penalty(sc, seq.qual(i));
	score += (int64_t)(h.len - mm) * sc.matchBonus;
	if(score < -(1 << 30)) { h.bad = 1; score = -(1 << 30); }
	h.score = (int32_t)score;
}
// positions (bit 2j = position j, j < n <= 32) at which two 2-bit strings differ
H2G_HD uint64_t fh_diff32(uint64_t r, uint64_t q, uint32_t n) {
	const uint64_t x = r ^ q;
	uint64_t m = (x | (x >> 1)) & 0x5555555555555555ull;
	if(n < 32u) m &= (1ull << (2u * n)) - 1ull;
	return m;
}
// alignWithALTs without ALTs (align_no_alts of h2g_core.h; hi_aligner.h:683-783, :2763-2853, :3168-3216): extends by up to `mm`
// mismatches; the new edits are committed to the hit.  Returns the extension length.
H2G_HD uint32_t fh_align(const DRef& ref, const SeqView& seq, uint32_t base_rdoff, uint32_t rdoff, uint32_t rdlen, int rfoff, uint32_t rflen,
                         bool left, FHit& h, uint32_t mm, uint32_t* numNs)
{
	if(numNs) *numNs = 0;
	const uint32_t n_old = h.nedits;
	uint64_t nw = 0;                                       // the first four new edits, in the order met
	uint32_t tmp_mm = 0, nNs = 0, extlen = 0;
	bool updated = false;
	const uint32_t contig_len = ref.refLens[h.tidx];
	bool run = !(rfoff < -16) && !((int64_t)rfoff >= (int64_t)contig_len);
	if(run) {
		if(rfoff >= 0 && (uint64_t)rfoff + rflen > contig_len) rflen = contig_len - (uint32_t)rfoff;
		else if(rfoff < 0 && rflen > contig_len) rflen = contig_len;
		if(rflen == 0) run = false;
	}
	if(run) {
		RefCursor rc;
		rc.init(&ref, h.tidx);
		const uint32_t rdoff_add = rdoff - base_rdoff;
		// 32 bases per step where the window lies inside one stretch of unambiguous reference bases (nearly always): the two 2-bit strings
		// are XORed word-wise and only the mismatches are visited, in the order the base-by-base loops below meet them
		const uint32_t limit = left ? (rflen < rdoff + 1 ? rflen : rdoff + 1) : (rflen < rdlen ? rflen : rdlen);
		const bool wordwise = seq.pk != nullptr && seq.pk_nomask && limit > 0 && rc.covers(left ? (int64_t)rfoff + rflen - limit : (int64_t)rfoff, limit);
		if(wordwise && left) {
			uint32_t k0 = 0;
			bool stop = false;
			while(k0 < limit && !stop) {
				uint32_t n = limit - k0 < 32u ? limit - k0 : 32u;
				const uint32_t a = rdoff - k0 - n + 1;                               // the chunk's first view position (ascending inside the chunk)
				const uint64_t R = rc.chunk32((int64_t)rfoff + rflen - k0 - n), Q = seq.chunk32(a);
				uint64_t M = fh_diff32(R, Q, n);
				while(M) {
					const uint32_t j = (63u - (uint32_t)__builtin_clzll(M)) >> 1;     // the mismatch met first going left
					if(tmp_mm >= mm) { n = n - 1 - j; stop = true; break; }
					if(tmp_mm < FG_FE) nw |= (uint64_t)FE_MAKE_BP(a + j, (uint32_t)(R >> (2u * j)) & 3u, (uint32_t)(Q >> (2u * j)) & 3u, H2G_EDIT_MM) << (16u * tmp_mm);
					tmp_mm++;
					M &= ~(3ull << (2u * j));
				}
				k0 += n;
			}
			if(k0 > 0) { updated = true; extlen = k0; if(numNs) *numNs = 0; }
		} else if(wordwise) {
			uint32_t i = 0;
			bool stop = false;
			while(i < limit && !stop) {
				uint32_t n = limit - i < 32u ? limit - i : 32u;
				const uint64_t R = rc.chunk32((int64_t)rfoff + i), Q = seq.chunk32(rdoff + i);
				uint64_t M = fh_diff32(R, Q, n);
				while(M) {
					const uint32_t j = (uint32_t)__builtin_ctzll(M) >> 1;
					if(tmp_mm >= mm) { n = j; stop = true; break; }
					if(tmp_mm < FG_FE) nw |= (uint64_t)FE_MAKE_BP(i + j + rdoff_add, (uint32_t)(R >> (2u * j)) & 3u, (uint32_t)(Q >> (2u * j)) & 3u, H2G_EDIT_MM) << (16u * tmp_mm);
					tmp_mm++;
					M &= M - 1;
				}
				i += n;
			}
			if(i > 0) { updated = true; extlen = i; }
		} else if(left) {
			int i = (int)rdoff;
			for(int rf_i = (int)rflen - 1; rf_i >= 0 && i >= 0; rf_i--, i--) {
				const int64_t p = (int64_t)rfoff + rf_i;
				const int rf_bp = p < 0 ? 4 : rc.get(p), rd_bp = seq.at((uint32_t)i);
				if(rf_bp != rd_bp || rd_bp == 4) {
					if(tmp_mm >= mm) break;
					if(tmp_mm < FG_FE) nw |= (uint64_t)FE_MAKE_BP(i, rf_bp, rd_bp, H2G_EDIT_MM) << (16u * tmp_mm);
					tmp_mm++;
				}
				if(rf_bp == 4) nNs++;
			}
			if(i < (int)rdoff) { updated = true; extlen = rdoff - (uint32_t)i; if(numNs) *numNs = nNs; }
		} else {
			uint32_t i = 0;
			for(uint32_t rf_i = 0; rf_i < rflen && i < rdlen; rf_i++, i++) {
				const int64_t p = (int64_t)rfoff + rf_i;
				const int rf_bp = p < 0 ? 4 : rc.get(p), rd_bp = seq.at(rdoff + i);
				if(rf_bp != rd_bp || rd_bp == 4) {
					if(tmp_mm >= mm) break;
					if(tmp_mm < FG_FE) nw |= (uint64_t)FE_MAKE_BP(i + rdoff_add, rf_bp, rd_bp, H2G_EDIT_MM) << (16u * tmp_mm);
					tmp_mm++;
				}
			}
			if(i > 0) { updated = true; extlen = i; }
		}
	}
	if(!updated) tmp_mm = 0;
	const uint32_t total = n_old + tmp_mm;
	if(tmp_mm > FG_FE) { h.bad = 1; return extlen; }       // (the new edits beyond the fourth were not kept)
	const uint32_t n0 = (uint32_t)nw & 0xffffu;
	const uint32_t nlast = tmp_mm == 0 ? 0u : (uint32_t)(nw >> (16u * (tmp_mm - 1))) & 0xffffu;
	if(extlen > 0 && total > 0) {   // :751-779: front() / back() of the list the reference would hold
		const uint32_t old_first = FE_GET(h, 0), old_last = n_old == 0 ? 0u : FE_GET(h, n_old - 1);
		uint32_t f, b;
		if(left) { f = tmp_mm ? nlast : old_first; b = n_old ? old_last : n0; }
		else     { f = n_old ? old_first : n0;     b = tmp_mm ? nlast : old_last; }
		if(FE_POS(f) + extlen == base_rdoff + 1) {
			if(fe_is_gap(f)) extlen = 0;
			if(FE_TYPE(f) == H2G_EDIT_MM && FE_CCODE(f) == 4) extlen = 0;
		}
		if(extlen > 0 && FE_POS(b) == rdoff - base_rdoff + extlen - 1) { if(fe_is_gap(b)) extlen = 0; }
	}
	if(extlen > 0 && tmp_mm > 0) {   // commit the new edits
		if(total > FG_FE) { h.bad = 1; return extlen; }
		if(left) {                   // new edits go to the front, in increasing read position (they were met right to left); the old ones move up
			const uint64_t rev = (nw << 48) | ((nw & 0xffff0000ull) << 16) | ((nw >> 16) & 0xffff0000ull) | (nw >> 48);
			h.e = (tmp_mm >= FG_FE ? 0ull : h.e << (16u * tmp_mm)) | (rev >> (16u * (FG_FE - tmp_mm)));
#if FG_GRAPH
			{   // ... and their ALT ids with them (the new edits are plain mismatches)
				uint32_t o[FG_FE];
#pragma unroll
				for(uint32_t k = 0; k < FG_FE; k++) o[k] = h.snp[k];
#pragma unroll
				for(uint32_t k = 0; k < FG_FE; k++) {
					uint32_t v = H2G_MAX;
#pragma unroll
					for(uint32_t j = 0; j < FG_FE; j++) if(j + tmp_mm == k) v = o[j];
					h.snp[k] = v;
				}
			}
#endif
		} else {
			h.e |= nw << (16u * n_old);       // (n_old + tmp_mm <= FG_FE)
		}
		h.nedits = total;
	}
	if(extlen == 0 && numNs) *numNs = updated ? nNs : 0;
	return extlen;
}
// GenomeHit::extend hi_aligner.h:2031-2232 (extend_item)
H2G_HD void fh_extend(const DRef& ref, const DScoring& sc, const SeqView& seq, FHit& h, uint32_t mm, uint32_t max_leftext, uint32_t max_rightext,
                      uint32_t* leftext, uint32_t* rightext)
{
	const uint32_t rdlen = seq.len;
	*leftext = 0; *rightext = 0;
	if(max_leftext > 0 && h.rdoff > 0) {
		if(h.toff <= 0) return;
		int rl = (int)h.toff - (int)h.rdoff;
		uint32_t reflen = h.rdoff + 10;
		rl -= (int)(reflen - h.rdoff);
		if(rl < 0) { reflen += rl; rl = 0; }
		uint32_t numNs = 0;
		const uint32_t n_prev = h.nedits;
		const uint32_t best_ext = fh_align(ref, seq, h.rdoff - 1, h.rdoff - 1, h.rdoff, rl, reflen, true, h, mm, &numNs);
		if(h.bad) return;
		if(h.len == 0 && mm == 0 && h.nedits > 0) { h.nedits = 0; return; }
		if(best_ext > 0) {
			*leftext = best_ext;
			const uint32_t added = h.nedits - n_prev;
			h.rdoff -= best_ext; h.toff -= best_ext; h.len += best_ext; h.joff -= best_ext - numNs;   // (the new edits are mismatches: ref_ext == best_ext)
#pragma unroll
			for(uint32_t i = 0; i < FG_FE; i++) if(i < h.nedits) {
				const uint32_t e = FE_GET(h, i);
				const uint32_t pos = i < added ? FE_POS(e) - h.rdoff : FE_POS(e) + best_ext;
				if(pos > 255) h.bad = 1;
				FE_SET(h, i, (e & 0xffffff00u) | (pos & 0xffu));
			}
		}
	}
	if(max_rightext > 0 && h.rdoff + h.len < rdlen) {
		uint32_t r_rdoff, r_len, r_toff;
		fh_get_right(h, &r_rdoff, &r_len, &r_toff);
		const uint32_t rl = r_toff + r_len;
		const uint32_t rr = rdlen - (r_rdoff + r_len);
		const uint32_t tlen = ref.refLens[h.tidx];
		if(rl < tlen) {
			uint32_t reflen = rr + 10;
			if(rl + reflen > tlen) reflen = tlen - rl;
			const uint32_t best_ext = fh_align(ref, seq, h.rdoff, h.rdoff + h.len, rdlen - (h.rdoff + h.len), (int)rl, reflen, false, h, mm, nullptr);
			if(h.bad) return;
			if(h.len == 0 && mm == 0 && h.nedits > 0) { h.nedits = 0; return; }
			if(best_ext > 0) { *rightext = best_ext; h.len += best_ext; }
		}
	}
	fh_calc_score(sc, seq, h);
}
#if !FG_GRAPH      // (on a graph index joins are hit_combine of h2g_align.h over the unpacked hits: the rescan looks mismatches up in the ALT database)
// combineWith hi_aligner.h:1420-2025 (hit_combine) without spliced alignment, for two hits with the same read / reference offset
// difference: concatenation (:1506-1525) or the rescan of the joint for mismatches (:1880-1931).  An insertion or deletion between
// them is the general machine's: *indel is set and nothing else happens.
// the mismatch scores getRight / getLeft return with their parts (hit_get_right_sc, hit_get_left): behind the last / before the first gap
H2G_HD int64_t fh_part_score(const DScoring& sc, const SeqView& seq, const FHit& h, bool right) {
	int first = -1, last = -1;
#pragma unroll
	for(int i = 0; i < FG_FE; i++) if((uint32_t)i < h.nedits && fe_is_gap(FE_GET(h, i))) { if(first < 0) first = i; last = i; }
	int64_t score = 0;
#pragma unroll
	for(int i = 0; i < FG_FE; i++) if((uint32_t)i < h.nedits) {
		const uint32_t e = FE_GET(h, i);
		if(FE_TYPE(e) != H2G_EDIT_MM) continue;
		if(right ? i > last : (first < 0 || i < first))
			score += score_cell(sc, base_code((uint8_t)FE_QCHR(e)), base_code((uint8_t)FE_CHR(e)), seq.qual(h.rdoff + FE_POS(e)) - 33);
	}
	return score;
}
H2G_HD bool fh_combine(const DRef& ref, const DScoring& sc, const SeqView& seq, FHit& a, const FHit& b, int64_t minsc, bool* indel) {
	*indel = false;
	uint32_t this_rdoff, this_len, this_toff, other_rdoff, other_len, other_toff;
	fh_get_right(a, &this_rdoff, &this_len, &this_toff);
	fh_get_left(b, &other_rdoff, &other_len, &other_toff);
	if(this_len != 0 && other_len != 0 && this_rdoff + this_len > other_rdoff + other_len) return false;
	const uint32_t len = other_rdoff - this_rdoff + other_len;
	const uint32_t reflen = ref.refLens[a.tidx];
	if(this_toff + len > reflen) return false;
	const uint32_t refdif = other_toff - this_toff, rddif = other_rdoff - this_rdoff;
	if(refdif != rddif) {
		// an insertion or a deletion: the gap budget decides first (:1539-1560); a join that survives it is the general machine's
		int64_t remainsc = minsc - ((int64_t)a.score - fh_part_score(sc, seq, a, true)) - ((int64_t)b.score - fh_part_score(sc, seq, b, false));
		if(remainsc > 0) remainsc = 0;
		const int read_gaps = max_gaps(remainsc + sc.cp, sc.rdGapConst + sc.rdGapLinear, sc.rdGapLinear);
		const int ref_gaps = max_gaps(remainsc + sc.cp, sc.rfGapConst + sc.rfGapLinear, sc.rfGapLinear);
		if(refdif < rddif) { if((int64_t)refdif + ref_gaps < (int64_t)rddif) return false; }
		else { if((int64_t)rddif + read_gaps < (int64_t)refdif) return false; }
		*indel = true;
		return false;
	}
	if(this_rdoff + this_len == other_rdoff) {
		const uint32_t addoff = b.rdoff - a.rdoff;
#pragma unroll
		for(uint32_t i = 0; i < FG_FE; i++) if(i < b.nedits) {
			const uint32_t e = FE_GET(b, i), pos = FE_POS(e) + addoff;
			if(a.nedits >= FG_FE || pos > 255) { a.bad = 1; break; }
			FE_SET(a, a.nedits, (e & 0xffffff00u) | pos); a.nedits++;
		}
		a.len += b.len;
		fh_calc_score(sc, seq, a);
		return true;
	}
	// keep this hit's edits up to (and including) its last gap; drop the mismatches after it (:1818-1831)
	{
		int last = -1;
#pragma unroll
		for(int i = 0; i < FG_FE; i++) if((uint32_t)i < a.nedits && fe_is_gap(FE_GET(a, i))) last = i;
		a.nedits = (uint32_t)(last + 1);
	}
	{
		RefCursor rc1;
		rc1.init(&ref, a.tidx);
		const uint32_t addoff = this_rdoff - a.rdoff;
		if(seq.pk != nullptr && seq.pk_nomask && len > 0 && rc1.covers((int64_t)this_toff, len)) {   // word-wise, as in fh_align
			for(uint32_t i = 0; i < len && !a.bad; i += 32) {
				const uint32_t n = len - i < 32u ? len - i : 32u;
				const uint64_t R = rc1.chunk32((int64_t)this_toff + i), Q = seq.chunk32(this_rdoff + i);
				uint64_t M = fh_diff32(R, Q, n);
				while(M) {
					const uint32_t j = (uint32_t)__builtin_ctzll(M) >> 1;
					if(a.nedits >= FG_FE || i + j + addoff > 255) { a.bad = 1; break; }
					FE_SET(a, a.nedits, FE_MAKE_BP(i + j + addoff, (uint32_t)(R >> (2u * j)) & 3u, (uint32_t)(Q >> (2u * j)) & 3u, H2G_EDIT_MM)); a.nedits++;
					M &= M - 1;
				}
			}
		} else for(uint32_t i = 0; i < len; i++) {
			const int rdc = seq.at(this_rdoff + i), rfc = rc1.get((int64_t)this_toff + i);
			if(rdc != rfc) {
				if(a.nedits >= FG_FE || i + addoff > 255) { a.bad = 1; break; }
				FE_SET(a, a.nedits, FE_MAKE_BP(i + addoff, rfc, rdc, H2G_EDIT_MM)); a.nedits++;
			}
		}
	}
	if(!a.bad) {   // the other hit's edits from its first gap on
		int fsi = -1;
#pragma unroll
		for(int i = FG_FE - 1; i >= 0; i--) if((uint32_t)i < b.nedits && fe_is_gap(FE_GET(b, i))) fsi = i;
		const uint32_t addoff = b.rdoff - a.rdoff;
#pragma unroll
		for(int i = 0; i < FG_FE; i++) if(fsi >= 0 && i >= fsi && (uint32_t)i < b.nedits) {
			const uint32_t e = FE_GET(b, i), pos = FE_POS(e) + addoff;
			if(a.nedits >= FG_FE || pos > 255) { a.bad = 1; break; }
			FE_SET(a, a.nedits, (e & 0xffffff00u) | pos); a.nedits++;
		}
	}
	a.len = b.rdoff + b.len - a.rdoff;
	a.trim3 += b.trim3;
	fh_calc_score(sc, seq, a);
	return true;
}
#else
// a register hit <-> the general GenomeHit record the ALT-aware item functions work on
H2G_HD void fh_to_ghit(const FHit& h, h2g_ghit* g) {
	g->read = h.hitcount; g->fw = h.fw; g->rdoff = h.rdoff; g->len = h.len; g->trim5 = h.trim5; g->trim3 = h.trim3; g->tidx = h.tidx; g->toff = h.toff; g->joinedOff = h.joff;
	g->splicescore = 0; g->score = h.score; g->nedits = h.nedits; g->overflow = 0;
#pragma unroll
	for(uint32_t k = 0; k < FG_FE; k++) if(k < h.nedits) {
		const uint32_t e = FE_GET(h, k);
		h2g_edit o;
		o.pos = FE_POS(e); o.chr = (uint8_t)FE_CHR(e); o.qchr = (uint8_t)FE_QCHR(e); o.type = (uint8_t)FE_TYPE(e); o.pad = 0; o.snp = h.snp[k];
		g->edits[k] = o;
	}
}
H2G_HD uint32_t fe_code_of(uint8_t ch) { return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : ch == 'N' ? 4u : ch == '-' ? 5u : 7u; }
// (h.bad: the record does not fit the register form — more than FG_FE edits, an edit type / position / base outside it, a flagged record)
H2G_HD void ghit_to_fh(const h2g_ghit& g, FHit& h) {
	h.tidx = g.tidx; h.toff = g.toff; h.joff = g.joinedOff; h.rdoff = g.rdoff; h.len = g.len; h.trim5 = g.trim5; h.trim3 = g.trim3; h.fw = g.fw ? 1u : 0u;
	h.hitcount = g.read; h.nedits = g.nedits; h.e = 0; h.bad = 0;
	if(g.overflow || g.nedits > FG_FE || g.splicescore != 0 || g.score < -(int64_t)(1 << 30) || g.score > (int64_t)(1 << 30)) { h.bad = 1; h.nedits = 0; h.score = 0; }
	else h.score = (int32_t)g.score;
#pragma unroll
	for(uint32_t k = 0; k < FG_FE; k++) {
		h.snp[k] = H2G_MAX;
		if(k < h.nedits) {
			const h2g_edit o = g.edits[k];
			const uint32_t cc = fe_code_of(o.chr), qc = fe_code_of(o.qchr);
			if(o.pos > 255 || cc > 5 || qc > 5 || o.pad != 0 || !(o.type == H2G_EDIT_MM || o.type == H2G_EDIT_READ_GAP || o.type == H2G_EDIT_REF_GAP)) h.bad = 1;
			FE_SET(h, k, FE_MAKE_BP(o.pos & 0xffu, cc & 7u, qc & 7u, o.type & 3u));
			h.snp[k] = o.snp;
		}
	}
}
#endif

// small per-strand arrays by explicit selects (registers, no private-memory indexing)

H2G_HD SeqView fg_view(const FCtx& C, const FState& S, uint32_t set, bool fw) {
	// (selects, not indexing: an indexed array member would pin the whole context in private memory)
	const uint8_t* codes = set ? C.rd[1].codes : C.rd[0].codes;
	const char* quals = set ? C.rd[1].quals : C.rd[0].quals;
	const uint32_t ro = fs_ro(S, set);
	SeqView s;
	s.fwc = codes + ro; s.q = quals ? quals + ro : nullptr; s.len = fs_rl(S, set); s.fw = fw;
	s.pk = set ? C.pk[1] : C.pk[0]; s.pk_stride = C.pk_stride; s.pk_nomask = true;
	return s;
}
H2G_HD SeqView fg_sv(const FCtx& C, const FState& S) { return fg_view(C, S, S.paired ? S.sv_rdi : 0u, S.sv_fw != 0); }

// the long partial hits of strand x leave the pool (the strand is done)
H2G_HD void fg_pool_free(const FWords& W, FState& S, uint32_t x) {
	for(uint32_t k = 0; k < FG_NLONG; k++) { const uint32_t m = W.ld(FW_LONG + FG_LW * k + 2); if(m && ((m >> 20) & 3u) == x) W.st(FW_LONG + FG_LW * k + 2, 0); }
	if(S.pool_x) for(uint32_t k = FG_NLONG; k < FG_NLONGT; k++) if((S.pool_x >> (k - FG_NLONG)) & 1u) {
		const uint32_t m = W.ld(fg_pool(k) + 2);
		if(((m >> 20) & 3u) == x) { W.st(fg_pool(k) + 2, 0); S.pool_x = S.pool_x & ~(1u << (k - FG_NLONG)); }
	}
}
// packs read i of `rd` for the fast path: 8 words of 2-bit codes (word k of this lane at pk[k * stride]).  false: an N, or longer than 128 bases.
H2G_HD bool fg_pack_read(const DReads& rd, uint32_t i, uint32_t* pk, uint32_t stride) {
	const uint32_t ro = rd.offs[i], rl = rd.offs[i + 1] - ro;
	if(rl > H2G_PK_MAXLEN) return false;
	const uint8_t* src = rd.codes + ro;
	uint32_t anyn = 0;
	for(uint32_t w = 0; w < H2G_PK_WORDS; w++) {
		uint32_t bits = 0;
		if(w * 16 < rl) {
			for(uint32_t q = 0; q < 4; q++) {
				const uint32_t at = w * 16 + q * 4;
				if(at >= rl) break;
				uint32_t four;
				memcpy(&four, src + at, 4);                       // (the code buffer is padded past its last read)
				const uint32_t left = rl - at;
				if(left < 4) four &= (1u << (8 * left)) - 1u;
				anyn |= four & 0x04040404u;
				const uint32_t lo = four & 0x03030303u;
				bits |= ((lo | (lo >> 6) | (lo >> 12) | (lo >> 18)) & 0xffu) << (8 * q);
			}
		}
		pk[w * stride] = bits;
	}
	return anyn == 0;
}
// result summaries: tidx, toff, fw | nedits << 1 | extent << 4 | (score + 32768) << 16
H2G_HD uint32_t fg_res_base(uint32_t m, uint32_t k) { return FW_RES + (m * FG_NRES + k) * 3; }
// ... and the hit behind each summary.  A completed read's output (records + ReadOut / PairOut) is written in ONE place, FPC_FINISH, as a pure
// function of the read: runs queued back to back share the result rows, and a machine pass of an earlier run may write the same read's rows
// at the same time — with identical bytes, as long as nobody reads rows back (profiles/r04_NOTES.md: the round-3 inequality was an in-place
// swap of two rows here against exactly such a pass)
H2G_HD uint32_t fg_res_hit(uint32_t m, uint32_t k) { return FW_RH + (m * FG_NRES + k) * FG_HW; }

// genRandSeed (pat.h:55-91, gen_rand_seed of h2g_align.h) of a read without N from its packed words: the 2-bit codes of 16 bases
// XOR into the seed exactly as they lie in a packed word
H2G_HD uint32_t fg_rand_seed(const FCtx& C, const FState& S, uint32_t set) {
	uint32_t rseed = (0u + 101u) * 59u * 61u * 67u * 71u * 73u * 79u * 83u;
	const uint32_t len = fs_rl(S, set);
	const uint32_t* pk = set ? C.pk[1] : C.pk[0];
	for(uint32_t w = 0; w < H2G_PK_WORDS; w++) rseed ^= pk[w * C.pk_stride];
	const char* quals = set ? C.rd[1].quals : C.rd[0].quals;
	if(quals) {
		const char* q = quals + fs_ro(S, set);
		for(uint32_t i = 0; i < len; i++) rseed ^= ((uint32_t)q[i] << ((i & 3) << 3));
	} else {
		for(uint32_t b = 0; b < 4; b++) if(((len + 3 - b) >> 2) & 1u) rseed ^= (uint32_t)'I' << (b << 3);
	}
	const char* name = set ? C.name[1] : C.name[0];
	const uint32_t namelen = set ? C.namelen[1] : C.namelen[0];
	for(uint32_t i = 0; i < namelen; i++) {
		const int p = name[i];
		if(p == '/') break;
		rseed ^= ((uint32_t)p << ((i & 3) << 3));
	}
	return rseed;
}

// The worker-loop prelude (mach_begin; hisat2.cpp:3380-3530).  `ok0/ok1`: the mates were packed without an N and have 32..128 bases.
H2G_HD void fast_begin(const FCtx& C, FState& S, uint32_t read, bool paired, bool packed_ok) {
	S.read = read; S.op = FOP_NONE; S.bail = FB_NONE; S.paired = paired ? 1u : 0u; S.nm = paired ? 2u : 1u;
	// every field has a defined value from here on: the packed form of the state (fs_pack) gives each its bits
	S.a0 = S.a1 = S.a2 = S.a3 = S.a4 = S.a5 = 0;
	S.sel_r = S.sel_f = S.nb_rdi = S.nb_fwi = 0; S.sv_rdi = S.sv_fw = 0;
	S.gh_hi = S.gh_hj = S.gh_nco = S.gh_rdoff = S.hs_hi = S.hs_hj = S.hs_found = 0;
	S.sp = -1; S.rc_minsc = 0; S.ret = 0; S.rc_ret_pc = S.pr_ret_pc = 0;
	S.f_hitoff = S.f_hitlen = S.f_extoff = S.f_extlen = S.f_lidx = S.f_state = S.f_count = S.f_ncoords = 0; S.f_ri = 0; S.f_maxsc = S.f_prev = 0;
	S.f_success = S.f_first = S.f_uselocal = S.f_unique = 0; S.f_top = S.f_bot = S.f_nelt = 0; S.f_noext = 0; S.f_maxHitLen = 0; S.f_nlocal = 0; S.f_ti = 0;
	S.rb_cur = S.rb_nps = S.rb_nus = S.rb_np = 0; S.rb_sumsq0 = S.rb_sumsq1 = S.rb_sumsq2 = S.rb_sumsq3 = 0;
	S.rb_done = S.rb_nonempty = S.found = 0; S.rnd = 0; S.ro0 = S.ro1 = 0; S.rl0 = S.rl1 = 0;
	S.nres0 = S.nres1 = S.nsearched0 = S.nsearched1 = 0; S.bestUnp0 = S.bestUnp1 = S.best2Unp0 = S.best2Unp1 = F_SMIN16; S.minsc0 = S.minsc1 = F_SMAX16;
	S.pad23_ = 0; S.pad27_ = 0; S.rc_mate = 0; S.pool_x = 0; S.pad30_ = 0; S.pad35_ = 0;
#if FG_GRAPH
	S.a6 = S.a7 = S.a8 = 0; S.gh_k = S.gh_gsize = 0; S.pad43_ = 0; S.f_ntop = S.f_nbot = 0; S.f_ie = 0; S.pad46_ = S.pad47_ = 0;
#endif
	S.npairs = S.pairs = S.insp_i = S.insp_j = 0; S.bestPair = S.best2Pair = F_SMIN;
	S.nrank = S.nside = S.nsteps = S.nframes_max = 0; S.nghits = S.ghit_done = 0; S.localindexatts = S.max_localindexatts = 0;
	S.ro0 = C.rd[0].offs[read]; S.ro1 = C.rd[1].offs[read];                     // (unpaired: rd[1] is rd[0])
	const uint32_t len0 = C.rd[0].offs[read + 1] - S.ro0, len1 = C.rd[1].offs[read + 1] - S.ro1;
	if(!packed_ok || len0 < 32 || len0 > 128 || (paired && (len1 < 32 || len1 > 128))) { S.pc = FPC_BAIL; S.bail = FB_INPUT; return; }
	S.rl0 = len0; S.rl1 = len1;
	// no N, length >= 2: both filters pass (read_passes_filters)
	Rng rnd;
	uint32_t seed = fg_rand_seed(C, S, 0);
	if(paired) seed ^= fg_rand_seed(C, S, 1);                  // hisat2.cpp:3463-3468
	rnd.init(seed);
	S.rnd = rnd.last;
	S.pc = FPC_GO_INIT;
}

#define F_GOTO(NEXT) do { S.pc = (NEXT); goto again; } while(0)
#define F_OP(OPC, NEXT) do { S.op = (OPC); S.pc = (NEXT); return; } while(0)
#define F_BAIL(WHY) do { S.pc = FPC_BAIL; S.bail = (WHY); S.op = FOP_NONE; return; } while(0)
#define F_RC_RET(V) do { S.ret = (V); S.sp--; if(S.sp >= 0) { fg_frame_restore(W, S); F_GOTO(S.f_state); } else F_GOTO(S.rc_ret_pc); } while(0)
#define F_RC_CALL(HB, HOFF, HLEN, RESUME) do { \
		const uint32_t hoff_ = (HOFF), hlen_ = (HLEN); \
		S.f_state = (RESUME); \
		if(S.sp + 1 >= FG_NFRAME) F_BAIL(FB_DEPTH); \
		fg_frame_save(W, S); fg_hit_copy(W, fg_frame_hit(S.sp + 1), (HB)); S.sp++; S.f_hitoff = hoff_; S.f_hitlen = hlen_; \
		if((uint32_t)S.sp + 1 > S.nframes_max) S.nframes_max = (uint32_t)S.sp + 1; \
		F_GOTO(FPC_RC_ENTRY); } while(0)
#if FG_ALIGN_MATE
#define F_CUSHION() (S.rc_mate ? (int32_t)((double)fs_rl(S, S.sv_rdi) * 0.03 * (double)sc.mmpMax) : 0)     /* spliced_aligner.h:363-366 */
// (with the mate cushion and no alignment of this mate yet the reference computes numeric_limits<TAlScore>::min() - cushion, which wraps
// to a huge positive score: nothing passes that bar — spliced_aligner.h:960, :1106 as compiled; the general machine does the same in 64 bits)
#define F_MINSC_LIVE(MV) do { const int32_t c_ = F_CUSHION(), u_ = fs_bestUnp(S, S.sv_rdi); const int32_t b_ = (u_ == F_SMIN16 && c_ > 0) ? 0x7fffffff : u_ - c_; if(b_ > (MV)) (MV) = b_; } while(0)
#else
#define F_CUSHION() 0
#define F_MINSC_LIVE(MV) do { const int32_t b_ = fs_bestUnp(S, S.sv_rdi); if(b_ > (MV)) (MV) = b_; } while(0)
#endif

H2G_HD void fg_frame_save(const FWords& W, const FState& S) {
	const uint32_t w0 = (uint32_t)S.f_hitoff | ((uint32_t)S.f_hitlen << 8) | ((uint32_t)S.f_extoff << 16) | ((uint32_t)S.f_extlen << 24);
	const uint32_t w4 = (uint32_t)S.f_state | ((uint32_t)S.f_count << 8) | ((uint32_t)S.f_ncoords << 10) | ((uint32_t)(S.f_ri + 1) << 13) | ((uint32_t)S.f_success << 17) |
	                    ((uint32_t)S.f_first << 18) | ((uint32_t)S.f_uselocal << 19) | ((uint32_t)S.f_unique << 20) | ((uint32_t)S.f_nlocal << 21) | ((uint32_t)S.f_ti << 23);
	const uint32_t v[FG_FRS] = {w0, (uint32_t)S.f_maxsc, (uint32_t)S.f_prev, S.f_lidx, w4};
	W.stv<FG_FRS>(fg_frame_base(S.sp), v);
}
H2G_HD void fg_frame_restore(const FWords& W, FState& S) {
	uint32_t v[FG_FRS];
	W.ldv<FG_FRS>(fg_frame_base(S.sp), v);
	const uint32_t w0 = v[0], w4 = v[4];
	S.f_hitoff = w0 & 0xffu; S.f_hitlen = (w0 >> 8) & 0xffu; S.f_extoff = (w0 >> 16) & 0xffu; S.f_extlen = w0 >> 24;
	S.f_maxsc = (int32_t)v[1]; S.f_prev = (int32_t)v[2]; S.f_lidx = v[3];
	S.f_state = w4 & 0xffu; S.f_count = (w4 >> 8) & 3u; S.f_ncoords = (w4 >> 10) & 7u; S.f_ri = (int32_t)((w4 >> 13) & 15u) - 1;
	S.f_success = (w4 >> 17) & 1u; S.f_first = (w4 >> 18) & 1u; S.f_uselocal = (w4 >> 19) & 1u; S.f_unique = (w4 >> 20) & 1u;
	S.f_nlocal = (w4 >> 21) & 3u; S.f_ti = (w4 >> 23) & 3u;
}

// sort_coords (Coord::operator< ref_coord.h:79: by text, then offset) over a frame's list
H2G_HD void fg_sort_coords(const FWords& W, uint32_t cb, uint32_t n) {
	for(uint32_t i = 1; i < n; i++) {
		const uint32_t xt = W.ld(cb + 3 * i), xo = W.ld(cb + 3 * i + 1), xj = W.ld(cb + 3 * i + 2);
		int j = (int)i - 1;
		while(j >= 0) {
			const uint32_t jt = W.ld(cb + 3 * j), jo = W.ld(cb + 3 * j + 1);
			if(!(jt > xt || (jt == xt && jo > xo))) break;
			W.st(cb + 3 * (j + 1), jt); W.st(cb + 3 * (j + 1) + 1, jo); W.st(cb + 3 * (j + 1) + 2, W.ld(cb + 3 * j + 2));
			j--;
		}
		W.st(cb + 3 * (j + 1), xt); W.st(cb + 3 * (j + 1) + 1, xo); W.st(cb + 3 * (j + 1) + 2, xj);
	}
}
// reportHit + AlnSinkWrap::report (al_report of h2g_align.h) for a full-length hit; the record goes straight to its output slot
H2G_HD void fg_write_rec(h2g_alnres& d, const FHit& hit, uint32_t rdlen) {
	d.fw = hit.fw; d.tidx = hit.tidx; d.toff = hit.toff; d.len = hit.len; d.trim5 = hit.trim5; d.trim3 = hit.trim3;
	d.nedits = hit.nedits; d.splicescore = 0; d.score = (int64_t)hit.score;
	const uint32_t trim5p = hit.fw ? hit.trim5 : hit.trim3;
#pragma unroll
	for(uint32_t k = 0; k < FG_FE; k++) if(k < hit.nedits) {
		const uint32_t e = hit.fw ? FE_GET(hit, k) : FE_GET(hit, hit.nedits - 1 - k);
		uint32_t pos = FE_POS(e) + hit.trim5;
		if(!hit.fw) pos = FE_TYPE(e) == H2G_EDIT_READ_GAP ? rdlen - pos : rdlen - pos - 1;   // Edit::invertPoss edit.cpp:70-111
		pos -= trim5p;
		h2g_edit o;
		o.pos = pos; o.chr = (uint8_t)FE_CHR(e); o.qchr = (uint8_t)FE_QCHR(e); o.type = (uint8_t)FE_TYPE(e); o.pad = 0; o.snp = H2G_MAX;
#if FG_GRAPH
		o.snp = hit.fw ? hit.snp[k] : hit.snp[hit.nedits - 1 - k];
#endif
		d.edits[k] = o;
	}
}
H2G_HD uint32_t fg_ref_extent(const FHit& h) {
	uint32_t ext = h.len;
#pragma unroll
	for(uint32_t k = 0; k < FG_FE; k++) if(k < h.nedits) {
		const uint32_t t = FE_TYPE(FE_GET(h, k));
		if(t == H2G_EDIT_READ_GAP) ext++; else if(t == H2G_EDIT_REF_GAP) ext--;
	}
	return ext;
}
H2G_HD int64_t fg_hisat2_key(int64_t score, uint32_t trim) {      // hisat2_score without splices (h2g_align.h)
	int64_t t = trim > 0xffff ? 0 : 0xffff - (int64_t)trim;
	return (int64_t)((uint64_t)score << 32) | ((int64_t)255 << 16) | t;
}

// Runs the control flow of this lane until it needs a primitive (S.op != FOP_NONE), completes (FPC_DONE) or gives up (FPC_BAIL).
H2G_HD void fast_step(const FCtx& C, FState& S, const FWords& W)
{
	const AlnParams& P = *C.P;
	const DScoring& sc = P.sc;
	const uint32_t minK = C.g->minK, minK_local = P.minK_local;
	const uint32_t maxsz = P.khits > P.kseeds ? P.khits : P.kseeds;
again:
	switch(S.pc) {
	// ======================================================================== go() hi_aligner.h:4048 / nextBWT :4644
	case FPC_GO_INIT: {
		S.nrank = S.nside = S.nsteps = S.nframes_max = 0;
		S.npairs = 0; S.pairs = 0; S.insp_i = S.insp_j = 0; S.bestPair = F_SMIN; S.best2Pair = F_SMIN;
		S.localindexatts = 0; S.max_localindexatts = 0;
		S.nghits = 0; S.ghit_done = 0;
		S.nres0 = S.nres1 = S.nsearched0 = S.nsearched1 = 0; S.bestUnp0 = S.bestUnp1 = S.best2Unp0 = S.best2Unp1 = F_SMIN16; S.minsc0 = S.minsc1 = F_SMAX16;
		{
			const int64_t m0 = min_score_for(P, S.rl0), m1 = S.nm > 1 ? min_score_for(P, S.rl1) : 0;
			if(m0 < -30000 || m1 < -30000) F_BAIL(FB_INPUT);        // (the scores are kept in 16 bits)
			S.minsc0 = (int32_t)m0; if(S.nm > 1) S.minsc1 = (int32_t)m1;
		}
		S.rb_cur = S.rb_nps = S.rb_nus = S.rb_np = 0; S.rb_sumsq0 = S.rb_sumsq1 = S.rb_sumsq2 = S.rb_sumsq3 = 0;
		S.rb_done = 0; S.rb_nonempty = 0;
		S.found = S.paired ? 15u : 3u;                           // found[0][0], [0][1], [1][0], [1][1]
		for(uint32_t k = 0; k < FG_NLONG; k++) W.st(FW_LONG + FG_LW * k + 2, 0);
		S.pool_x = 0;
		F_GOTO(FPC_NB_PICK);
	}
	case FPC_NB_PICK: {                                   // one iteration of nextBWT's loop (:4644-4760)
		int rdi = -1, fwi = -1;
		int64_t maxScore = INT64_MIN;
		for(uint32_t r = 0; r < S.nm; r++) for(int k = 0; k < 2; k++) {
			const uint32_t x = r * 2 + (uint32_t)k;
			if((S.rb_done >> x) & 1u) continue;
			const uint32_t act = fs_b4(S.rb_nps, x) - fs_b4(S.rb_nus, x);
			int64_t cs = (int64_t)fs_sumsq(S, x) - (int64_t)act * minK * minK - ((int64_t)1 << (act << 1));   // ReadBWTHit::searchScore :320
			if(fs_b4(S.rb_cur, x) == 0) cs = INT64_MAX;
			if(cs > maxScore) { maxScore = cs; rdi = (int)r; fwi = k; }
		}
		if(rdi < 0) F_GOTO(FPC_AFTER_LOOP);
		const uint32_t x = (uint32_t)rdi * 2 + (uint32_t)fwi, xr = (uint32_t)rdi * 2 + (uint32_t)(1 - fwi);
		{
			const uint32_t numSearched = fs_b4(S.rb_nps, x) - fs_b4(S.rb_nus, x);
			const int32_t bestScore = fs_bestUnp(S, rdi), msc = fs_minsc(S, rdi);
			if(bestScore >= msc) {
				const uint32_t maxmm = (uint32_t)((-(int64_t)bestScore + sc.mmpMax - 1) / sc.mmpMax);
				if(numSearched > maxmm + 1) {
					S.rb_done |= 1u << x; fg_pool_free(W, S, x);
					if(S.paired) {
						const int32_t ob = fs_bestUnp(S, 1 - rdi), om = fs_minsc(S, 1 - rdi);
						if(ob >= om && S.npairs > 0) F_GOTO(FPC_AFTER_LOOP); else F_GOTO(FPC_NB_PICK);
					} else F_GOTO(FPC_AFTER_LOOP);
				}
			}
			if(((S.rb_done >> xr) & 1u) && bestScore < msc) {
				const uint32_t rcs = fs_b4(S.rb_nps, xr) - fs_b4(S.rb_nus, xr);
				if(numSearched > rcs + (P.anchorStop ? 1u : 0u)) { S.rb_done |= 1u << x; fg_pool_free(W, S, x); F_GOTO(FPC_AFTER_LOOP); }
			}
		}
		S.nb_rdi = rdi; S.nb_fwi = fwi;
		S.sv_rdi = (uint32_t)rdi; S.sv_fw = fwi == 0;
		S.a0 = fs_b4(S.rb_cur, x);
		F_OP(FOP_PSEARCH, FPC_NB_AFTER_PS);
	}
	case FPC_NB_AFTER_PS: {
		// a0 top, a1 bot, a2 len | hit_type << 8 | done << 16 | anchorStop << 17 | numUniqueSearch << 18, a3 cur, a4 nrank | nside << 16, a5 bwoff
		const int rdi = S.nb_rdi, fwi = S.nb_fwi;
		const uint32_t x = (uint32_t)rdi * 2 + (uint32_t)fwi;
		const uint32_t top = S.a0, bot = S.a1, len = S.a2 & 0xffu, type = (S.a2 >> 8) & 0xffu, done = (S.a2 >> 16) & 1u, anchor = (S.a2 >> 17) & 1u, nus = S.a2 >> 18;
		{ const uint32_t nr_ = S.nrank + (S.a4 & 0xffffu), ns_ = S.nside + (S.a4 >> 16); if(nr_ > 0xffffu || ns_ > 0xffffu) F_BAIL(FB_OTHER); S.nrank = nr_; S.nside = ns_; }
		S.rb_nps = fs_b4_set(S.rb_nps, x, fs_b4(S.rb_nps, x) + 1); S.rb_nus = fs_b4_set(S.rb_nus, x, fs_b4(S.rb_nus, x) + nus); S.rb_cur = fs_b4_set(S.rb_cur, x, S.a3);
		const uint32_t np = fs_b4(S.rb_np, x);
		if(np >= 16) F_BAIL(FB_PARTIAL);                    // AL_MAX_PARTIAL of the default workspace
		fs_sumsq_add(S, x, len * len);
		if(bot > top && bot != H2G_MAX) {
			S.rb_nonempty |= 1u << x;
			if(len > minK + 2) {                            // getAnchorHits looks at these only (:5033)
				uint32_t k = 0;
				for(; k < FG_NLONG; k++) if(W.ld(FW_LONG + FG_LW * k + 2) == 0) break;
				if(k >= FG_NLONG) {                             // the hot entries are taken: a cold one
					for(; k < FG_NLONGT; k++) if(!((S.pool_x >> (k - FG_NLONG)) & 1u)) break;
					if(k >= FG_NLONGT) F_BAIL(FB_LONGPOOL);
					S.pool_x = S.pool_x | (1u << (k - FG_NLONG));
				}
				W.st(fg_pool(k), top); W.st(fg_pool(k) + 1, bot);
#if FG_GRAPH
				if(S.a8 == FG_IE_NOFIT) F_BAIL(FB_IEDGES);
				{ const uint32_t pn[3] = {S.a6, S.a7, S.a8}; W.stv<3>(FW_LONGX + 3 * k, pn); }   // pnode: node range, in-edges
#endif
				W.st(fg_pool(k) + 2, 0x80000000u | S.a5 | (len << 8) | (type << 16) | (x << 20) | (np << 24));   // bwoff, len, type, strand, index
			}
		}
		S.rb_np = fs_b4_set(S.rb_np, x, np + 1);
		if(done) { S.rb_done |= 1u << x; S.sel_r = rdi; S.sel_f = fwi; F_GOTO(FPC_ALIGN); }
		{ const uint32_t cur = fs_b4(S.rb_cur, x); if(cur + 1 < fs_rl(S, rdi)) S.rb_cur = fs_b4_set(S.rb_cur, x, cur + 1); }   // !pseudogeneStop (never set without spliced alignment)
		if(anchor) { S.rb_done |= 1u << x; S.sel_r = rdi; S.sel_f = fwi; F_GOTO(FPC_ALIGN); }
		F_GOTO(FPC_NB_PICK);
	}
	// ======================================================================== align() :5484-5573
	case FPC_ALIGN: {
		S.sv_rdi = (uint32_t)S.sel_r; S.sv_fw = S.sel_f == 0;
		const uint32_t x = (uint32_t)S.sel_r * 2 + (uint32_t)S.sel_f;
		if(!((S.rb_nonempty >> x) & 1u)) { S.hs_found = 0; F_GOTO(FPC_AFTER_ALIGN); }
		int32_t bestScore = fs_bestUnp(S, S.sel_r);
		const int32_t msc = fs_minsc(S, S.sel_r);
		if(bestScore < msc) bestScore = msc;
		const uint32_t maxmm = (uint32_t)((-(int64_t)bestScore + sc.mmpMax - 1) / sc.mmpMax);
		const uint32_t nact = fs_b4(S.rb_nps, x) - fs_b4(S.rb_nus, x);
		if(nact > maxmm + 1) { S.hs_found = 1; F_GOTO(FPC_AFTER_ALIGN); }
		S.nghits = 0; S.gh_hi = 0;
		F_GOTO(FPC_GAH_LOOP);
	}
	case FPC_AFTER_ALIGN: {
		// the strand is done for good (align() runs once per strand): its long partial hits leave the pool
		{
			const uint32_t x = (uint32_t)S.sel_r * 2 + (uint32_t)S.sel_f;
			fg_pool_free(W, S, x);
			if(S.hs_found) S.found |= 1u << x; else S.found &= ~(1u << x);
		}
		if(S.found == 0) F_GOTO(FPC_AFTER_LOOP);
		if(S.paired) { S.pr_ret_pc = FPC_NB_PICK; F_GOTO(FPC_PAIR_READS); }
		F_GOTO(FPC_NB_PICK);
	}
	case FPC_PAIR_READS: {                                // pairReads hi_aligner.h:5948-6055 (al_pair_reads) over the summaries
		const uint32_t start_i = S.insp_i, start_j = S.insp_j;
		S.insp_i = S.nres0; S.insp_j = S.nres1;
		for(uint32_t i = 0; i < S.nres0; i++) {
			for(uint32_t j = (i >= start_i ? 0 : start_j); j < S.nres1; j++) {
				const uint32_t b1 = fg_res_base(0, i), b2 = fg_res_base(1, j);
				const uint32_t t1 = W.ld(b1), t2 = W.ld(b2);
				if(t1 != t2) continue;
				const uint32_t o1 = W.ld(b1 + 1), o2 = W.ld(b2 + 1), m1 = W.ld(b1 + 2), m2 = W.ld(b2 + 2);
				const bool fw1 = (m1 & 1u) != 0, fw2 = (m2 & 1u) != 0;
				const uint32_t e1 = (m1 >> 4) & 0xfffu, e2 = (m2 >> 4) & 0xfffu;
				const int32_t s1 = (int32_t)(m1 >> 16) - 32768, s2 = (int32_t)(m2 >> 16) - 32768;
				int64_t l = o1, r = (int64_t)o1 + e1 - 1, l2 = o2, rr2 = (int64_t)o2 + e2 - 1;
				if(fw1) { if(fw2) continue; }
				else {
					if(!fw2) continue;
					int64_t t = l; l = l2; l2 = t; t = r; r = rr2; rr2 = t;
				}
				if(l > l2) continue;
				if(r > rr2) continue;
				if(r + (int64_t)P.maxIntronLen < l2) continue;
				bool pass;
				if(o1 < o2) pass = pe_concordant(o1, e1, fw1, o2, e2, fw2, P.maxFragLen);
				else        pass = pe_concordant(o2, e2, fw2, o1, e1, fw1, P.maxFragLen);
				if(pass) {
					int64_t threshold = S.bestPair == F_SMIN ? INT64_MIN : (int64_t)S.bestPair;
					if(S.bestUnp0 != F_SMIN16 && S.bestUnp1 != F_SMIN16 && S.bestUnp0 >= S.minsc0 && S.bestUnp1 >= S.minsc1) {
						const int64_t tmp = (int64_t)((double)((int64_t)S.bestUnp0 + S.bestUnp1) - (double)(S.rl0 + S.rl1) * 0.03 * (double)sc.mmpMax);
						if(tmp > threshold) threshold = tmp;
					}
					const int32_t score = s1 + s2;
					if((int64_t)score >= threshold) {
						if(S.npairs >= FG_NPAIR) F_BAIL(FB_NPAIRS);
						S.pairs |= (i | (j << 2)) << (4 * S.npairs); S.npairs++;
						if(S.bestPair == F_SMIN || score > S.bestPair) { S.best2Pair = S.bestPair; S.bestPair = score; }
						else if(S.best2Pair == F_SMIN || score > S.best2Pair) S.best2Pair = score;
					}
				}
			}
		}
		F_GOTO(S.pr_ret_pc);
	}
	case FPC_AFTER_LOOP: {
		// no concordant pair but an aligned mate: alignMate (hi_aligner.h:4092-4148) is the general machine's
		if(S.paired && S.npairs == 0 && ((S.bestUnp0 != F_SMIN16 && S.bestUnp0 >= S.minsc0) || (S.bestUnp1 != F_SMIN16 && S.bestUnp1 >= S.minsc1))) {
#if FG_ALIGN_MATE
			const uint32_t am[4] = {(uint32_t)fs_nres(S, 0) << 3 | (uint32_t)fs_nres(S, 1) << 5, H2G_MAX, 0u, 0u};   // mp_i = mp_j = 0, mate_found = 0
			W.stv<4>(FW_AM, am);
			F_GOTO(FPC_MP_LOOP);
#else
			if(C.mate_handover) F_OP(FOP_HANDOVER, FPC_AFTER_LOOP);      // (the alignMate build re-enters this state and takes the branch above)
			F_BAIL(FB_MATE);
#endif
		}
		F_GOTO(FPC_FINISH);
	}
#if FG_ALIGN_MATE
	// ======================================================================== alignMate hi_aligner.h:4092-4148, :5579-5770
	// Each alignment of a mate anchors a search for the OTHER mate near it: the local index at the anchor (and its neighbour in the
	// anchor's direction), backward searches from the read's end, hits within 2 x maxFragLen of the anchor, hybridSearch_recur with
	// the mate cushion.  Loop state (FAm) in four cold words: w0 = mp_i | mp_j << 1 | mp_rs0 << 3 | mp_rs1 << 5 | mate_found << 7 |
	// first << 8 | count << 9 | hi << 11 | ri << 13 | nco << 16 | fw << 19; w1 = the local index; w2 = hitoff | hitlen << 8 | maxhitlen << 16.
#define FAM_LOAD() uint32_t am[4]; W.ldv<4>(FW_AM, am)
#define FAM_STORE() W.stv<4>(FW_AM, am)
#define FAM_GET(WORD, SH, BITS) ((am[WORD] >> (SH)) & ((1u << (BITS)) - 1u))
#define FAM_SET(WORD, SH, BITS, V) (am[WORD] = (am[WORD] & ~(((1u << (BITS)) - 1u) << (SH))) | (((uint32_t)(V) & ((1u << (BITS)) - 1u)) << (SH)))
	case FPC_MP_LOOP: {
		FAM_LOAD();
		uint32_t mp_i = FAM_GET(0, 0, 1), mp_j = FAM_GET(0, 1, 2);
		// (mp_i needs the value 2 as well: kept as mp_i | done << 20)
		if(FAM_GET(0, 20, 1)) {
			if(FAM_GET(0, 7, 1)) { S.pr_ret_pc = FPC_FINISH; F_GOTO(FPC_PAIR_READS); }
			F_GOTO(FPC_FINISH);
		}
		const uint32_t rs = mp_i == 0 ? FAM_GET(0, 3, 2) : FAM_GET(0, 5, 2);
		if(mp_j >= rs) {
			if(mp_i == 1) FAM_SET(0, 20, 1, 1); else FAM_SET(0, 0, 1, 1);
			FAM_SET(0, 1, 2, 0);
			FAM_STORE();
			F_GOTO(FPC_MP_LOOP);
		}
		const uint32_t rb = fg_res_base(mp_i, mp_j);
		uint32_t r3[3];
		W.ldv<3>(rb, r3);
		const bool fw = (r3[2] & 1u) != 0;
		FAM_SET(0, 19, 1, fw ? 1u : 0u);
		S.sv_rdi = 1u - mp_i; S.sv_fw = fw ? 0u : 1u;               // ofw = (fw == gMate2fw ? gMate1fw : gMate2fw) = !fw with --fr (:5605)
		S.nghits = 0; S.ghit_done = 0;
		am[1] = local_index_of(*C.ls, r3[0], r3[1]);
		FAM_SET(0, 8, 1, 1); FAM_SET(0, 9, 2, 0);
		am[2] = 0;                                               // hitoff, hitlen, maxhitlen
		FAM_STORE();
		F_GOTO(FPC_AM_WHILE);
	}
	case FPC_AM_WHILE: {
		FAM_LOAD();
		const uint32_t count = FAM_GET(0, 9, 2);
		FAM_SET(0, 9, 2, count < 3 ? count + 1 : 3);
		bool ext = !(count < 2);
		if(!ext) {
			if(FAM_GET(0, 8, 1)) FAM_SET(0, 8, 1, 0);
			else {
				if(S.nghits > 0) ext = true;
				else {
					if(am[1] != H2G_MAX) am[1] = FAM_GET(0, 19, 1) ? local_index_next(*C.ls, am[1]) : local_index_prev(*C.ls, am[1]);
					if(am[1] == H2G_MAX || C.ls->desc[am[1]].len == 0) ext = true;
				}
			}
			if(!ext && am[1] == H2G_MAX) ext = true;
		}
		if(ext) { FAM_SET(0, 11, 2, 0); FAM_STORE(); F_GOTO(FPC_AM_EXT_LOOP); }
		am[2] = (am[2] & ~0xffu) | ((fs_rl(S, S.sv_rdi) - 1u) & 0xffu);
		FAM_STORE();
		F_GOTO(FPC_AM_INNER);
	}
	case FPC_AM_INNER: {
		FAM_LOAD();
		const uint32_t hitoff = am[2] & 0xffu;
		if(!(hitoff >= minK_local - 1)) F_GOTO(FPC_AM_WHILE);
		if(C.ls->desc[am[1]].len == 0) { S.a0 = 0; S.a1 = 0; S.a2 = H2G_MAX; S.a3 = H2G_MAX; S.a4 = 0; F_GOTO(FPC_AM_AFTER_LS); }
		S.a0 = am[1]; S.a1 = hitoff; S.a2 = 0xffffu; S.a3 = 0; S.a4 = H2G_MAX; S.a5 = H2G_MAX;
		F_OP(FOP_LSEARCH, FPC_AM_AFTER_LS);
	}
	case FPC_AM_AFTER_LS: {
		FAM_LOAD();
		const uint32_t nelt = S.a0, hitlen = S.a1, top = S.a2, bot = S.a3, hitoff = am[2] & 0xffu, maxhitlen = (am[2] >> 16) & 0xffu;
		if(hitlen > 255) F_BAIL(FB_OTHER);
		am[2] = (am[2] & ~0xff00u) | (hitlen << 8);
		FAM_STORE();
		if(nelt > 0 && nelt <= P.kseeds && hitlen > maxhitlen) {
			if(bot - top > FG_NCO) F_BAIL(FB_COORDS);
			S.a0 = am[1]; S.a1 = top; S.a2 = bot; S.a3 = hitoff - hitlen + 1; S.a4 = hitlen; S.a5 = fg_frame_co(0);
			F_OP(FOP_LCOORDS, FPC_AM_AFTER_LC);
		}
		F_GOTO(FPC_AM_ADV);
	}
	case FPC_AM_AFTER_LC: {
		FAM_LOAD();
		FAM_SET(0, 16, 3, S.a0); FAM_SET(0, 13, 3, 0);
		FAM_STORE();
		S.nghits = 0; S.ghit_done = 0;
		F_GOTO(FPC_AM_RI_LOOP);
	}
	case FPC_AM_RI_LOOP: {
		FAM_LOAD();
		const uint32_t nco = FAM_GET(0, 16, 3), hitoff = am[2] & 0xffu, hitlen = (am[2] >> 8) & 0xffu;
		const uint32_t mp_i = FAM_GET(0, 0, 1), mp_j = FAM_GET(0, 1, 2);
		const uint32_t toff = W.ld(fg_res_base(mp_i, mp_j) + 1);
		for(uint32_t ri = 0; ri < nco; ri++) {
			uint32_t co3[3];
			W.ldv<3>(fg_frame_co(0) + 3 * ri, co3);
			if((uint64_t)co3[1] + (uint64_t)P.maxFragLen * 2 < toff || (uint64_t)toff + (uint64_t)P.maxFragLen * 2 < co3[1]) continue;   // (no_spliced_alignment :5683)
			if(S.nghits >= FG_NGH) F_BAIL(FB_NGHITS);
			fg_hit_init(W, fg_gbase(S.nghits), S.sv_fw != 0, hitoff - hitlen + 1, hitlen, co3[0], co3[1], co3[2]);
			S.nghits++;
		}
		am[2] = (am[2] & ~0xff0000u) | (hitlen << 16);
		FAM_STORE();
		F_GOTO(FPC_AM_ADV);
	}
	case FPC_AM_ADV: {
		FAM_LOAD();
		uint32_t hitoff = am[2] & 0xffu;
		const uint32_t hitlen = (am[2] >> 8) & 0xffu;
		if(hitlen > 0) hitoff -= (hitlen - 1);
		if(hitoff > 0) hitoff -= 1;
		am[2] = (am[2] & ~0xffu) | (hitoff & 0xffu);
		FAM_STORE();
		F_GOTO(FPC_AM_INNER);
	}
	case FPC_AM_EXT_LOOP: {
		FAM_LOAD();
		const uint32_t hi = FAM_GET(0, 11, 2);
		if(hi >= S.nghits) {
			FAM_SET(0, 7, 1, 1);
			FAM_SET(0, 1, 2, FAM_GET(0, 1, 2) + 1);
			FAM_STORE();
			F_GOTO(FPC_MP_LOOP);
		}
		S.a0 = 0; S.a1 = H2G_MAX; S.a2 = H2G_MAX; S.a3 = fg_gbase(hi);
		F_OP(FOP_EXTEND, FPC_AM_EXT_AFTER);
	}
	case FPC_AM_EXT_AFTER: {
		FAM_LOAD();
		const uint32_t hi = FAM_GET(0, 11, 2);
		const uint32_t gb = fg_gbase(hi);
		const uint32_t w4 = W.ld(gb + 4);
		fg_hit_copy(W, fg_frame_hit(0), gb);                    // RC_START(tmp2, rdoff, len, the mate's minimum score, alignMate = true)
		S.sp = 0; S.f_hitoff = w4 & 0xffu; S.f_hitlen = (w4 >> 8) & 0xffu;
		S.rc_minsc = fs_minsc(S, S.sv_rdi); S.rc_ret_pc = FPC_AM_REC_AFTER; S.ret = F_SMIN; S.rc_mate = 1;
		F_GOTO(FPC_RC_ENTRY);
	}
	case FPC_AM_REC_AFTER: {
		FAM_LOAD();
		S.rc_mate = 0;
		FAM_SET(0, 11, 2, FAM_GET(0, 11, 2) + 1);
		FAM_STORE();
		F_GOTO(FPC_AM_EXT_LOOP);
	}
#undef FAM_LOAD
#undef FAM_STORE
#undef FAM_GET
#undef FAM_SET
#endif
	// ======================================================================== getAnchorHits :5007-5193
	case FPC_GAH_LOOP: {
		const uint32_t x = (uint32_t)S.sel_r * 2 + (uint32_t)S.sel_f;
		const uint32_t offsetSize = fs_b4(S.rb_np, x);
		if(S.gh_hi >= offsetSize) F_GOTO(FPC_GAH_END);
		// candidates in index order: pool entries of this strand that were not resolved yet (bit 30 = has coordinates)
		uint32_t hj = FG_NLONGT, mj = 0, tj_top = 0, tj_bot = 0;
		const uint32_t npool = S.pool_x ? (uint32_t)FG_NLONGT : (uint32_t)FG_NLONG;
		int last = -1;                                          // index of the candidate looked at last
		for(uint32_t pass = 0; pass < npool; pass++) {
			// the unresolved entry of this strand with the smallest index above `last`
			uint32_t k = FG_NLONGT, m = 0;
			int kidx = 1 << 20;
#pragma unroll
			for(uint32_t q = 0; q < FG_NLONGT; q++) {
				if(q >= FG_NLONG && !((S.pool_x >> (q - FG_NLONG)) & 1u)) continue;
				const uint32_t mq = W.ld(fg_pool(q) + 2);
				const int iq = (int)((mq >> 24) & 31u);
				if(mq && !(mq & 0x40000000u) && ((mq >> 20) & 3u) == x && iq > last && iq < kidx) { k = q; m = mq; kidx = iq; }
			}
			if(k == FG_NLONGT) break;
			last = kidx;
			const uint32_t top = W.ld(fg_pool(k)), bot = W.ld(fg_pool(k) + 1);
			if(hj == FG_NLONGT) { hj = k; mj = m; tj_top = top; tj_bot = bot; continue; }
			const uint32_t tj = (mj >> 16) & 15u, tk = (m >> 16) & 15u, lj = (mj >> 8) & 0xffu, lk = (m >> 8) & 0xffu;
			const uint32_t sj = tj_bot - tj_top, sk = bot - top;
			const bool better = tj == tk ? (sj > sk || (sj == sk && lj < lk)) : (tk > tj);
			if(better) { hj = k; mj = m; tj_top = top; tj_bot = bot; }
		}
		if(hj == FG_NLONGT) F_GOTO(FPC_GAH_END);
		const uint32_t remained = maxsz - S.nghits;
		if(remained == 0) F_GOTO(FPC_GAH_END);
		const uint32_t len = (mj >> 8) & 0xffu, bwoff = mj & 0xffu;
#if FG_GRAPH
		uint32_t pn[3];                                     // the hit's node range and in-edge list: the elements are NODES
		W.ldv<3>(FW_LONGX + 3 * hj, pn);
		const uint32_t expected = pn[1] - pn[0];
#else
		const uint32_t expected = tj_bot - tj_top;
#endif
		if(expected > remained) F_BAIL(FB_SUBSAMPLE);       // the random sub-sample of a repeat's rows (:5096-5136)
		if(expected > FG_NCO) F_BAIL(FB_COORDS);
		S.gh_hj = hj; S.gh_nco = 0;
		S.gh_rdoff = fs_rl(S, S.sel_r) - bwoff - len;
		S.a0 = tj_top; S.a1 = tj_bot; S.a2 = expected; S.a3 = len; S.a4 = 0; S.a5 = fg_frame_co(0);
#if FG_GRAPH
		S.a6 = pn[0]; S.a7 = pn[1]; S.a8 = pn[2];
#endif
		F_OP(FOP_GCOORDS, FPC_GAH_FULL_AFTER);
	}
	case FPC_GAH_FULL_AFTER: {
		const uint32_t nco = S.a0;
		{ const uint32_t nt_ = S.nsteps + S.a1; if(nt_ > 0xffffu) F_BAIL(FB_OTHER); S.nsteps = nt_; }
		if(nco == 0) F_BAIL(FB_COORDS);                     // joinedToTextOff failed: the reference retries the same hit (:5140)
		const uint32_t mslot = fg_pool(S.gh_hj) + 2;
		const uint32_t m = W.ld(mslot);
		W.st(mslot, m | 0x40000000u);                       // ph.ncoords = nco
#if FG_GRAPH
		S.gh_nco = nco; S.gh_gsize = S.nghits; S.gh_k = 0;  // gsize + nco <= maxsz: no shuffle (:5147)
		F_GOTO(FPC_GAH_K_LOOP);
	}
	case FPC_GAH_K_LOOP: {                                // each coordinate goes through adjustWithALT, which may yield several hits or none (:5175)
		const uint32_t m = W.ld(fg_pool(S.gh_hj) + 2);
		const uint32_t len = (m >> 8) & 0xffu, type = (m >> 16) & 15u;
		const uint32_t rl = fs_rl(S, S.sel_r);
		while(S.gh_k < S.gh_nco) {
			uint32_t co3[3];
			W.ldv<3>(fg_frame_co(0) + 3 * S.gh_k, co3);
			const uint32_t tidx = co3[0], toff = co3[1];
			if(tidx == H2G_MAX) F_BAIL(FB_STRADDLE);
			bool overlapped = false;
			for(uint32_t l = 0; l < S.gh_gsize; l++) {
				const uint32_t gb = fg_gbase(l);
				const uint32_t w5 = W.ld(gb + 5);
				if(W.ld(gb) != tidx || ((w5 & 1u) != 0) != (S.sv_fw != 0)) continue;
				const uint32_t g_rdoff = W.ld(gb + 4) & 0xffu;
				const uint32_t hitoff = W.ld(gb + 1) + rl - g_rdoff, hitoff2 = toff + rl - S.gh_rdoff;
				if(hitoff == hitoff2) { overlapped = true; W.st(gb + 5, w5 + (1u << 8)); break; }   // _hitcount++
			}
			if(!overlapped) {
				// (the coordinate's third word is fetched HERE, not before the loop above: hipcc 7.2 at -O2 / -O3 loses a value that lives across
				// that divergent loop with a break — the joined offset arrived as the stale a4 whenever the loop had run; -O1 keeps it.  Measured on
				// the device with a per-trip trace of one read: profiles/r04_NOTES.md §2)
				S.a0 = S.gh_rdoff; S.a1 = len; S.a2 = tidx; S.a3 = toff; S.a4 = W.ld(fg_frame_co(0) + 3 * S.gh_k + 2);
				F_OP(FOP_ADJUST, FPC_GAH_K_AFTER);
			}
			if(type == H2G_CANDIDATE_HIT && S.nghits >= maxsz) break;
			S.gh_k = S.gh_k + 1;
		}
		if(type == H2G_CANDIDATE_HIT && S.nghits >= maxsz) F_GOTO(FPC_GAH_END);
		S.gh_hi++;
		F_GOTO(FPC_GAH_LOOP);
	}
	case FPC_GAH_K_AFTER: {
		const uint32_t type = (W.ld(fg_pool(S.gh_hj) + 2) >> 16) & 15u;
		if(type == H2G_CANDIDATE_HIT && S.nghits >= maxsz) F_GOTO(FPC_GAH_END);
		S.gh_k = S.gh_k + 1;
		F_GOTO(FPC_GAH_K_LOOP);
	}
#else
		const uint32_t len = (m >> 8) & 0xffu, type = (m >> 16) & 15u;
		const uint32_t gsize = S.nghits;                    // gsize + nco <= maxsz: no shuffle (:5147)
		const uint32_t rl = fs_rl(S, S.sel_r);
		for(uint32_t k = 0; k < nco; k++) {
			uint32_t co3[3];
			W.ldv<3>(fg_frame_co(0) + 3 * k, co3);
			const uint32_t tidx = co3[0], toff = co3[1], joff = co3[2];
			if(tidx == H2G_MAX) F_BAIL(FB_STRADDLE);
			bool overlapped = false;
			for(uint32_t l = 0; l < gsize; l++) {
				const uint32_t gb = fg_gbase(l);
				const uint32_t w5 = W.ld(gb + 5);
				if(W.ld(gb) != tidx || ((w5 & 1u) != 0) != (S.sv_fw != 0)) continue;
				const uint32_t g_rdoff = W.ld(gb + 4) & 0xffu;
				const uint32_t hitoff = W.ld(gb + 1) + rl - g_rdoff, hitoff2 = toff + rl - S.gh_rdoff;
				if(hitoff == hitoff2) { overlapped = true; W.st(gb + 5, w5 + (1u << 8)); break; }   // _hitcount++
			}
			if(!overlapped) {
				if(S.nghits >= FG_NGH) F_BAIL(FB_NGHITS);
				fg_hit_init(W, fg_gbase(S.nghits), S.sv_fw != 0, S.gh_rdoff, len, tidx, toff, joff);
				S.nghits++;
			}
			if(type == H2G_CANDIDATE_HIT && S.nghits >= maxsz) break;
		}
		if(type == H2G_CANDIDATE_HIT && S.nghits >= maxsz) F_GOTO(FPC_GAH_END);
		S.gh_hi++;
		F_GOTO(FPC_GAH_LOOP);
	}
#endif
	case FPC_GAH_END: {
		const uint32_t numHits = S.nghits;
		if(numHits == 0) { S.hs_found = 0; F_GOTO(FPC_AFTER_ALIGN); }
		const uint64_t add = (uint64_t)((-(int64_t)fs_minsc(S, S.sel_r)) / sc.mmpMax) * numHits;
		S.max_localindexatts = S.localindexatts + (uint32_t)(add > 10 ? add : 10);
		S.hs_hi = 0;
		F_GOTO(FPC_HS_EXT_LOOP);
	}
	// ======================================================================== hybridSearch spliced_aligner.h:112-322
	case FPC_HS_EXT_LOOP: {
		if(S.hs_hi >= S.nghits) { S.hs_hi = 0; F_GOTO(FPC_HS_LOOP); }
		S.a0 = 0; S.a1 = H2G_MAX; S.a2 = H2G_MAX; S.a3 = fg_gbase(S.hs_hi);
		F_OP(FOP_EXTEND, FPC_HS_EXT_AFTER);
	}
	case FPC_HS_EXT_AFTER: { S.ghit_done &= ~(1u << S.hs_hi); S.hs_hi++; F_GOTO(FPC_HS_EXT_LOOP); }
	case FPC_HS_LOOP: {
		if(S.hs_hi >= S.nghits) { S.hs_found = 1; F_GOTO(FPC_AFTER_ALIGN); }
		uint32_t hj = 0;
		for(; hj < S.nghits; hj++) if(!((S.ghit_done >> hj) & 1u)) break;
		if(hj >= S.nghits) { S.hs_found = 1; F_GOTO(FPC_AFTER_ALIGN); }
		for(uint32_t hk = hj + 1; hk < S.nghits; hk++) {
			if((S.ghit_done >> hk) & 1u) continue;
			const uint32_t bj = fg_gbase(hj), bk = fg_gbase(hk);
			const uint32_t ar = W.ld(bj + 5) >> 8, al = (W.ld(bj + 4) >> 8) & 0xffu, br = W.ld(bk + 5) >> 8, bl = (W.ld(bk + 4) >> 8) & 0xffu;
			if(br > ar || (br == ar && bl > al)) hj = hk;
		}
		S.hs_hj = hj;
		const uint32_t gb = fg_gbase(hj);
		// hybridSearch_recur(root) (RC_START): frame 0
		const uint32_t w4 = W.ld(gb + 4);
		fg_hit_copy(W, fg_frame_hit(0), gb);
		S.sp = 0; S.f_hitoff = w4 & 0xffu; S.f_hitlen = (w4 >> 8) & 0xffu;
		S.rc_minsc = fs_minsc(S, S.sel_r); S.rc_ret_pc = FPC_HS_AFTER_REC1; S.ret = F_SMIN;
		F_GOTO(FPC_RC_ENTRY);
	}
	case FPC_HS_AFTER_REC1: { S.ghit_done |= 1u << S.hs_hj; S.hs_hi++; F_GOTO(FPC_HS_LOOP); }     // (no SwAligner pass: bowtie2_dp == 0)
	// ======================================================================== hybridSearch_recur spliced_aligner.h:331-2052
	case FPC_RC_ENTRY: {
		const FHit hit = fh_load(W, fg_frame_hit(S.sp));
		const uint32_t hitoff = S.f_hitoff, hitlen = S.f_hitlen, rdlen = fs_rl(S, S.sv_rdi);
		const int32_t minsc = S.rc_minsc;
		S.f_maxsc = F_SMIN;
		if(hit.score + F_CUSHION() < minsc) F_RC_RET(S.f_maxsc);    // (the cushion: alignMate only)
		if(hitoff == hit.rdoff - hit.trim5 && hitlen == hit.len + hit.trim5 + hit.trim3) {
			const uint32_t hsh = fh_hash(hit);
			const uint32_t sb = FW_SRCH + S.sv_rdi * FG_NSRCH * (1 + FG_HW), ns = fs_nsearched(S, S.sv_rdi);
			bool searched = false;
			if(ns > 0) {
				uint32_t hs[FG_NSRCH];
#pragma unroll
				for(uint32_t i = 0; i < FG_NSRCH; i++) hs[i] = W.cold[sb - FW_HOT + i * (1 + FG_HW)];      // (every hash at once: one HBM latency)
#pragma unroll
				for(uint32_t i = 0; i < FG_NSRCH; i++) if(!searched && i < ns && hs[i] == hsh && fh_equal(fh_load(W, sb + i * (1 + FG_HW) + 1), hit)) searched = true;
			}
			if(searched) F_RC_RET(S.f_maxsc);                   // isSearched :6898
			if(ns >= FG_NSRCH) F_BAIL(FB_SEARCHED);
			W.st(sb + ns * (1 + FG_HW), hsh);
			if(!fh_store(W, sb + ns * (1 + FG_HW) + 1, hit)) F_BAIL(FB_EDITS);
			if(S.sv_rdi == 0) S.nsearched0 = ns + 1; else S.nsearched1 = ns + 1;
		}
		if(hitoff == 0 && hitlen == rdlen) {
			// redundant() :6311 over the summaries: same locus, strand and edit count => the general machine compares the edits
			const uint32_t nr = fs_nres(S, S.sv_rdi);
			for(uint32_t i = 0; i < nr; i++) {
				const uint32_t rb = fg_res_base(S.sv_rdi, i), m = W.ld(rb + 2);
				if(W.ld(rb) == hit.tidx && W.ld(rb + 1) == hit.toff && (m & 1u) == (hit.fw ? 1u : 0u) && ((m >> 1) & 7u) == hit.nedits) F_BAIL(FB_REDUNDANT);
			}
			// reportHit :6064 (al_report)
			if(!(hit.rdoff - hit.trim5 > 0 || hit.len + hit.trim5 + hit.trim3 < rdlen) && hit.score >= minsc) {
				if(nr >= FG_NRES) F_BAIL(FB_NRES);
				const uint32_t ext = fg_ref_extent(hit);
				if(hit.score < -32000 || hit.score > 32000 || ext > 0xfffu) F_BAIL(FB_OTHER);
				if(nr >= (S.paired ? C.O.pair_slots : C.O.aln_slots)) F_BAIL(FB_NRES);
				if(!fh_store(W, fg_res_hit(S.sv_rdi, nr), hit)) F_BAIL(FB_EDITS);      // (its record is written by FPC_FINISH)
				const uint32_t rb = fg_res_base(S.sv_rdi, nr);
				W.st(rb, hit.tidx); W.st(rb + 1, hit.toff);
				W.st(rb + 2, (hit.fw ? 1u : 0u) | (hit.nedits << 1) | (ext << 4) | ((uint32_t)(hit.score + 32768) << 16));
				const int32_t s = hit.score;
				if(S.sv_rdi == 0) {
					S.nres0 = nr + 1;
					if(S.bestUnp0 == F_SMIN16 || s > S.bestUnp0) { S.best2Unp0 = S.bestUnp0; S.bestUnp0 = s; } else if(S.best2Unp0 == F_SMIN16 || s > S.best2Unp0) S.best2Unp0 = s;
				} else {
					S.nres1 = nr + 1;
					if(S.bestUnp1 == F_SMIN16 || s > S.bestUnp1) { S.best2Unp1 = S.bestUnp1; S.bestUnp1 = s; } else if(S.best2Unp1 == F_SMIN16 || s > S.best2Unp1) S.best2Unp1 = s;
				}
			}
			// (the frame's maximum follows the hit whether or not it was reported: spliced_aligner.h:676)
			if(S.f_maxsc == F_SMIN || hit.score > S.f_maxsc) S.f_maxsc = hit.score;
			F_RC_RET(S.f_maxsc);
		}
		if(S.sp >= FG_NFRAME - 1) F_BAIL(FB_DEPTH);          // the deepest frame holds no lists: it may only report
		if(hitoff > 0 && (hitoff + hitlen == rdlen || hitoff + hitoff < rdlen - hitlen)) F_GOTO(FPC_RC_ENTRY_LX);
		F_GOTO(FPC_RC_ENTRY_RX);
	}
	case FPC_RC_ENTRY_LX: {                                // extend to the left (spliced_aligner.h:813-1360)
		const uint32_t hb = fg_frame_hit(S.sp);
		const uint32_t h_rdoff = W.ld(hb + 4) & 0xffu;
		S.f_uselocal = 1;
		if(S.f_hitoff == h_rdoff && S.f_hitoff <= minK) {
			fg_hit_copy(W, FW_T1, hb);
			S.a0 = 1; S.a1 = H2G_MAX; S.a2 = 0; S.a3 = FW_T1;
			F_OP(FOP_EXTEND, FPC_RC_ENTRY_L2);
		}
		F_GOTO(FPC_RC_ENTRY_L3);
	}
	case FPC_RC_ENTRY_RX: {                                // extend to the right (:1496-2050)
		const uint32_t hb = fg_frame_hit(S.sp);
		const uint32_t w4 = W.ld(hb + 4), h_len = (w4 >> 8) & 0xffu;
		const uint32_t rdlen = fs_rl(S, S.sv_rdi);
		S.f_uselocal = 1;
		if(h_len == S.f_hitlen && S.f_hitoff + S.f_hitlen + minK > rdlen) {
			fg_hit_copy(W, FW_T1, hb);
			S.a0 = 1; S.a1 = 0; S.a2 = H2G_MAX; S.a3 = FW_T1;
			F_OP(FOP_EXTEND, FPC_RC_ENTRY_R2);
		}
		F_GOTO(FPC_RC_ENTRY_R3);
	}
	case FPC_RC_ENTRY_L2: { if((W.ld(FW_T1 + 4) & 0xffu) == 0) S.f_uselocal = 0; F_GOTO(FPC_RC_ENTRY_L3); }
	case FPC_RC_ENTRY_R2: {
		const uint32_t w4 = W.ld(FW_T1 + 4);
		if((w4 & 0xffu) + ((w4 >> 8) & 0xffu) == fs_rl(S, S.sv_rdi)) S.f_uselocal = 0;
		F_GOTO(FPC_RC_ENTRY_R3);
	}
	case FPC_RC_ENTRY_L3:
	case FPC_RC_ENTRY_R3: {
		const uint32_t hb = fg_frame_hit(S.sp);
		S.f_lidx = local_index_of(*C.ls, W.ld(hb), W.ld(hb + 1));
		S.f_success = 0; S.f_first = 1; S.f_count = 0; S.f_prev = (int32_t)W.ld(hb + 3); S.f_ncoords = 0; S.f_ri = 0; S.f_nlocal = 0; S.f_ti = 0;
		if(S.pc == FPC_RC_ENTRY_L3) F_GOTO(FPC_L_WHILE);
		F_GOTO(FPC_R_WHILE);
	}
	// =============================== LEFT ===============================
	case FPC_L_WHILE: {
		if(S.f_success) F_GOTO(FPC_L_AFTER_WHILE);
		if(!(S.f_count++ < 2)) F_GOTO(FPC_L_AFTER_WHILE);
		if(!S.f_uselocal) F_GOTO(FPC_L_AFTER_WHILE);
		if(S.localindexatts >= S.max_localindexatts) F_GOTO(FPC_L_AFTER_WHILE);
		if(S.f_first) S.f_first = 0;
		else {
			S.f_lidx = S.f_lidx == H2G_MAX ? H2G_MAX : local_index_prev(*C.ls, S.f_lidx);
			if(S.f_lidx == H2G_MAX || C.ls->desc[S.f_lidx].len == 0) F_GOTO(FPC_L_AFTER_WHILE);
		}
		if(S.f_lidx == H2G_MAX) F_GOTO(FPC_L_AFTER_WHILE);
		uint32_t extoff = S.f_hitoff - 1;
		if(extoff > 0) extoff -= 1;
		if(extoff < P.minAnchorLen) extoff = P.minAnchorLen;
		S.f_extoff = extoff; S.f_extlen = 0; S.f_top = H2G_MAX; S.f_bot = H2G_MAX; S.f_nelt = H2G_MAX; S.f_noext = 0; S.f_unique = 0;
		F_GOTO(FPC_L_LS_LOOP);
	}
	case FPC_L_LS_LOOP: {
		if(!(S.f_extoff < fs_rl(S, S.sv_rdi))) F_GOTO(FPC_L_LS_DONE);
		S.f_extlen = 0; S.f_unique = 1;
		S.localindexatts++;
		if(C.ls->desc[S.f_lidx].len == 0) {
#if FG_GRAPH
			S.a6 = S.f_ntop; S.a7 = S.f_nbot; S.a8 = S.f_ie;
#endif
			S.a0 = 0; S.a1 = 0; S.a2 = S.f_top; S.a3 = S.f_bot; S.a4 = 1; F_GOTO(FPC_L_LS_AFTER);
		}
		S.a0 = S.f_lidx; S.a1 = S.f_extoff; S.a2 = 0xffffu; S.a3 = 1; S.a4 = S.f_top; S.a5 = S.f_bot;
		F_OP(FOP_LSEARCH, FPC_L_LS_AFTER);
	}
	case FPC_L_LS_AFTER: {
		S.f_nelt = S.a0; S.f_extlen = S.a1; S.f_top = S.a2; S.f_bot = S.a3; S.f_unique = S.a4 & 1u;
#if FG_GRAPH
		S.f_ntop = S.a6; S.f_nbot = S.a7; S.f_ie = S.a8;
#endif
		if(S.f_extoff + 1 - S.f_extlen >= S.f_hitoff) { S.f_noext = 1; F_GOTO(FPC_L_LS_DONE); }
		if(S.f_nelt <= 5) F_GOTO(FPC_L_LS_DONE);
		S.f_extoff++;
		F_GOTO(FPC_L_LS_LOOP);
	}
	case FPC_L_LS_DONE: {
		S.f_ncoords = 0; S.f_ri = -1;
		if(S.f_nelt > 0 && S.f_nelt <= 5 && S.f_extlen >= P.minAnchorLen && !S.f_noext) {
#if FG_GRAPH
			if(S.f_nbot - S.f_ntop > FG_NCO || S.f_bot - S.f_top > 255) F_BAIL(FB_COORDS);      // (the elements are nodes)
			if(S.f_ie == FG_IE_NOFIT) F_BAIL(FB_IEDGES);
			S.a6 = S.f_ntop; S.a7 = S.f_nbot; S.a8 = S.f_ie;
#else
			if(S.f_bot - S.f_top > FG_NCO) F_BAIL(FB_COORDS);
#endif
			S.a0 = S.f_lidx; S.a1 = S.f_top; S.a2 = S.f_bot; S.a3 = S.f_extoff + 1 - S.f_extlen; S.a4 = S.f_extlen; S.a5 = fg_frame_co(S.sp);
			F_OP(FOP_LCOORDS, FPC_L_LC_AFTER);
		}
		F_GOTO(FPC_L_FOR_RI);
	}
	case FPC_L_LC_AFTER:
	case FPC_R_LC_AFTER: {
		S.f_ncoords = S.a0;
		if(S.f_ncoords > 1) fg_sort_coords(W, fg_frame_co(S.sp), S.f_ncoords);
		if(S.pc == FPC_L_LC_AFTER) { S.f_ri = (int32_t)S.f_ncoords - 1; F_GOTO(FPC_L_FOR_RI); }
		F_GOTO(FPC_R_FOR_RI);
	}
	case FPC_L_FOR_RI: {
		if(S.f_ri < 0) F_GOTO(FPC_L_AFTER_FOR);
		const uint32_t cb = fg_frame_co(S.sp) + 3 * (uint32_t)S.f_ri, hb = fg_frame_hit(S.sp);
		const bool fw = (W.ld(hb + 5) & 1u) != 0;
		{ uint32_t co3[3]; W.ldv<3>(cb, co3); fg_hit_init(W, FW_T1, fw, S.f_extoff + 1 - S.f_extlen, S.f_extlen, co3[0], co3[1], co3[2]); }
#if FG_GRAPH
		S.a3 = FW_T1; F_OP(FOP_ADJMEMBER, FPC_L_RI_A);      // tempHit.adjustWithALT (spliced_aligner.h:946)
	}
	case FPC_L_RI_A: {
		if(!S.a0) { S.f_ri--; F_GOTO(FPC_L_FOR_RI); }
		const uint32_t hb = fg_frame_hit(S.sp);
#endif
		if(!fh_compatible(fh_load(W, FW_T1), fh_load(W, hb))) {
			if(S.f_count == 1) { S.f_ri--; F_GOTO(FPC_L_FOR_RI); }
			F_GOTO(FPC_L_AFTER_FOR);
		}
		if(S.f_unique) { S.a0 = 0; S.a1 = H2G_MAX; S.a2 = 0; S.a3 = FW_T1; F_OP(FOP_EXTEND, FPC_L_RI_B); }
		F_GOTO(FPC_L_RI_B);
	}
	case FPC_L_RI_B: { S.a3 = FW_T1; S.a4 = fg_frame_hit(S.sp); F_OP(FOP_COMBINE, FPC_L_RI_C); }
	case FPC_L_RI_C: {
		const bool combined = S.a0 != 0;
		const int32_t tscore = (int32_t)W.ld(FW_T1 + 3);
		int32_t m = S.rc_minsc;
		F_MINSC_LIVE(m);
		S.f_ri--;
		if(combined && tscore >= m) {
			if(tscore >= S.f_prev - sc.mmpMax) {
				const uint32_t w4 = W.ld(FW_T1 + 4);
				F_RC_CALL(FW_T1, w4 & 0xffu, ((w4 >> 8) & 0xffu) + (w4 >> 24), FPC_L_R1);
			}
			if(S.f_nlocal >= FG_NLOCAL) F_BAIL(FB_LOCALHITS);   // _local_genomeHits: kept for later (:1040)
			fg_hit_copy(W, fg_frame_lh(S.sp, S.f_nlocal), FW_T1); S.f_nlocal = S.f_nlocal + 1;
		}
		F_GOTO(FPC_L_FOR_RI);
	}
	case FPC_L_R1: { if(S.ret > S.f_maxsc) S.f_maxsc = S.ret; F_GOTO(FPC_L_FOR_RI); }
	case FPC_L_AFTER_FOR: {
		if(S.f_maxsc != F_SMIN && S.f_maxsc >= S.f_prev - sc.mmpMax) S.f_success = 1;
		S.f_ti = 0;
		if(!S.f_success && S.f_nlocal > 0 && (S.localindexatts >= S.max_localindexatts || S.f_count == 2 ||
		                                      (S.f_lidx == H2G_MAX || local_index_prev(*C.ls, S.f_lidx) == H2G_MAX)))
			F_GOTO(FPC_L_FOR_TI);
		F_GOTO(FPC_L_WHILE);
	}
	case FPC_L_FOR_TI: {
		if(S.f_ti >= S.f_nlocal) F_GOTO(FPC_L_WHILE);
		const uint32_t tb = fg_frame_lh(S.sp, S.f_ti);
		S.f_ti = S.f_ti + 1;
		int32_t m = S.rc_minsc;
		F_MINSC_LIVE(m);
		if((int32_t)W.ld(tb + 3) >= m) {
			const uint32_t w4 = W.ld(tb + 4);
			F_RC_CALL(tb, w4 & 0xffu, ((w4 >> 8) & 0xffu) + (w4 >> 24), FPC_L_R2);
		}
		F_GOTO(FPC_L_FOR_TI);
	}
	case FPC_L_R2: { if(S.ret > S.f_maxsc) S.f_maxsc = S.ret; F_GOTO(FPC_L_FOR_TI); }
	case FPC_L_AFTER_WHILE: {
		if(S.f_success) F_RC_RET(S.f_maxsc);
		S.f_ncoords = 0; S.f_ri = -1;
		if(S.f_hitoff > minK && S.localindexatts < S.max_localindexatts) {   // global search for long introns (:1085)
			S.f_extoff = S.f_hitoff - 1; S.f_extlen = 0;
			S.a1 = S.f_extoff; S.a3 = 1; S.a4 = H2G_MAX; S.a5 = H2G_MAX;
			F_OP(FOP_GSEARCH, FPC_L_GS_AFTER);
		}
		F_GOTO(FPC_L_FOR_G);
	}
	case FPC_L_GS_AFTER:
	case FPC_R_GS_AFTER: {
		const uint32_t nelt = S.a0, top = S.a2, bot = S.a3;
		const bool left = S.pc == FPC_L_GS_AFTER;
		S.f_extlen = S.a1; S.f_unique = S.a4 & 1u;
		if(nelt > 0 && nelt <= 5 && S.f_extlen >= minK) {
#if FG_GRAPH
			if(S.a8 == FG_IE_NOFIT) F_BAIL(FB_IEDGES);          // (a6 .. a8: the search's node range and in-edge list, as it left them)
#endif
			S.a0 = top; S.a1 = bot; S.a2 = bot - top; S.a3 = S.f_extlen; S.a4 = 1; S.a5 = fg_frame_co(S.sp);
			if(left) F_OP(FOP_GCOORDS, FPC_L_GC_AFTER); else F_OP(FOP_GCOORDS, FPC_R_GC_AFTER);
		}
		if(left) F_GOTO(FPC_L_FOR_G);
		F_GOTO(FPC_R_FOR_G);
	}
	case FPC_L_GC_AFTER:
	case FPC_R_GC_AFTER: {
		{ const uint32_t nt_ = S.nsteps + S.a1; if(nt_ > 0xffffu) F_BAIL(FB_OTHER); S.nsteps = nt_; }
		S.f_ncoords = S.a0;
		if(S.pc == FPC_L_GC_AFTER) {
			if(S.f_ncoords > 1) fg_sort_coords(W, fg_frame_co(S.sp), S.f_ncoords);
			S.f_ri = (int32_t)S.f_ncoords - 1;
			F_GOTO(FPC_L_FOR_G);
		}
		fg_sort_coords(W, fg_frame_co(S.sp), S.f_ncoords);
		F_GOTO(FPC_R_FOR_G);
	}
	case FPC_L_FOR_G: {
		if(S.f_ri < 0) F_GOTO(FPC_L_TRIM);
		const uint32_t cb = fg_frame_co(S.sp) + 3 * (uint32_t)S.f_ri, hb = fg_frame_hit(S.sp);
		S.f_ri--;
		const bool fw = (W.ld(hb + 5) & 1u) != 0;
		{ uint32_t co3[3]; W.ldv<3>(cb, co3); fg_hit_init(W, FW_T1, fw, S.f_extoff + 1 - S.f_extlen, S.f_extlen, co3[0], co3[1], co3[2]); }
#if FG_GRAPH
		S.a3 = FW_T1; F_OP(FOP_ADJMEMBER, FPC_L_G_A);       // (:1139)
	}
	case FPC_L_G_A: {
		if(!S.a0) F_GOTO(FPC_L_FOR_G);
		const uint32_t hb = fg_frame_hit(S.sp);
#endif
		if(!fh_compatible(fh_load(W, FW_T1), fh_load(W, hb))) F_GOTO(FPC_L_FOR_G);
		if(S.f_unique) { S.a0 = 0; S.a1 = H2G_MAX; S.a2 = 0; S.a3 = FW_T1; F_OP(FOP_EXTEND, FPC_L_G_B); }
		F_GOTO(FPC_L_G_B);
	}
	case FPC_L_G_B: { S.a3 = FW_T1; S.a4 = fg_frame_hit(S.sp); F_OP(FOP_COMBINE, FPC_L_G_C); }
	case FPC_L_G_C: {
		const bool combined = S.a0 != 0;
		const int32_t tscore = (int32_t)W.ld(FW_T1 + 3);
		int32_t m = S.rc_minsc;
		F_MINSC_LIVE(m);
		if(combined && tscore >= m) {
			const uint32_t w4 = W.ld(FW_T1 + 4);
			F_RC_CALL(FW_T1, w4 & 0xffu, ((w4 >> 8) & 0xffu) + (w4 >> 24), FPC_L_R3);
		}
		F_GOTO(FPC_L_FOR_G);
	}
	case FPC_L_R3: { if(S.ret > S.f_maxsc) S.f_maxsc = S.ret; F_GOTO(FPC_L_FOR_G); }
	case FPC_L_TRIM: {
		FHit hit = fh_load(W, fg_frame_hit(S.sp));
		const int64_t minsc = S.rc_minsc;
		const int64_t floor_ = (S.f_maxsc != F_SMIN && (int64_t)S.f_maxsc > minsc) ? (int64_t)S.f_maxsc : minsc;
		const int64_t tm = ((int64_t)hit.score - floor_) / sc_penalty(sc, 0);
		const uint32_t trimMax = (uint32_t)tm;
		if(hit.rdoff < trimMax) {
			hit.trim5 = hit.rdoff;                            // GenomeHit::trim5 hi_aligner.h:831
			fh_calc_score(sc, fg_sv(C, S), hit);
			if((S.f_maxsc == F_SMIN || hit.score > S.f_maxsc) && (int64_t)hit.score >= minsc) {
				if(!fh_store(W, FW_T1, hit)) F_BAIL(FB_EDITS);
				F_RC_CALL(FW_T1, 0, hit.len + hit.trim5 + hit.trim3, FPC_L_R4);
			}
		}
		F_GOTO(FPC_L_EXT);
	}
	case FPC_L_R4: { if(S.ret > S.f_maxsc) S.f_maxsc = S.ret; F_GOTO(FPC_L_EXT); }
	case FPC_L_EXT: {
		const uint32_t hb = fg_frame_hit(S.sp);
		fg_hit_copy(W, FW_T1, hb);
		const int32_t tscore = (int32_t)W.ld(hb + 3);
		const uint32_t t_rdoff = W.ld(hb + 4) & 0xffu;
		const uint32_t mm = (uint32_t)(((int64_t)tscore - S.rc_minsc) / sc.mmpMax);
		uint32_t nmm = 1;
		if(S.f_hitoff <= minK_local) nmm = t_rdoff < mm ? t_rdoff : mm;
		S.a0 = nmm; S.a1 = H2G_MAX; S.a2 = 0; S.a3 = FW_T1;
		F_OP(FOP_EXTEND, FPC_L_EXT_A);
	}
	case FPC_L_EXT_A: {
		const uint32_t hb = fg_frame_hit(S.sp);
		const uint32_t le = S.a0, hitoff = S.f_hitoff, hitlen = S.f_hitlen;
		const uint32_t h_rdoff = W.ld(hb + 4) & 0xffu;
		const int32_t hscore = (int32_t)W.ld(hb + 3), tscore = (int32_t)W.ld(FW_T1 + 3);
		int32_t m = S.rc_minsc;
		F_MINSC_LIVE(m);
		const uint32_t need = minK_local < h_rdoff ? minK_local : h_rdoff;
		if(tscore >= m && le >= need) {
			const uint32_t w4 = W.ld(FW_T1 + 4);
			F_RC_CALL(FW_T1, w4 & 0xffu, ((w4 >> 8) & 0xffu) + (w4 >> 24), FPC_L_R5);
		} else if(hitoff > minK_local) {
			const uint32_t jumplen = hitoff > minK ? minK : minK_local;
			const int64_t expected = (int64_t)hscore - (int64_t)((h_rdoff - hitoff) / jumplen) * sc.mmpMax - sc.mmpMax;
			if(expected >= (int64_t)m) F_RC_CALL(hb, hitoff - jumplen, hitlen + jumplen, FPC_L_R5);
		}
		F_RC_RET(S.f_maxsc);
	}
	case FPC_L_R5: { if(S.ret > S.f_maxsc) S.f_maxsc = S.ret; F_RC_RET(S.f_maxsc); }
	// =============================== RIGHT ===============================
	case FPC_R_WHILE: {
		const uint32_t rdlen = fs_rl(S, S.sv_rdi);
		if(S.f_success) F_GOTO(FPC_R_AFTER_WHILE);
		if(!(S.f_count++ < 2)) F_GOTO(FPC_R_AFTER_WHILE);
		if(!S.f_uselocal) F_GOTO(FPC_R_AFTER_WHILE);
		if(S.localindexatts >= S.max_localindexatts) F_GOTO(FPC_R_AFTER_WHILE);
		if(S.f_first) S.f_first = 0;
		else {
			S.f_lidx = S.f_lidx == H2G_MAX ? H2G_MAX : local_index_next(*C.ls, S.f_lidx);
			if(S.f_lidx == H2G_MAX || C.ls->desc[S.f_lidx].len == 0) F_GOTO(FPC_R_AFTER_WHILE);
		}
		if(S.f_lidx == H2G_MAX) F_GOTO(FPC_R_AFTER_WHILE);
		uint32_t extoff = S.f_hitoff + S.f_hitlen + minK_local;
		if(extoff + 1 < rdlen) extoff += 1;
		if(extoff >= rdlen) extoff = rdlen - 1;
		uint32_t maxHitLen = extoff - S.f_hitoff - S.f_hitlen;
		if(maxHitLen < minK_local) maxHitLen = minK_local;
		S.f_extoff = extoff; S.f_extlen = 0; S.f_top = H2G_MAX; S.f_bot = H2G_MAX; S.f_nelt = H2G_MAX; S.f_noext = 0; S.f_unique = 0; S.f_maxHitLen = maxHitLen;
		F_GOTO(FPC_R_LS_LOOP);
	}
	case FPC_R_LS_LOOP: {
		if(!(S.f_maxHitLen < S.f_extoff + 1 && S.f_extoff < fs_rl(S, S.sv_rdi))) F_GOTO(FPC_R_LS_DONE);
		S.f_extlen = 0; S.f_unique = 0;
		S.localindexatts++;
		if(C.ls->desc[S.f_lidx].len == 0) {
#if FG_GRAPH
			S.a6 = S.f_ntop; S.a7 = S.f_nbot; S.a8 = S.f_ie;
#endif
			S.a0 = 0; S.a1 = 0; S.a2 = S.f_top; S.a3 = S.f_bot; S.a4 = 0; F_GOTO(FPC_R_LS_AFTER);
		}
		S.a0 = S.f_lidx; S.a1 = S.f_extoff; S.a2 = S.f_maxHitLen; S.a3 = 0; S.a4 = S.f_top; S.a5 = S.f_bot;
		F_OP(FOP_LSEARCH, FPC_R_LS_AFTER);
	}
	case FPC_R_LS_AFTER: {
		const uint32_t rdlen = fs_rl(S, S.sv_rdi);
		S.f_nelt = S.a0; S.f_extlen = S.a1; S.f_top = S.a2; S.f_bot = S.a3; S.f_unique = S.a4 & 1u;
#if FG_GRAPH
		S.f_ntop = S.a6; S.f_nbot = S.a7; S.f_ie = S.a8;
#endif
		if(S.f_extoff < S.f_hitoff + S.f_hitlen) { S.f_noext = 1; F_GOTO(FPC_R_LS_DONE); }
		if(S.f_nelt <= 5) F_GOTO(FPC_R_LS_DONE);
		if(S.f_extoff + 1 < rdlen) S.f_extoff++;
		else { if(S.f_extlen < S.f_maxHitLen) F_GOTO(FPC_R_LS_DONE); else S.f_maxHitLen++; }
		F_GOTO(FPC_R_LS_LOOP);
	}
	case FPC_R_LS_DONE: {
		S.f_ncoords = 0; S.f_ri = 0;
		if(S.f_nelt > 0 && S.f_nelt <= 5 && S.f_extlen >= P.minAnchorLen && !S.f_noext) {
#if FG_GRAPH
			if(S.f_nbot - S.f_ntop > FG_NCO || S.f_bot - S.f_top > 255) F_BAIL(FB_COORDS);      // (the elements are nodes)
			if(S.f_ie == FG_IE_NOFIT) F_BAIL(FB_IEDGES);
			S.a6 = S.f_ntop; S.a7 = S.f_nbot; S.a8 = S.f_ie;
#else
			if(S.f_bot - S.f_top > FG_NCO) F_BAIL(FB_COORDS);
#endif
			S.a0 = S.f_lidx; S.a1 = S.f_top; S.a2 = S.f_bot; S.a3 = S.f_extoff + 1 - S.f_extlen; S.a4 = S.f_extlen; S.a5 = fg_frame_co(S.sp);
			F_OP(FOP_LCOORDS, FPC_R_LC_AFTER);
		}
		F_GOTO(FPC_R_FOR_RI);
	}
	case FPC_R_FOR_RI: {
		if(S.f_ri >= (int32_t)S.f_ncoords) F_GOTO(FPC_R_AFTER_FOR);
		const uint32_t cb = fg_frame_co(S.sp) + 3 * (uint32_t)S.f_ri, hb = fg_frame_hit(S.sp);
		const bool fw = (W.ld(hb + 5) & 1u) != 0;
		{ uint32_t co3[3]; W.ldv<3>(cb, co3); fg_hit_init(W, FW_T1, fw, S.f_extoff + 1 - S.f_extlen, S.f_extlen, co3[0], co3[1], co3[2]); }
#if FG_GRAPH
		S.a3 = FW_T1; F_OP(FOP_ADJMEMBER, FPC_R_RI_A);      // (:1635)
	}
	case FPC_R_RI_A: {
		if(!S.a0) { S.f_ri++; F_GOTO(FPC_R_FOR_RI); }
		const uint32_t hb = fg_frame_hit(S.sp);
#endif
		if(!fh_compatible(fh_load(W, hb), fh_load(W, FW_T1))) {
			if(S.f_count == 1) { S.f_ri++; F_GOTO(FPC_R_FOR_RI); }
			F_GOTO(FPC_R_AFTER_FOR);
		}
		S.a0 = 0; S.a1 = 0; S.a2 = H2G_MAX; S.a3 = FW_T1;
		F_OP(FOP_EXTEND, FPC_R_RI_B);
	}
	case FPC_R_RI_B: {
		const uint32_t t2 = fg_frame_hit(S.sp + 1);            // tmp2 lives where the callee's hit goes
		fg_hit_copy(W, t2, fg_frame_hit(S.sp));
		S.a3 = t2; S.a4 = FW_T1;
		F_OP(FOP_COMBINE, FPC_R_RI_C);
	}
	case FPC_R_RI_C: {
		const uint32_t t2 = fg_frame_hit(S.sp + 1);
		const bool combined = S.a0 != 0;
		const int32_t cscore = (int32_t)W.ld(t2 + 3);
		int32_t m = S.rc_minsc;
		F_MINSC_LIVE(m);
		S.f_ri++;
		if(combined && cscore >= m) {
			if(cscore >= S.f_prev - sc.mmpMax) {
				const uint32_t w4 = W.ld(t2 + 4);
				F_RC_CALL(t2, (w4 & 0xffu) - ((w4 >> 16) & 0xffu), ((w4 >> 8) & 0xffu) + ((w4 >> 16) & 0xffu), FPC_R_R1);
			}
			if(S.f_nlocal >= FG_NLOCAL) F_BAIL(FB_LOCALHITS);
			fg_hit_copy(W, fg_frame_lh(S.sp, S.f_nlocal), t2); S.f_nlocal = S.f_nlocal + 1;
		}
		F_GOTO(FPC_R_FOR_RI);
	}
	case FPC_R_R1: { if(S.ret > S.f_maxsc) S.f_maxsc = S.ret; F_GOTO(FPC_R_FOR_RI); }
	case FPC_R_AFTER_FOR: {
		if(S.f_maxsc != F_SMIN && S.f_maxsc >= S.f_prev - sc.mmpMax) S.f_success = 1;
		S.f_ti = 0;
		if(!S.f_success && S.f_nlocal > 0 && (S.localindexatts >= S.max_localindexatts || S.f_count == 2 ||
		                                      (S.f_lidx == H2G_MAX || local_index_next(*C.ls, S.f_lidx) == H2G_MAX)))
			F_GOTO(FPC_R_FOR_TI);
		F_GOTO(FPC_R_WHILE);
	}
	case FPC_R_FOR_TI: {
		if(S.f_ti >= S.f_nlocal) F_GOTO(FPC_R_WHILE);
		const uint32_t tb = fg_frame_lh(S.sp, S.f_ti);
		S.f_ti = S.f_ti + 1;
		int32_t m = S.rc_minsc;
		F_MINSC_LIVE(m);
		if((int32_t)W.ld(tb + 3) >= m) {
			const uint32_t w4 = W.ld(tb + 4);
			F_RC_CALL(tb, (w4 & 0xffu) - ((w4 >> 16) & 0xffu), ((w4 >> 8) & 0xffu) + ((w4 >> 16) & 0xffu), FPC_R_R2);
		}
		F_GOTO(FPC_R_FOR_TI);
	}
	case FPC_R_R2: { if(S.ret > S.f_maxsc) S.f_maxsc = S.ret; F_GOTO(FPC_R_FOR_TI); }
	case FPC_R_AFTER_WHILE: {
		if(S.f_success) F_RC_RET(S.f_maxsc);
		S.f_ncoords = 0; S.f_ri = 0;
		if(S.f_hitoff + S.f_hitlen + minK + 1 < fs_rl(S, S.sv_rdi) && S.localindexatts < S.max_localindexatts) {
			S.f_extoff = S.f_hitoff + S.f_hitlen + minK + 1; S.f_extlen = 0;
			S.a1 = S.f_extoff; S.a3 = 1; S.a4 = H2G_MAX; S.a5 = H2G_MAX;
			F_OP(FOP_GSEARCH, FPC_R_GS_AFTER);
		}
		F_GOTO(FPC_R_FOR_G);
	}
	case FPC_R_FOR_G: {
		if(S.f_ri >= (int32_t)S.f_ncoords) F_GOTO(FPC_R_TRIM);
		const uint32_t cb = fg_frame_co(S.sp) + 3 * (uint32_t)S.f_ri, hb = fg_frame_hit(S.sp);
		S.f_ri++;
		const bool fw = (W.ld(hb + 5) & 1u) != 0;
		{ uint32_t co3[3]; W.ldv<3>(cb, co3); fg_hit_init(W, FW_T1, fw, S.f_extoff + 1 - S.f_extlen, S.f_extlen, co3[0], co3[1], co3[2]); }
#if FG_GRAPH
		S.a3 = FW_T1; F_OP(FOP_ADJMEMBER, FPC_R_G_A);       // (:1826)
	}
	case FPC_R_G_A: {
		if(!S.a0) F_GOTO(FPC_R_FOR_G);
		const uint32_t hb = fg_frame_hit(S.sp);
#endif
		if(!fh_compatible(fh_load(W, hb), fh_load(W, FW_T1))) F_GOTO(FPC_R_FOR_G);
		S.a0 = 0; S.a1 = 0; S.a2 = H2G_MAX; S.a3 = FW_T1;
		F_OP(FOP_EXTEND, FPC_R_G_B);
	}
	case FPC_R_G_B: {
		const uint32_t t2 = fg_frame_hit(S.sp + 1);            // tmp2 lives where the callee's hit goes
		fg_hit_copy(W, t2, fg_frame_hit(S.sp));
		S.a3 = t2; S.a4 = FW_T1;
		F_OP(FOP_COMBINE, FPC_R_G_C);
	}
	case FPC_R_G_C: {
		const uint32_t t2 = fg_frame_hit(S.sp + 1);
		const bool combined = S.a0 != 0;
		const int32_t cscore = (int32_t)W.ld(t2 + 3);
		int32_t m = S.rc_minsc;
		F_MINSC_LIVE(m);
		if(combined && cscore >= m) {
			const uint32_t w4 = W.ld(t2 + 4);
			F_RC_CALL(t2, (w4 & 0xffu) - ((w4 >> 16) & 0xffu), ((w4 >> 8) & 0xffu) + ((w4 >> 16) & 0xffu), FPC_R_R3);
		}
		F_GOTO(FPC_R_FOR_G);
	}
	case FPC_R_R3: { if(S.ret > S.f_maxsc) S.f_maxsc = S.ret; F_GOTO(FPC_R_FOR_G); }
	case FPC_R_TRIM: {
		FHit hit = fh_load(W, fg_frame_hit(S.sp));
		const int64_t minsc = S.rc_minsc;
		const uint32_t trimLen = fs_rl(S, S.sv_rdi) - S.f_hitoff - hit.len - hit.trim5;
		const int64_t floor_ = (S.f_maxsc != F_SMIN && (int64_t)S.f_maxsc > minsc) ? (int64_t)S.f_maxsc : minsc;
		const uint32_t trimMax = (uint32_t)(((int64_t)hit.score - floor_) / sc_penalty(sc, 0));
		if(trimLen < trimMax) {
			hit.trim3 = trimLen;                              // GenomeHit::trim3 hi_aligner.h:855
			fh_calc_score(sc, fg_sv(C, S), hit);
			if((S.f_maxsc == F_SMIN || hit.score > S.f_maxsc) && (int64_t)hit.score >= minsc) {
				if(!fh_store(W, FW_T1, hit)) F_BAIL(FB_EDITS);
				F_RC_CALL(FW_T1, hit.rdoff - hit.trim5, hit.len + hit.trim5 + hit.trim3, FPC_R_R4);
			}
		}
		F_GOTO(FPC_R_EXT);
	}
	case FPC_R_R4: { if(S.ret > S.f_maxsc) S.f_maxsc = S.ret; F_GOTO(FPC_R_EXT); }
	case FPC_R_EXT: {
		const uint32_t hb = fg_frame_hit(S.sp);
		fg_hit_copy(W, FW_T1, hb);
		const uint32_t rdlen = fs_rl(S, S.sv_rdi);
		const int32_t tscore = (int32_t)W.ld(hb + 3);
		const uint32_t w4 = W.ld(hb + 4);
		const uint32_t mm = (uint32_t)(((int64_t)tscore - S.rc_minsc) / sc.mmpMax);
		uint32_t nmm = 1;
		if(rdlen - S.f_hitoff - S.f_hitlen <= minK_local) {
			const uint32_t rest = rdlen - (w4 & 0xffu) - ((w4 >> 8) & 0xffu);
			nmm = rest < mm ? rest : mm;
		}
		S.a0 = nmm; S.a1 = 0; S.a2 = H2G_MAX; S.a3 = FW_T1;
		F_OP(FOP_EXTEND, FPC_R_EXT_A);
	}
	case FPC_R_EXT_A: {
		const uint32_t hb = fg_frame_hit(S.sp);
		const uint32_t re = S.a1, hitoff = S.f_hitoff, hitlen = S.f_hitlen, rdlen = fs_rl(S, S.sv_rdi);
		const uint32_t hw4 = W.ld(hb + 4), h_rdoff = hw4 & 0xffu, h_len = (hw4 >> 8) & 0xffu;
		const int32_t hscore = (int32_t)W.ld(hb + 3), tscore = (int32_t)W.ld(FW_T1 + 3);
		int32_t m = S.rc_minsc;
		F_MINSC_LIVE(m);
		const uint32_t rest0 = rdlen - h_len - h_rdoff;
		const uint32_t need = minK_local < rest0 ? minK_local : rest0;
		if(tscore >= m && re >= need) {
			const uint32_t w4 = W.ld(FW_T1 + 4);
			F_RC_CALL(FW_T1, (w4 & 0xffu) - ((w4 >> 16) & 0xffu), ((w4 >> 8) & 0xffu) + ((w4 >> 16) & 0xffu), FPC_R_R5);
		} else if(hitoff + hitlen + minK_local < rdlen) {
			const uint32_t jumplen = hitoff + hitlen + minK < rdlen ? minK : minK_local;
			const int64_t expected = (int64_t)hscore - (int64_t)((hitlen - h_len) / jumplen) * sc.mmpMax - sc.mmpMax;
			if(expected >= (int64_t)m) F_RC_CALL(hb, hitoff, hitlen + jumplen, FPC_R_R5);
		}
		F_RC_RET(S.f_maxsc);
	}
	case FPC_R_R5: { if(S.ret > S.f_maxsc) S.f_maxsc = S.ret; F_RC_RET(S.f_maxsc); }
	// ======================================================================== finishRead's device half (mach_finish)
	case FPC_FINISH: {
		if(!S.paired) {
			ReadOut o;
			Rng rnd; rnd.last = S.rnd;
			const uint32_t sz = S.nres0;
			o.nres = sz; o.overflow = 0; o.nrank = S.nrank; o.nsteps = S.nsteps; o.depth = S.nframes_max; o.nside = S.nside;
			for(uint32_t k = 0; k < H2G_SELECT_CAP; k++) o.select[k] = 0;
			// selectByScore (al_select) over at most two reported hits
			int64_t key[FG_NRES] = {0, 0}; int64_t scv[FG_NRES] = {0, 0};
			h2g_alnres* recs = C.O.aln + (size_t)S.read * C.O.aln_slots;
			const FHit h0 = fh_load(W, fg_res_hit(0, 0)), h1 = fh_load(W, fg_res_hit(0, sz > 1 ? 1u : 0u));
			if(sz >= 1) { scv[0] = h0.score; key[0] = fg_hisat2_key(h0.score, h0.trim5 + h0.trim3); }
			if(sz >= 2) { scv[1] = h1.score; key[1] = fg_hisat2_key(h1.score, h1.trim5 + h1.trim3); }
			uint32_t nsel = 0; uint32_t ord[FG_NRES] = {0, 1};
			if(sz == 1) { nsel = 1; }
			else if(sz == 2) {
				// descending by (key, original offset)
				if(key[0] < key[1] || key[0] == key[1]) { ord[0] = 1; ord[1] = 0; }
				if(key[0] == key[1]) { const uint32_t r = rnd.nextU32() % 2; if(r > 0) { const uint32_t t = ord[0]; ord[0] = ord[1]; ord[1] = t; } }
				nsel = P.khits < 2 ? P.khits : 2;
				if(nsel == 2 && key[0] != key[1]) nsel = 1;
			}
			o.nselect = nsel;
			if(nsel >= 1) o.select[0] = (uint8_t)ord[0];
			if(nsel >= 2) o.select[1] = (uint8_t)ord[1];
			int64_t b = INT64_MIN, sb = INT64_MIN, bh = 0, sbh = 0;        // AlnSetSumm::init aligner_result.cpp:1209
			for(uint32_t k = 0; k < sz; k++) {
				const int64_t h = key[k], s = scv[k];
				if(b == INT64_MIN || s > b || (s == b && h > bh)) { sb = b; sbh = bh; b = s; bh = h; }
				else if(sb == INT64_MIN || s > sb || (s == sb && h > sbh)) { sb = s; sbh = h; }
			}
			o.best = b == INT64_MIN ? INT32_MIN : (int32_t)b; o.secbest = sb == INT64_MIN ? INT32_MIN : (int32_t)sb;
			o.best_h2 = (uint32_t)(uint64_t)bh; o.secbest_h2 = (uint32_t)(uint64_t)sbh;
			// output slot k holds res[select[k]]: written here, once, in that order (nothing is read back from the rows)
			if(sz >= 1) fg_write_rec(recs[0], ord[0] == 1 ? h1 : h0, S.rl0);
			if(sz >= 2) fg_write_rec(recs[1], ord[1] == 1 ? h1 : h0, S.rl0);
			C.O.rout[S.read] = o;
			S.rnd = rnd.last;
			S.a0 = nsel > 0;
		} else {
			PairOut o;
			o.nres[0] = S.nres0; o.nres[1] = S.nres1; o.npairs = S.npairs; o.overflow = 0;
			o.nrank = S.nrank; o.nsteps = S.nsteps; o.depth = S.nframes_max; o.nside = S.nside; o.rnd_state = S.rnd; o.pad = 0;
			for(uint32_t k = 0; k < AL_MAX_PAIRS; k++) {
				o.pair_i[k] = k < S.npairs ? (uint8_t)((S.pairs >> (4 * k)) & 3u) : 0;
				o.pair_j[k] = k < S.npairs ? (uint8_t)((S.pairs >> (4 * k + 2)) & 3u) : 0;
			}
			for(uint32_t m = 0; m < 2; m++) {                   // the records, in report order
				h2g_alnres* pl = (m ? C.O.paln[1] : C.O.paln[0]) + (size_t)S.read * C.O.pair_slots;
				const uint32_t nr = fs_nres(S, m), rl = fs_rl(S, m);
				for(uint32_t k = 0; k < nr; k++) fg_write_rec(pl[k], fh_load(W, fg_res_hit(m, k)), rl);
			}
			C.O.pout[S.read] = o;
			S.a0 = S.npairs > 0;
		}
		S.pc = FPC_DONE; S.op = FOP_NONE;
		return;
	}
	case FPC_DONE:
	case FPC_BAIL:
	default: return;
	}
}
#undef F_GOTO
#undef F_OP
#undef F_RC_RET
#undef F_RC_CALL
#undef F_MINSC_LIVE

// ---------------------------------------------------------------------------------------- the primitives (one code site each)
H2G_HD void fast_op_psearch(const FCtx& C, FState& S) {
	const AlnParams& P = *C.P;
	h2g_fm_hit fh;
#if FG_GRAPH
	IEdges ie;
	partial_search_graph_item(*C.g, fg_sv(C, S), S.a0, P.pseudogeneStop != 0, P.anchorStop != 0, P.khits, P.kseeds, &fh, &ie);
	S.a6 = fh.node_top; S.a7 = fh.node_bot; S.a8 = fg_ie_pack(ie);
#else
	partial_search_item(*C.g, fg_sv(C, S), S.a0, P.pseudogeneStop != 0, P.anchorStop != 0, P.khits, &fh);
#endif
	S.a5 = S.a0;
	S.a0 = fh.top; S.a1 = fh.bot;
	S.a2 = (fh.len & 0xffu) | (fh.hit_type << 8) | ((fh.done ? 1u : 0u) << 16) | ((fh.anchorStop ? 1u : 0u) << 17) | (fh.numUniqueSearch << 18);
	S.a3 = fh.cur; S.a4 = (fh.nrank & 0xffffu) | (fh.nside << 16);
}
// SA walks are chunked: in lock-step a wave waits for its longest walk (the walk length is geometric: mean 15, the longest of 60 about 65),
// so a primitive gives up after FG_WALK_STEPS LF steps, keeps (element, row, steps so far) in its arguments and asks to be queued again.
#ifndef FG_WALK_STEPS
#define FG_WALK_STEPS 20
#endif
#ifndef FG_LWALK_STEPS
#define FG_LWALK_STEPS 10
#endif
// getGenomeCoords :5774 (genome_coords_item), coordinates straight to the word store.  a0 top a1 bot a2 maxelt a3 len a4 rejectStraddle a5 dst;
// while it is under way a1 = the row the walk stands at, a2 = elements | (1 | element << 1 | coordinates written << 4 | jumps << 7) << 8.
// true: not finished
#if FG_GRAPH
// getGenomeCoords on a graph index.  a0 top a1 bot a2 maxelt a3 len a4 rejectStraddle a5 dst a6 / a7 node range a8 in-edges.
// The common anchor is ONE node reached through ONE row: its walk is a row walk (gw_walk_single), kept in registers and chunked like the
// linear one — under way: a4 bit 1 set, a0 = the row, a6 = the node, a2 = steps so far.  Everything else (several nodes, a node with
// extra in-edges) is the node-based group walk (genome_coords_graph_item) on this lane's scratch, in one go.
H2G_HD bool fast_op_gcoords(const FCtx& C, FState& S, const FWords& W) {
	if(!(S.a4 & 2u)) {
		uint32_t nelt = S.a7 - S.a6;
		if(nelt > S.a2) nelt = S.a2;
		if(nelt == 1 && S.a1 - S.a0 == 1) { S.a4 |= 2u; S.a2 = 0; }
	}
	if(S.a4 & 2u) {
		uint32_t row = S.a0, node = S.a6, steps = S.a2, off = 0;
		const uint32_t before = steps;
		const bool found = gw_walk_single(*C.g, &row, &node, &steps, FG_WALK_STEPS, &off);
		S.nsteps += steps - before;
		if(!found) {
			if(steps > 60000u) { S.pc = FPC_BAIL; S.bail = FB_GWALK; return false; }
			S.a0 = row; S.a6 = node; S.a2 = steps;
			return true;
		}
		uint32_t tidx = 0, toff = 0, n = 0;
		bool st2 = false;
		joined_to_text(*C.g, S.a3, off, &tidx, &toff, (S.a4 & 1u) != 0, &st2);
		if(tidx != H2G_MAX) { const uint32_t co3[3] = {st2 ? H2G_MAX : tidx, toff, off}; W.stv<3>(S.a5, co3); n = 1; }
		S.a0 = n; S.a1 = 0;
		return false;
	}
	if(C.defer_slow && !(S.a4 & 4u)) { S.a4 |= 4u; return true; }                          // the group walk: on the queue of slow walks
	S.a4 &= ~4u;
	IEdges ie;
	fg_ie_unpack(S.a8, &ie);
	h2g_coord co[FG_NCO];
	h2g_sa_result res;
	genome_coords_graph_item(*C.g, &C.gws->gw, S.a0, S.a1, S.a6, S.a7, &ie, S.a2, S.a3, (S.a4 & 1u) != 0, co, FG_NCO, &res);
	if(res.nsteps == H2G_MAX) { S.pc = FPC_BAIL; S.bail = FB_GWALK; return false; }       // a capacity of the group walk
#pragma unroll
	for(uint32_t e = 0; e < FG_NCO; e++) if(e < res.ncoords) { const uint32_t co3[3] = {co[e].tidx, co[e].toff, co[e].joinedOff}; W.stv<3>(S.a5 + 3 * e, co3); }
	S.nsteps += res.nsteps;
	S.a0 = res.ncoords; S.a1 = 0;
	return false;
}
#else
H2G_HD bool fast_op_gcoords(const FCtx& C, FState& S, const FWords& W) {
	const DGfm& g = *C.g;
	const bool reject = (S.a4 & 1u) != 0;
	const uint32_t prog = S.a2 >> 8;
	uint32_t nelt, e = 0, n = 0, jumps = 0, row = 0;
	bool resume = false;
	if(prog & 1u) { nelt = S.a2 & 0xffu; e = (prog >> 1) & 7u; n = (prog >> 4) & 7u; jumps = prog >> 7; row = S.a1; resume = true; }
	else {
		nelt = S.a1 - S.a0;
		if(nelt > (S.a2 & 0xffu)) nelt = S.a2 & 0xffu;
		if(nelt > FG_NCO) nelt = FG_NCO;
	}
	uint32_t budget = FG_WALK_STEPS, nsteps = 0;
	for(; e < nelt; e++) {
		if(!resume) { row = S.a0 + e; jumps = 0; }
		resume = false;
		// sa_walk (h2g_core.h) with a step budget
		uint32_t joff = 0;
		bool found = false;
		while(true) {
			if(g.nZ && row == g.zoff) { joff = jumps; found = true; break; }
			if((row & g.offMask) == row) {
				const uint32_t off = g.offs[row >> g.offRate];
				if(off != H2G_MAX) { joff = off + jumps; found = true; break; }
			}
			if(budget == 0) break;
			const uint32_t s0 = row / 192u, c0 = row - s0 * 192u;
			const Side64 sd = load_side64(g.sides + (size_t)s0 * 64);
			const int c = rowL_in_side64(sd, c0);
			row = rank_in_side64(g, sd, s0, c0, c);
			jumps++; budget--; nsteps++;
		}
		if(!found) {                                            // out of budget: come back
			if(jumps >= (1u << 16)) { S.pc = FPC_BAIL; S.bail = FB_OTHER; return false; }
			S.a1 = row;
			S.a2 = nelt | ((1u | (e << 1) | (n << 4) | (jumps << 7)) << 8);
			S.nsteps += nsteps;
			return true;
		}
		uint32_t tidx = 0, toff = 0;
		bool st2 = false;
		joined_to_text(g, S.a3, joff, &tidx, &toff, reject, &st2);
		if(tidx == H2G_MAX) break;
		{ const uint32_t co3[3] = {st2 ? H2G_MAX : tidx, toff, joff}; W.stv<3>(S.a5 + 3 * e, co3); }
		n = e + 1;
	}
	S.nsteps += nsteps;
	S.a0 = n; S.a1 = 0;
	return false;
}
#endif
#if FG_GRAPH
// GenomeHit::extend on a graph index: alignWithALTs through the ALT database (extend_item_alts) over the unpacked hit
H2G_HD bool fast_op_extend(const FCtx& C, FState& S, const FWords& W) {
	const bool deferred = (S.a3 >> 31) != 0;         // (asked again from the queue of slow extensions: the tests below said so already)
	S.a3 &= 0x7fffffffu;
	FHit h = fh_load(W, S.a3);
	if(!deferred) {
		// No ALT within reach of either extension (extend_item_alts' own tests; the right end of a hit does not move when it grows to the left):
		// alignWithALTs degenerates to the mismatch scan of a linear index — the word-wise register code, no unpacked record
		const DAlts& A = *C.alts;
		const SeqView sv = fg_sv(C, S);
		bool plain = true;
		if(S.a1 > 0 && h.rdoff > 0) {
			const uint32_t wlo = h.joff > h.rdoff + 16 ? h.joff - h.rdoff - 16 : 0;
			if(alt_lobound(A, wlo) != alt_lobound(A, h.joff + 2)) plain = false;
		}
		if(plain && S.a2 > 0 && h.rdoff + h.len < sv.len) {
			int ref_ext = (int)h.len;
#pragma unroll
			for(uint32_t k = 0; k < FG_FE; k++) if(k < h.nedits) {
				const uint32_t e = FE_GET(h, k), t = FE_TYPE(e);
				if(t == H2G_EDIT_REF_GAP) ref_ext--; else if(t == H2G_EDIT_READ_GAP) ref_ext++; else if(t == H2G_EDIT_MM && FE_CCODE(e) == 4) ref_ext--;
			}
			const uint32_t jr = h.joff + (uint32_t)ref_ext, rr = sv.len - (h.rdoff + h.len);
			// (a left extension of up to rdoff bases can add mismatches against reference Ns, which move jr by nothing; the window is taken 16 wider on both sides all the same)
			if(alt_lobound(A, jr > 18 ? jr - 18 : 0) != alt_lobound(A, jr + rr + 32)) plain = false;
		}
		if(plain) {
			uint32_t le = H2G_MAX, re = H2G_MAX;
			fh_extend(*C.ref, C.P->sc, sv, h, S.a0, S.a1, S.a2, &le, &re);
			if(!fh_store(W, S.a3, h)) { S.pc = FPC_BAIL; S.bail = FB_EDITS; }
			S.a0 = le; S.a1 = re;
			return false;
		}
		if(C.defer_slow) { S.a3 |= 0x80000000u; return true; }
	}
	h2g_ghit g;
	fh_to_ghit(h, &g);
	uint32_t le = H2G_MAX, re = H2G_MAX;
	extend_item_alts(*C.ref, *C.alts, C.P->sc, fg_sv(C, S), &g, S.a0, S.a1, S.a2, &le, &re, &C.gws->awa);
	ghit_to_fh(g, h);
	if(!fh_store(W, S.a3, h)) { S.pc = FPC_BAIL; S.bail = FB_EDITS; }
	S.a0 = le; S.a1 = re;
	return false;
}
// static adjustWithALT (hi_aligner.h:2239; adjust_with_alt): the anchor (a0 rdoff, a1 len) placed at (a2 tidx, a3 toff, a4 joinedOff) becomes the
// genome hits it yields.  It compares what it adds with the hits already there, so those go in first.
H2G_HD void fast_op_adjust(const FCtx& C, FState& S, const FWords& W) {
	h2g_ghit arr[FG_NGH];
	uint32_t n = S.nghits, ovf = 0;
	for(uint32_t k = 0; k < n && k < FG_NGH; k++) fh_to_ghit(fh_load(W, fg_gbase(k)), &arr[k]);
	adjust_with_alt(*C.g, *C.ref, *C.alts, fg_sv(C, S), S.a0, S.a1, S.a2, S.a3, S.a4, arr, &n, FG_NGH, &C.gws->awa, &ovf);
	if(ovf) { S.pc = FPC_BAIL; S.bail = FB_NGHITS; return; }
	for(uint32_t k = S.nghits; k < n && k < FG_NGH; k++) {
		FHit h;
		ghit_to_fh(arr[k], h);
		if(!fh_store(W, fg_gbase(k), h)) { S.pc = FPC_BAIL; S.bail = FB_EDITS; return; }
	}
	S.nghits = n;
	S.a0 = 0;
}
// member adjustWithALT (hi_aligner.h:2395; adjust_with_alt_member) of the hit at a3: re-seated, or a0 = 0
H2G_HD void fast_op_adjmember(const FCtx& C, FState& S, const FWords& W) {
	FHit h = fh_load(W, S.a3);
	h2g_ghit g;
	fh_to_ghit(h, &g);
	uint32_t ovf = 0;
	const bool ok = adjust_with_alt_member(*C.g, *C.ref, *C.alts, fg_sv(C, S), &g, &C.gws->awa, &ovf);
	if(ovf) { S.pc = FPC_BAIL; S.bail = FB_OTHER; return; }
	ghit_to_fh(g, h);
	if(ok && !fh_store(W, S.a3, h)) { S.pc = FPC_BAIL; S.bail = FB_EDITS; return; }
	S.a0 = ok ? 1u : 0u;
}
// localGFMSearch on the three kinds of local index a graph index holds (al_local_search)
H2G_HD void fast_op_lsearch(const FCtx& C, FState& S) {
	const AlnParams& P = *C.P;
	uint32_t extlen = 0, top = S.a4, bot = S.a5, nr[2] = {0, 0};
	bool uniqueStop = S.a3 != 0;
	LIdx lx; lx.ls = C.ls; lx.d = &C.ls->desc[S.a0];
	uint32_t nelt;
	if(local_is_linear(*lx.d)) {
		LIdxW lw; lw.ls = C.ls; lw.d = lx.d;
		nelt = gfm_search(lw, fg_sv(C, S), S.a1, &extlen, &top, &bot, &uniqueStop, P.minK_local, S.a2, P.kseeds, true, nr);
		S.a6 = top; S.a7 = bot; S.a8 = 0;
	} else {
		const LGfm x = lgfm_of(*C.ls, *lx.d);
		GRange r;
		r.top = top; r.bot = bot; r.node_top = r.node_bot = 0;
		IEdges ie;
		nelt = gfm_search_graph(x, lx, fg_sv(C, S), S.a1, &extlen, &r, &ie, &uniqueStop, P.minK_local, S.a2, P.kseeds, true, P.kseeds, nr);
		top = r.top; bot = r.bot;
		S.a6 = r.node_top; S.a7 = r.node_bot; S.a8 = fg_ie_pack(ie);
	}
	S.nrank += nr[0]; S.nside += nr[1];
	S.a0 = nelt; S.a1 = extlen; S.a2 = top; S.a3 = bot; S.a4 = uniqueStop ? 1u : 0u;
}
#else
H2G_HD void fast_op_extend(const FCtx& C, FState& S, const FWords& W) {
	FHit h = fh_load(W, S.a3);
	uint32_t le = H2G_MAX, re = H2G_MAX;
	fh_extend(*C.ref, C.P->sc, fg_sv(C, S), h, S.a0, S.a1, S.a2, &le, &re);
	if(!fh_store(W, S.a3, h)) { S.pc = FPC_BAIL; S.bail = FB_EDITS; }
	S.a0 = le; S.a1 = re;
}
H2G_HD void fast_op_lsearch(const FCtx& C, FState& S) {
	const AlnParams& P = *C.P;
	uint32_t extlen = 0, top = S.a4, bot = S.a5, nr[2] = {0, 0};
	bool uniqueStop = S.a3 != 0;
	LIdxR lx; lx.init(C.ls, &C.ls->desc[S.a0]);
	const uint32_t nelt = gfm_search(lx, fg_sv(C, S), S.a1, &extlen, &top, &bot, &uniqueStop, P.minK_local, S.a2, P.kseeds, true, nr);
	S.nrank += nr[0]; S.nside += nr[1];
	S.a0 = nelt; S.a1 = extlen; S.a2 = top; S.a3 = bot; S.a4 = uniqueStop ? 1u : 0u;
}
#endif
// getGenomeCoords_local :5861 (genome_coords_local), chunked like fast_op_gcoords.  a0 lidx a1 top a2 bot a3 rdoff a4 rdlen a5 dst; under way:
// a2 bits 16.. = 1 | element << 1 | coordinates written << 4 | jumps << 7, a4 bits 8.. = the row (16 bits)
template <typename LX>
H2G_HD bool fast_op_lcoords_walk(const FCtx& C, FState& S, const FWords& W, const LX& lx) {
	const uint32_t offMask = (0xffffu << C.ls->offRate) & 0xffffu, offRate = C.ls->offRate;
	const uint16_t* offs = C.ls->words + lx.d->offs_off;
	const uint32_t top = S.a1, bot = S.a2 & 0xffffu, rdlen = S.a4 & 0xffu;
	uint32_t prog = S.a2 >> 16, e = 0, n = 0, jumps = 0, row = 0;
	bool resume = false;
	if(prog & 1u) { e = (prog >> 1) & 7u; n = (prog >> 4) & 7u; jumps = prog >> 7; row = S.a4 >> 8; resume = true; }
	uint32_t budget = FG_LWALK_STEPS, steps = 0;
	for(; e < bot - top; e++) {
		if(!resume) { row = top + e; jumps = 0; }
		resume = false;
		uint32_t joff = 0;
		bool found = false;
		while(true) {                                            // sa_walk_idx (h2g_align.h) with a step budget
			if(lx.is_zoff(row)) { joff = jumps; found = true; break; }
			if((row & offMask) == row) {
				const uint32_t off = offs[row >> offRate];
				if(off != 0xffffu) { joff = off + jumps; found = true; break; }
			}
			if(budget == 0) break;
			row = lx.lf(row);
			jumps++; budget--; steps++;
			if(jumps > 500) { S.pc = FPC_BAIL; S.bail = FB_OTHER; return false; }
		}
		if(!found) {
			S.a2 = bot | ((1u | (e << 1) | (n << 4) | (jumps << 7)) << 16);
			S.a4 = rdlen | (row << 8);
			S.nsteps += steps;
			return true;
		}
		h2g_coord c;
		if(!local_joff_to_coord(*C.ls, lx.d, joff, S.a3, rdlen, &c)) continue;
		if(n < FG_NCO) { const uint32_t co3[3] = {c.tidx, c.toff, c.joinedOff}; W.stv<3>(S.a5 + 3 * n, co3); n++; }
	}
	S.nsteps += steps;
	S.a0 = n;
	return false;
}
#if FG_GRAPH
// getGenomeCoords_local on whichever kind of local index this is (al_local_coords): a linear one inside the graph index walks as on a
// linear index (chunked), a graph one through the group walk.  a6 / a7 node range, a8 in-edges of the search that found the rows
H2G_HD bool fast_op_lcoords(const FCtx& C, FState& S, const FWords& W) {
	const DLocalDesc* d = &C.ls->desc[S.a0];
	if(local_is_linear(*d)) { LIdxW lw; lw.ls = C.ls; lw.d = d; return fast_op_lcoords_walk(C, S, W, lw); }
	const LGfm x = lgfm_of(*C.ls, *d);
	const uint32_t rdlen = S.a4 & 0xffu;
	// one node through one row: the row walk, chunked (under way: a2 bit 31 set, a1 = the row, a6 = the node, a7 = steps so far)
	if(!(S.a2 & 0x80000000u) && S.a7 - S.a6 == 1 && (S.a2 & 0xffffu) - S.a1 == 1) { S.a2 |= 0x80000000u; S.a7 = 0; }
	if(S.a2 & 0x80000000u) {
		uint32_t row = S.a1, node = S.a6, steps = S.a7, off = 0;
		const uint32_t before = steps;
		const bool found = gw_walk_single(x, &row, &node, &steps, FG_LWALK_STEPS, &off);
		S.nsteps += steps - before;
		if(!found) {
			if(steps > 60000u) { S.pc = FPC_BAIL; S.bail = FB_GWALK; return false; }
			S.a1 = row; S.a6 = node; S.a7 = steps;
			return true;
		}
		h2g_coord c;
		uint32_t n1 = 0;
		if(local_joff_to_coord(*C.ls, d, off & 0xffffu, S.a3, rdlen, &c)) { const uint32_t co3[3] = {c.tidx, c.toff, c.joinedOff}; W.stv<3>(S.a5, co3); n1 = 1; }
		S.a0 = n1;
		return false;
	}
	if(C.defer_slow && !(S.a2 & 0x40000000u)) { S.a2 |= 0x40000000u; return true; }      // the group walk: on the queue of slow walks
	S.a2 &= ~0x40000000u;
	const uint32_t top = S.a1, bot = S.a2 & 0xffffu;
	IEdges ie;
	fg_ie_unpack(S.a8, &ie);
	uint32_t nelt = 0, n = 0;
	if(!gw_resolve(x, &C.gws->gw, top, bot, S.a6, S.a7, &ie, bot - top, &nelt) || nelt > FG_NCO) { S.pc = FPC_BAIL; S.bail = FB_GWALK; return false; }
	S.nsteps += C.gws->gw.nsteps;
	for(uint32_t e = 0; e < nelt; e++) {
		h2g_coord c;
		if(!local_joff_to_coord(*C.ls, d, C.gws->gw.offs[e] & 0xffffu, S.a3, rdlen, &c)) continue;
		{ const uint32_t co3[3] = {c.tidx, c.toff, c.joinedOff}; W.stv<3>(S.a5 + 3 * n, co3); n++; }
	}
	S.a0 = n;
	return false;
}
// combineWith through hit_combine of h2g_align.h (the rescan of the joint looks mismatches up in the ALT database; insertions and deletions
// between the two hits are placed there as well) over the unpacked hits
H2G_HD void fast_op_combine(const FCtx& C, FState& S, const FWords& W) {
	FHit a = fh_load(W, S.a3);
	h2g_ghit ga, gb;
	fh_to_ghit(a, &ga);
	fh_to_ghit(fh_load(W, S.a4), &gb);
	const AlnParams& P = *C.P;
	const bool ok = hit_combine(*C.ref, P.sc, fg_sv(C, S), &ga, &gb, (int64_t)S.rc_minsc, P.minIntronLen, true,
	                            ScVec{C.sc, C.sc_stride}, ScVec{C.sc + (size_t)H2G_COMBINE_MAXLEN * C.sc_stride, C.sc_stride}, C.alts, nullptr, 0u, 0u);
	ghit_to_fh(ga, a);
	if(!fh_store(W, S.a3, a)) { S.pc = FPC_BAIL; S.bail = FB_EDITS; }
	S.a0 = ok ? 1u : 0u;
}
H2G_HD void fast_op_gsearch(const FCtx& C, FState& S) {       // globalGFMSearch :6606 (al_global_search)
	const AlnParams& P = *C.P;
	uint32_t extlen = 0, nr[2] = {0, 0};
	bool uniqueStop = S.a3 != 0;
	GIdx gx; gx.g = C.g;
	GRange r;
	r.top = S.a4; r.bot = S.a5; r.node_top = r.node_bot = 0;
	IEdges ie;
	const uint32_t nelt = gfm_search_graph(*C.g, gx, fg_sv(C, S), S.a1, &extlen, &r, &ie, &uniqueStop, C.g->minK, H2G_MAX, P.kseeds, false, P.kseeds, nr);
	S.nrank += nr[0]; S.nside += nr[1];
	S.a0 = nelt; S.a1 = extlen; S.a4 = uniqueStop ? 1u : 0u;
	if(nelt > 0) { S.a2 = r.top; S.a3 = r.bot; } else { S.a2 = S.a3 = H2G_MAX; }      // (top / bot stay what the caller passed: none)
	S.a6 = r.node_top; S.a7 = r.node_bot; S.a8 = fg_ie_pack(ie);
}
#else
// (LIdxR needs the descriptor pointer for local_joff_to_coord and the offs array: fast_op_lcoords_walk reads lx.d)
struct LIdxRD : LIdxR { const DLocalDesc* d; };
H2G_HD bool fast_op_lcoords(const FCtx& C, FState& S, const FWords& W) {
	LIdxRD lx; lx.d = &C.ls->desc[S.a0]; lx.init(C.ls, lx.d);
	return fast_op_lcoords_walk(C, S, W, lx);
}
H2G_HD void fast_op_combine(const FCtx& C, FState& S, const FWords& W) {
	FHit a = fh_load(W, S.a3);
	const FHit b = fh_load(W, S.a4);
	bool indel = false;
	const bool ok = fh_combine(*C.ref, C.P->sc, fg_sv(C, S), a, b, (int64_t)S.rc_minsc, &indel);
	if(indel) { S.pc = FPC_BAIL; S.bail = FB_INDEL; }
	else if(!fh_store(W, S.a3, a)) { S.pc = FPC_BAIL; S.bail = FB_EDITS; }
	S.a0 = ok ? 1u : 0u;
}
H2G_HD void fast_op_gsearch(const FCtx& C, FState& S) {       // globalGFMSearch :6606 (al_global_search)
	const AlnParams& P = *C.P;
	uint32_t extlen = 0, top = S.a4, bot = S.a5, nr[2] = {0, 0};
	bool uniqueStop = S.a3 != 0;
	GIdx gx; gx.g = C.g;
	const uint32_t nelt = gfm_search(gx, fg_sv(C, S), S.a1, &extlen, &top, &bot, &uniqueStop, C.g->minK, H2G_MAX, P.kseeds, false, nr);
	S.nrank += nr[0]; S.nside += nr[1];
	S.a0 = nelt; S.a1 = extlen; S.a2 = top; S.a3 = bot; S.a4 = uniqueStop ? 1u : 0u;
}
#endif
// S.op stays set when the primitive is not finished (a chunked walk): the slot goes back to the same site's queue
H2G_HD void fast_exec(const FCtx& C, FState& S, const FWords& W, uint32_t op) {
	bool again = false;
	switch(op) {
	case FOP_GSEARCH: fast_op_gsearch(C, S); break;
	case FOP_PSEARCH: fast_op_psearch(C, S); break;
	case FOP_GCOORDS: again = fast_op_gcoords(C, S, W); break;
#if FG_GRAPH
	case FOP_EXTEND:  again = fast_op_extend(C, S, W); break;
#else
	case FOP_EXTEND:  fast_op_extend(C, S, W); break;
#endif
	case FOP_LSEARCH: fast_op_lsearch(C, S); break;
	case FOP_LCOORDS: again = fast_op_lcoords(C, S, W); break;
	case FOP_COMBINE: fast_op_combine(C, S, W); break;
#if FG_GRAPH
	case FOP_ADJUST:    fast_op_adjust(C, S, W); break;
	case FOP_ADJMEMBER: fast_op_adjmember(C, S, W); break;
#endif
	default: break;
	}
	S.op = again ? op : (uint32_t)FOP_NONE;
}

// ---------------------------------------------------------------------------------------- the state between two trips
// The state goes to / comes from its slot word by word.
template <typename ST>   // st(i, v): store word i
H2G_HD void fs_pack(const FState& S, ST&& st) {
	uint32_t w[FS_WORDS];
	memcpy(w, &S, sizeof S);
#pragma unroll
	for(uint32_t i = 0; i < FS_WORDS; i++) st(i, w[i]);
}
template <typename LD>   // ld(i): word i
H2G_HD void fs_unpack(FState& S, LD&& ld) {
	uint32_t w[FS_WORDS];
#pragma unroll
	for(uint32_t i = 0; i < FS_WORDS; i++) w[i] = ld(i);
	memcpy(&S, w, sizeof S);
}

#if FG_ALIGN_MATE
#define FG_SITES_AM(X) X(FOP_LSEARCH, FPC_AM_AFTER_LS) X(FOP_LCOORDS, FPC_AM_AFTER_LC) X(FOP_EXTEND, FPC_AM_EXT_AFTER)
#else
#define FG_SITES_AM(X)
#endif
// Every place the fast machine requests a primitive: (primitive, pc it resumes at).  The queued kernel keeps one queue per site, so
// the lanes of a wave resume at the same pc (as H2G_MACH_SITES of h2g_machine.h).
#if FG_GRAPH
#define FG_SITES_GR(X) X(FOP_ADJUST, FPC_GAH_K_AFTER) X(FOP_ADJMEMBER, FPC_L_RI_A) X(FOP_ADJMEMBER, FPC_L_G_A) X(FOP_ADJMEMBER, FPC_R_RI_A) X(FOP_ADJMEMBER, FPC_R_G_A)
#else
#define FG_SITES_GR(X)
#endif
#define FG_SITES(X) \
	X(FOP_PSEARCH, FPC_NB_AFTER_PS) X(FOP_GCOORDS, FPC_GAH_FULL_AFTER) \
	X(FOP_EXTEND, FPC_HS_EXT_AFTER) X(FOP_EXTEND, FPC_RC_ENTRY_L2) X(FOP_EXTEND, FPC_RC_ENTRY_R2) X(FOP_EXTEND, FPC_L_RI_B) X(FOP_EXTEND, FPC_R_RI_B) \
	X(FOP_EXTEND, FPC_L_EXT_A) X(FOP_EXTEND, FPC_R_EXT_A) \
	X(FOP_LSEARCH, FPC_L_LS_AFTER) X(FOP_LSEARCH, FPC_R_LS_AFTER) X(FOP_LCOORDS, FPC_L_LC_AFTER) X(FOP_LCOORDS, FPC_R_LC_AFTER) \
	X(FOP_COMBINE, FPC_L_RI_C) X(FOP_COMBINE, FPC_R_RI_C) \
	X(FOP_GSEARCH, FPC_L_GS_AFTER) X(FOP_GSEARCH, FPC_R_GS_AFTER) X(FOP_GCOORDS, FPC_L_GC_AFTER) X(FOP_GCOORDS, FPC_R_GC_AFTER) \
	X(FOP_EXTEND, FPC_L_G_B) X(FOP_EXTEND, FPC_R_G_B) X(FOP_COMBINE, FPC_L_G_C) X(FOP_COMBINE, FPC_R_G_C) FG_SITES_AM(X) FG_SITES_GR(X)
enum : uint32_t {
#define X(OPC, PC) FSITE_##PC,
	FSITE_FREE = 0, FG_SITES(X) FSITE_COUNT
#undef X
};
H2G_HD uint32_t fg_site_of(uint32_t pc) {
	switch(pc) {
#define X(OPC, PC) case PC: return FSITE_##PC;
	FG_SITES(X)
#undef X
	default: return 0;
	}
}
H2G_HD uint32_t fg_site_op(uint32_t site) {
	switch(site) {
#define X(OPC, PC) case FSITE_##PC: return OPC;
	FG_SITES(X)
#undef X
	default: return FOP_NONE;
	}
}
// The queues of the kernel: the two hot sites have their own, the other sites share one queue per primitive (their lanes resume at
// different pcs, which the control loop handles anyway): 8 queues of slot ids instead of 25 (LDS: 16 KB instead of 50 KB).
enum : uint32_t { FQ_FREE = 0, FQ_PSEARCH, FQ_GCOORDS, FQ_EXTEND_HS, FQ_EXTEND, FQ_LSEARCH, FQ_LCOORDS, FQ_COMBINE, FQ_GSEARCH,
#if FG_GRAPH
	FQ_ADJUST, FQ_ADJMEMBER,
	FQ_EXTEND_SLOW, FQ_WALK_SLOW,       // requests that turned out to need a primitive's slow form (FCtx::defer_slow): FOP_EXTEND; FOP_LCOORDS / FOP_GCOORDS, each lane its own
#endif
	FQ_COUNT };
H2G_HD uint32_t fg_queue_of(uint32_t pc) {
	const uint32_t site = fg_site_of(pc);
	if(site == 0) return 0;
	if(pc == FPC_HS_EXT_AFTER) return FQ_EXTEND_HS;
	switch(fg_site_op(site)) {
	case FOP_PSEARCH: return FQ_PSEARCH;
	case FOP_GCOORDS: return FQ_GCOORDS;
	case FOP_EXTEND:  return FQ_EXTEND;
	case FOP_LSEARCH: return FQ_LSEARCH;
	case FOP_LCOORDS: return FQ_LCOORDS;
	case FOP_COMBINE: return FQ_COMBINE;
	case FOP_GSEARCH: return FQ_GSEARCH;
#if FG_GRAPH
	case FOP_ADJUST: return FQ_ADJUST;
	case FOP_ADJMEMBER: return FQ_ADJMEMBER;
#endif
	default: return 0;
	}
}
H2G_HD uint32_t fg_queue_op(uint32_t q) {
	switch(q) {
	case FQ_PSEARCH: return FOP_PSEARCH;
	case FQ_GCOORDS: return FOP_GCOORDS;
	case FQ_EXTEND_HS: case FQ_EXTEND: return FOP_EXTEND;
	case FQ_LSEARCH: return FOP_LSEARCH;
	case FQ_LCOORDS: return FOP_LCOORDS;
	case FQ_COMBINE: return FOP_COMBINE;
	case FQ_GSEARCH: return FOP_GSEARCH;
#if FG_GRAPH
	case FQ_ADJUST: return FOP_ADJUST;
	case FQ_ADJMEMBER: return FOP_ADJMEMBER;
	case FQ_EXTEND_SLOW: return FOP_EXTEND;
	case FQ_WALK_SLOW: return FOP_PER_LANE;
#endif
	default: return FOP_NONE;
	}
}
// the queue a read waits in: that of its request site — or, when its primitive asked to be queued again for its slow form, the queue of such requests
H2G_HD uint32_t fg_queue_of_state(const FState& S) {
#if FG_GRAPH
	if(S.op == FOP_EXTEND && (S.a3 >> 31)) return FQ_EXTEND_SLOW;
	if(S.op == FOP_LCOORDS && (S.a2 & 0x40000000u)) return FQ_WALK_SLOW;
	if(S.op == FOP_GCOORDS && (S.a4 & 4u)) return FQ_WALK_SLOW;
#endif
	return fg_queue_of(S.pc);
}

// One read / pair on ONE lane until it completes or bails (tests/emul).  true = completed.
H2G_HD bool fast_run_single(const FCtx& C, FState& S, const FWords& W, uint32_t read, bool paired, bool packed_ok) {
	fast_begin(C, S, read, paired, packed_ok);
	while(S.pc != FPC_DONE && S.pc != FPC_BAIL) {
		if(S.op == FOP_NONE) fast_step(C, S, W);
		if(S.op != FOP_NONE) {
#if !defined(__HIP_DEVICE_COMPILE__)
			// what the queued kernel does between two trips: the state through its packed form (and the site table must know the resume pc)
			uint32_t pw[FS_WORDS];
			const uint32_t op = S.op;
			AL_TRACE(" FAST op %u resume pc %u a %u %u %u %u %u %u sv %u/%u sp %d\n", op, (unsigned)S.pc, S.a0, S.a1, S.a2, S.a3, S.a4, S.a5, (unsigned)S.sv_rdi, (unsigned)S.sv_fw, (int)S.sp);
			if(fg_site_of(S.pc) == 0 || fg_site_op(fg_site_of(S.pc)) != op) { S.pc = FPC_BAIL; S.bail = FB_OTHER; break; }
			fs_pack(S, [&](uint32_t i, uint32_t v) { pw[i] = v; });
			memset(&S, 0x5a, sizeof S);
			fs_unpack(S, [&](uint32_t i) { return pw[i]; });
#endif
			fast_exec(C, S, W, S.op);
		}
	}
	return S.pc == FPC_DONE;
}

}  // namespace h2g
