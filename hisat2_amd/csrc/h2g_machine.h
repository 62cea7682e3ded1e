// h2g_machine.h — HI_Aligner::go as a flat per-lane micro-op machine.
//
// Every read (or pair) is one lane-resident continuation: `mach_step` runs the control flow of go() / nextBWT / align /
// getAnchorHits / hybridSearch / hybridSearch_recur / alignMate (hi_aligner.h:4048-5770, spliced_aligner.h:112-2052) until
// it needs one of a dozen heavy primitives (an FM search, an SA walk, an extension, a combineWith, ...), posts that as an
// op request and yields.  The kernel (h2g_go_kernels.h) then lets the whole wavefront vote, and executes ONE primitive at
// ONE code site for every lane that asked for it — instead of 64 unrelated recursive state machines diverging through
// 27 inlined copies of the same primitives.  Recursion frames, loop variables that live across an op and every list are
// in the lane's AlignWS (HBM); pc / op / op arguments are registers.  `__host__ __device__` like the rest: tests/emul
// runs the identical source one lane at a time.
#pragma once

#if defined(H2G_MACH_PCTRACE) && !defined(__HIP_DEVICE_COMPILE__)
extern "C" void mach_pctrace(unsigned pc);
#endif
#if defined(H2G_MEMPROF) && !defined(__HIP_DEVICE_COMPILE__)
extern "C" { extern const void* g_mp_ws; extern const void* g_mp_mach; extern size_t g_mp_mach_sz; extern int g_mp_phase, g_mp_on; void mp_trip(); }
#endif
// The control function and each primitive are separate functions on the device (H2G_MACH_NOINLINE): one register allocation
// per primitive instead of one over the whole machine.
#ifdef H2G_MACH_NOINLINE
#define H2G_MACH_FN H2G_HDN
#else
#define H2G_MACH_FN H2G_HD
#endif
namespace h2g {

enum : uint32_t {
	OP_NONE = 0,
	OP_PSEARCH,    // partialSearch hi_aligner.h:6361                a0 = cur                                -> ws->fh (+ graph in-edges)
	OP_GCOORDS,    // getGenomeCoords :5774                          a0 top a1 bot a2 maxelt a3 len a4 reject a5 cap a6/a7 node range p0 dst p1 in-edges -> a0 n a1 steps
	OP_EXTEND,     // GenomeHit::extend :2031                        p0 hit a0 mm a1 maxleft a2 maxright     -> a0 leftext a1 rightext
	OP_LSEARCH,    // localGFMSearch :6751                           a0 lidx a1 extoff a2 maxHitLen a3 uniqueStop a4 top a5 bot -> a0 nelt a1 extlen a2 top a3 bot a4 uniqueStop
	OP_LCOORDS,    // getGenomeCoords_local :5861                    a0 lidx a1 top a2 bot a3 rdoff a4 rdlen a5 cap p0 dst -> a0 n
	OP_GSEARCH,    // globalGFMSearch :6606                          a1 extoff a3 uniqueStop a4 top a5 bot   -> as OP_LSEARCH
	OP_COMBINE,    // GenomeHit::combineWith :1420                   p0 this p1 other                        -> a0 combined
	OP_ADJUST,     // static adjustWithALT :2239 (graph)             a0 rdoff a1 len a2 tidx a3 toff a4 joinedOff -> appends to ws->ghits, a0 overflow
	OP_ADJMEMBER,  // member adjustWithALT :2395 (graph)             p0 hit                                  -> a0 ok
	OP_SW,         // SwAligner pass of hybridSearch spliced_aligner.h:209-317   p0 genome hit               -> a0 found
	OP_FINISH,     // selectByScore + result records
	OP_COUNT
};

enum : uint32_t {
	PC_IDLE = 0,
	PC_GO_INIT, PC_NB_PICK, PC_NB_AFTER_PS, PC_ALIGN, PC_AFTER_ALIGN, PC_GO_AFTER_LOOP, PC_PAIR_READS, PC_FINISH, PC_FINISHED,
	PC_GAH_BEGIN, PC_GAH_LOOP, PC_GAH_FULL_AFTER, PC_GAH_SUB_LOOP, PC_GAH_SUB_AFTER, PC_GAH_HAVE, PC_GAH_K_LOOP, PC_GAH_K_AFTER, PC_GAH_END,
	PC_HS_BEGIN, PC_HS_EXT_LOOP, PC_HS_EXT_AFTER, PC_HS_LOOP, PC_HS_AFTER_REC1, PC_HS_AFTER_SW, PC_HS_ONE_DONE,
	PC_MP_LOOP, PC_MP_AFTER_PAIR, PC_AM_WHILE, PC_AM_INNER, PC_AM_AFTER_LS, PC_AM_AFTER_LC, PC_AM_RI_LOOP, PC_AM_RI_AFTER, PC_AM_ADV, PC_AM_EXT_LOOP,
	PC_AM_EXT_AFTER, PC_AM_REC_AFTER,
	// hybridSearch_recur (one Frame per activation)
	PC_RC_ENTRY, PC_RC_ENTRY_L2, PC_RC_ENTRY_L3, PC_RC_ENTRY_R2, PC_RC_ENTRY_R3,
	PC_L_WHILE, PC_L_LS_LOOP, PC_L_LS_AFTER, PC_L_LS_DONE, PC_L_LC_AFTER, PC_L_FOR_RI, PC_L_RI_A, PC_L_RI_B, PC_L_RI_C, PC_L_R1, PC_L_AFTER_FOR,
	PC_L_FOR_TI, PC_L_R2, PC_L_AFTER_WHILE, PC_L_GS_AFTER, PC_L_GC_AFTER, PC_L_FOR_G, PC_L_G_A, PC_L_G_B, PC_L_G_C, PC_L_R3, PC_L_TRIM, PC_L_R4,
	PC_L_EXT, PC_L_EXT_A, PC_L_R5,
	PC_R_WHILE, PC_R_LS_LOOP, PC_R_LS_AFTER, PC_R_LS_DONE, PC_R_LC_AFTER, PC_R_FOR_RI, PC_R_RI_A, PC_R_RI_B, PC_R_RI_C, PC_R_R1, PC_R_AFTER_FOR,
	PC_R_FOR_TI, PC_R_R2, PC_R_AFTER_WHILE, PC_R_GS_AFTER, PC_R_GC_AFTER, PC_R_FOR_G, PC_R_G_A, PC_R_G_B, PC_R_G_C, PC_R_R3, PC_R_TRIM, PC_R_R4,
	PC_R_EXT, PC_R_EXT_A, PC_R_R5,
	// joins through the splice-site database (spliced_aligner.h:409-676 full alignments, :685-811 left, :1365-1496 right)
	PC_FS_L_LOOP, PC_FS_L_EXT, PC_FS_L_COMB, PC_FS_R_I, PC_FS_R_LOOP, PC_FS_R_EXT, PC_FS_R_COMB, PC_FS_REPORT,
	PC_RC_ENTRY_LX, PC_LSS_LOOP, PC_LSS_EXT, PC_LSS_COMB, PC_LSS_RET,
	PC_RC_ENTRY_RX, PC_RSS_LOOP, PC_RSS_EXT, PC_RSS_COMB, PC_RSS_RET
};

// What a finished read leaves behind (the selection half of AlnSinkWrap::finishRead for unpaired reads; the report events of
// both mates + the PRNG state for pairs, whose finishRead runs in h2g_sam_format_paired)
// Every place the machine requests a primitive: (primitive, pc it resumes at).  The kernels queue reads in flight per SITE, not
// per primitive: the lanes of a wave then resume at the same pc, which halves the number of distinct pc bodies a wave walks
// through between two primitives (tools/memprof/simt.py).  tests/test_machine_sites.py checks the list against the M_OP uses.
// The joins through the splice-site database (H2G_SPLICE_DB) are compiled into the units that run spliced alignment only: the
// other kernels keep their rings and their control switch small (the eight extra sites cost the unspliced kernels 8-10 %).
#ifndef H2G_SPLICE_DB
#define H2G_SPLICE_DB 1
#endif
#if H2G_SPLICE_DB
#define H2G_MACH_SITES_DB(X) \
	X(OP_EXTEND, PC_FS_L_EXT) X(OP_COMBINE, PC_FS_L_COMB) X(OP_EXTEND, PC_FS_R_EXT) X(OP_COMBINE, PC_FS_R_COMB) \
	X(OP_EXTEND, PC_LSS_EXT) X(OP_COMBINE, PC_LSS_COMB) X(OP_EXTEND, PC_RSS_EXT) X(OP_COMBINE, PC_RSS_COMB)
#else
#define H2G_MACH_SITES_DB(X)
#endif
#define H2G_MACH_SITES(X) \
	X(OP_PSEARCH, PC_NB_AFTER_PS) \
	X(OP_GCOORDS, PC_GAH_FULL_AFTER) X(OP_GCOORDS, PC_GAH_SUB_AFTER) X(OP_GCOORDS, PC_L_GC_AFTER) X(OP_GCOORDS, PC_R_GC_AFTER) \
	X(OP_EXTEND, PC_HS_EXT_AFTER) X(OP_EXTEND, PC_AM_EXT_AFTER) X(OP_EXTEND, PC_L_EXT_A) X(OP_EXTEND, PC_L_G_B) X(OP_EXTEND, PC_L_RI_B) \
	X(OP_EXTEND, PC_RC_ENTRY_L2) X(OP_EXTEND, PC_RC_ENTRY_R2) X(OP_EXTEND, PC_R_EXT_A) X(OP_EXTEND, PC_R_G_B) X(OP_EXTEND, PC_R_RI_B) \
	X(OP_LSEARCH, PC_AM_AFTER_LS) X(OP_LSEARCH, PC_L_LS_AFTER) X(OP_LSEARCH, PC_R_LS_AFTER) \
	X(OP_LCOORDS, PC_AM_AFTER_LC) X(OP_LCOORDS, PC_L_LC_AFTER) X(OP_LCOORDS, PC_R_LC_AFTER) \
	X(OP_GSEARCH, PC_L_GS_AFTER) X(OP_GSEARCH, PC_R_GS_AFTER) \
	X(OP_COMBINE, PC_L_G_C) X(OP_COMBINE, PC_L_RI_C) X(OP_COMBINE, PC_R_G_C) X(OP_COMBINE, PC_R_RI_C) \
	X(OP_ADJUST, PC_AM_RI_AFTER) X(OP_ADJUST, PC_GAH_K_AFTER) \
	X(OP_ADJMEMBER, PC_L_G_A) X(OP_ADJMEMBER, PC_L_RI_A) X(OP_ADJMEMBER, PC_R_G_A) X(OP_ADJMEMBER, PC_R_RI_A) \
	X(OP_SW, PC_HS_AFTER_SW) H2G_MACH_SITES_DB(X)
enum : uint32_t {
#define X(OPC, PC) SITE_##PC,
	SITE_FREE = 0, H2G_MACH_SITES(X) SITE_COUNT
#undef X
};
H2G_HD uint32_t mach_site_of(uint32_t pc) {        // resume pc -> site (queue) id; 0 = not a resume pc
	switch(pc) {
#define X(OPC, PC) case PC: return SITE_##PC;
	H2G_MACH_SITES(X)
#undef X
	default: return 0;
	}
}
H2G_HD uint32_t mach_site_op(uint32_t site) {      // the primitive a site waits for
	switch(site) {
#define X(OPC, PC) case SITE_##PC: return OPC;
	H2G_MACH_SITES(X)
#undef X
	default: return OP_NONE;
	}
}

struct MachOut {
	ReadOut*    rout;      // unpaired: [n]
	h2g_alnres* aln;       // unpaired: [n * aln_slots]
	uint32_t    aln_slots; // records kept per read (>= -k)
	PairOut*    pout;      // paired: [n]
	h2g_alnres* paln[2];   // paired: [n * pair_slots] each
	uint32_t    pair_slots;
	// a pair whose mates reported more than pair_slots alignments keeps ALL its records in a block of this area (mate 1's, then mate 2's),
	// PairOut::pad = block start + 1 (the sink's lists grow on demand, aln_sink.h:2565; soft-clipped variants of a tandem duplicate's loci)
	h2g_alnres* ovf = nullptr;
	uint32_t*   ovf_cursor = nullptr;
	uint32_t    ovf_cap = 0;
	// a record with more edits than the H2G_MAX_EDITS inline entries of h2g_alnres (the units with the large workspace hold H2G_GHIT_EDITS of them:
	// the reference's lists are unbounded, reportHit hi_aligner.h:6129-6166) keeps its whole list in this area: nedits says how many, edits[0].pos
	// where they start, edits[0].snp == H2G_LONG_EDITS_TAG.  No room: the read is flagged (overflow bit 1, the edit capacity's).
	h2g_edit*   ledits = nullptr;
	uint32_t*   ledits_cursor = nullptr;
	uint32_t    ledits_cap = 0;
	// A pass that is followed by a second pass over the reads whose lists overflowed its workspace (go_run) does not write such a read's partial
	// result: it appends the read to this list and leaves the rows alone.  Every store into the result rows is then FINAL content — runs queued
	// back to back over the same reads (fast passes and several machine passes in flight) write identical bytes, whatever their order.
	uint32_t*   defer_list = nullptr;
	uint32_t*   defer_count = nullptr;
};
#define H2G_LONG_EDITS_TAG 0x4c4f4e47u

struct Lane {                 // the registers of one lane's machine
	uint32_t pc, op;
	uint32_t a0, a1, a2, a3, a4, a5, a6, a7;
	void *p0, *p1;
};

struct Mach {                 // per-lane context (kernel locals; nothing of this lives in HBM)
	Lane      L;
	AlignWS*  ws;
	DReads    rd[2];           // mate-1 / mate-2 read sets (with this lane's LDS pack of the current read)
	const char* name[2];
	uint32_t  namelen[2];
	uint32_t  read;
	uint32_t  ro[2], rl[2];    // offset / length of the current read in each read set (cached: seq_view would load them every time)
	const MachOut* out;        // where a finished read leaves its result
	bool      paired_input;
};

H2G_HD SeqView mach_view(const Mach& M, uint32_t set, bool fw) {   // seq_view() from the cached offset / length
	const DReads& r = M.rd[set];
	SeqView s;
	s.fwc = r.codes + M.ro[set];
	s.q = r.quals ? r.quals + M.ro[set] : nullptr;
	s.len = M.rl[set];
	s.fw = fw;
	if(r.pk && r.pk_read == M.read) { s.pk = r.pk; s.pk_stride = r.pk_stride; }
	return s;
}
H2G_HD SeqView mach_sv(const Mach& M) {
	const GoVars& gv = M.ws->gv;
	return mach_view(M, gv.rd_sel[gv.sv_rdi], gv.sv_fw != 0);
}
H2G_HD void mach_cache_read(Mach& M, uint32_t read, bool paired_input) {
	for(int k = 0; k < (paired_input ? 2 : 1); k++) { M.ro[k] = M.rd[k].offs[read]; M.rl[k] = M.rd[k].offs[read + 1] - M.ro[k]; }
	if(!paired_input) { M.ro[1] = M.ro[0]; M.rl[1] = M.rl[0]; }
}

// The prelude of the worker loop body for one read / pair (hisat2.cpp:3380-3530): filters, PRNG seed, which mates go() sees.
H2G_MACH_FN void mach_begin(Mach& M, uint32_t read, bool paired_input) {
	AlignWS* ws = M.ws;
	GoVars& gv = ws->gv;
	M.read = read;
	M.L.op = OP_NONE;
	gv.read = read;
	ws->m[0].nres = 0; ws->m[1].nres = 0; ws->npairs = 0; ws->overflow = 0; ws->nrank = 0; ws->nsteps = 0; ws->nframes_max = 0; ws->nside = 0;
	mach_cache_read(M, read, paired_input);
	SeqView v1 = mach_view(M, 0, true);
	Rng rnd;
	if(!paired_input) {
		rnd.init(gen_rand_seed(v1, M.name[0], M.namelen[0], 0));       // rnd.init(ps->bufa().seed) hisat2.cpp:3468
		gv.rnd = rnd.last;
		gv.paired = 0; gv.nm = 1; gv.slot0 = 0; gv.rd_sel[0] = 0; gv.rd_sel[1] = 0;
		M.L.pc = read_passes_filters(v1) ? PC_GO_INIT : PC_FINISH;      // filt[0] false: go() is skipped (hisat2.cpp:3518)
		return;
	}
	SeqView v2 = mach_view(M, 1, true);
	const bool f1 = read_passes_filters(v1), f2 = read_passes_filters(v2);
	const uint32_t s1 = gen_rand_seed(v1, M.name[0], M.namelen[0], 0), s2 = gen_rand_seed(v2, M.name[1], M.namelen[1], 0);
	rnd.init((f1 && f2) ? (s1 ^ s2) : s1);                              // hisat2.cpp:3463-3468
	gv.rnd = rnd.last;
	gv.slot0 = 0; gv.rd_sel[0] = 0; gv.rd_sel[1] = 1;
	if(f1 && f2) { gv.paired = 1; gv.nm = 2; M.L.pc = PC_GO_INIT; }
	else if(f1)  { gv.paired = 0; gv.nm = 1; M.L.pc = PC_GO_INIT; }                                   // initRead(rds[0]) hisat2.cpp:3522
	else if(f2)  { gv.paired = 0; gv.nm = 1; gv.slot0 = 1; gv.rd_sel[0] = 1; M.L.pc = PC_GO_INIT; }   // initRead(rds[1], rightendonly) :3524
	else M.L.pc = PC_FINISH;
}

#define M_GOTO(NEXT) do { L.pc = (NEXT); goto again; } while(0)
#define M_OP(OPC, NEXT) do { L.op = (OPC); L.pc = (NEXT); return; } while(0)
#define FR (ws->stack[gv.sp])
#define RC_CALL(HITPTR, HOFF, HLEN, RESUME) do { \
		FR.state = (RESUME); \
		if(gv.sp + 1 >= AL_MAX_DEPTH) { ws->overflow |= 8; gv.ret = INT64_MIN; M_GOTO(RESUME); } \
		else { Frame& nf_ = ws->stack[gv.sp + 1]; hit_copy(&nf_.hit, (HITPTR)); nf_.hitoff = (HOFF); nf_.hitlen = (HLEN); gv.sp++; \
		       if((uint32_t)gv.sp + 1 > ws->nframes_max) { ws->nframes_max = (uint32_t)gv.sp + 1; } \
		       M_GOTO(PC_RC_ENTRY); } } while(0)
#define RC_RET(V) do { gv.ret = (V); gv.sp--; if(gv.sp >= 0) M_GOTO(ws->stack[gv.sp].state); else M_GOTO(gv.rc_ret_pc); } while(0)
// hybridSearch_recur(root, hitoff, hitlen) on the current SeqView / MateWS; control resumes at RETPC with the result in gv.ret
#define RC_START(ROOT, HOFF, HLEN, MINSC, MATE, RETPC) do { \
		Frame& f0_ = ws->stack[0]; hit_copy(&f0_.hit, (ROOT)); f0_.hitoff = (HOFF); f0_.hitlen = (HLEN); \
		gv.sp = 0; gv.rc_minsc = (MINSC); gv.rc_alignMate = (MATE) ? 1u : 0u; gv.rc_ret_pc = (RETPC); gv.ret = INT64_MIN; \
		/* spliced_aligner.h:363-366: cushion = alignMate ? rdlen * 0.03 * sc.mm(255) : 0 (no_spliced_alignment only) */ \
		gv.rc_cushion = (no_spliced && (MATE)) ? (int64_t)((double)mach_sv(M).len * 0.03 * (double)sc.mmpMax) : 0; \
		M_GOTO(PC_RC_ENTRY); } while(0)
#if H2G_EXT_OPTS
#define H2G_XS_ONLY(P_) (((P_).xs_only & 1u) != 0)      // the upper bits carry -I and the pair orientation (aln_params_from)
#else
#define H2G_XS_ONLY(P_) ((P_).xs_only != 0)
#endif
// sink.bestSplicedUnp1/2() (aln_sink.h:2618-2637): the number of introns of the alignment that set bestUnp — the FIRST reported one with
// that score, the update being a strict '>'.  nextBWT (hi_aligner.h:4680) and align (:5520) let a strand run that many more partial
// searches before they give it up against the other strand's best alignment.  Unspliced units never hold a spliced record.
#if H2G_SPLICE_DB
H2G_HD uint32_t best_spliced_unp(const MateWS& mw) {
	if(mw.bestUnp == INT64_MIN) return 0;
	for(uint32_t i = 0; i < mw.nres; i++) if(mw.res[i].score == mw.bestUnp) {
		uint32_t n = 0;
		for(uint32_t k = 0; k < mw.res[i].nedits; k++) n += mw.res[i].edits[k].type == H2G_EDIT_SPL;
		return n;
	}
	return 0;
}
#else
#define best_spliced_unp(MW) 0u
#endif
#define MINSC_LIVE(MV) do { if(!P.secondary) { int64_t b_ = mw->bestUnp - gv.rc_cushion; if(b_ > (MV)) (MV) = b_; } } while(0)

H2G_MACH_FN void mach_finish(const AlnCtx& C, Mach& M);

// Runs the control flow of this lane until it needs a primitive (L.op != OP_NONE) or the read is finished (PC_FINISHED).
H2G_MACH_FN void mach_step(const AlnCtx& C, Mach& M)
{
	Lane& L = M.L;
	AlignWS* ws = M.ws;
	GoVars& gv = ws->gv;
	const AlnParams& P = *C.P;
	const DScoring& sc = P.sc;
	const uint32_t minK = C.g->minK, minK_local = P.minK_local;
	const bool no_spliced = P.no_spliced != 0;
again:
#if defined(H2G_MACH_PCTRACE) && !defined(__HIP_DEVICE_COMPILE__)   // tools/memprof/simt.py: the pcs a read visits between two primitives
	mach_pctrace(L.pc);
#endif
	switch(L.pc) {
	// ======================================================================== go() hi_aligner.h:4048 / nextBWT :4644
	case PC_GO_INIT: {
		ws->nghits = 0; ws->overflow = 0; ws->nrank = 0; ws->nside = 0; ws->nsteps = 0; ws->nframes_max = 0;
		ws->npairs = 0; ws->insp_i = 0; ws->insp_j = 0; ws->bestPair = INT64_MIN; ws->best2Pair = INT64_MIN;
		ws->localindexatts = 0; ws->max_localindexatts = 0;
		gv.rdlens[0] = gv.rdlens[1] = 0;
		for(uint32_t r = 0; r < 2; r++) {
			MateWS& pm = ws->m[r];
			pm.searched = ws->marr[r].searched; pm.res = ws->marr[r].res;
			pm.rb[0].partial = ws->marr[r].partial[0]; pm.rb[1].partial = ws->marr[r].partial[1];
		}
		for(uint32_t r = 0; r < 2; r++) {
			MateWS& mw = ws->m[r ^ gv.slot0];
			mw.nsearched = 0; mw.nres = 0; mw.bestUnp = INT64_MIN; mw.best2Unp = INT64_MIN; mw.minsc = INT64_MAX;
			mw.sink_hidden = (gv.slot0 != 0 && gv.nm == 1) ? 1u : 0u;
			if(r < gv.nm) {
				SeqView v = mach_view(M, gv.rd_sel[r], true);
				gv.rdlens[r] = v.len;
				mw.minsc = min_score_for(P, v.len);   // scoreMin.f<TAlScore>(len) (hisat2.cpp:440, simple_func.h:88)
				for(int k = 0; k < 2; k++) {
					RBHit& h = mw.rb[k];
					h.len = v.len; h.cur = 0; h.done = 0; h.numPartialSearch = 0; h.numUniqueSearch = 0; h.npartial = 0;
				}
			}
		}
		gv.found[0][0] = gv.found[0][1] = 1; gv.found[1][0] = gv.found[1][1] = (uint8_t)gv.paired;
		M_GOTO(PC_NB_PICK);
	}
	case PC_NB_PICK: {                                   // one iteration of nextBWT's loop (:4644-4760)
		int rdi = -1, fwi = -1;
		int64_t maxScore = INT64_MIN;
		for(uint32_t r = 0; r < gv.nm; r++) for(int k = 0; k < 2; k++) {
#if H2G_EXT_OPTS
			{   // _nofw / _norc of the mate in this slot (hisat2.cpp:3449-3452, hi_aligner.h:4875)
				const bool mfw = gv.rd_sel[1] != 1 || ((C.pe_flags >> (r ^ gv.slot0)) & 1u) != 0;   // paired input (also a lone mate of it): gMate1fw / gMate2fw
				const bool gnofw = (C.pe_flags & 4u) != 0, gnorc = (C.pe_flags & 8u) != 0;
				if(k == 0 ? (mfw ? gnofw : gnorc) : (mfw ? gnorc : gnofw)) continue;
			}
#endif
			const RBHit& h = ws->m[r ^ gv.slot0].rb[k];
			if(h.done) continue;
			int64_t cs = rb_search_score(h, minK);
			if(h.cur == 0) cs = INT64_MAX;
			if(cs > maxScore) { maxScore = cs; rdi = (int)r; fwi = k; }
		}
		if(rdi < 0) M_GOTO(PC_GO_AFTER_LOOP);
		MateWS& mw = ws->m[(uint32_t)rdi ^ gv.slot0];
		MateWS& ow = ws->m[(uint32_t)(1 - rdi) ^ gv.slot0];
		RBHit& hit = mw.rb[fwi];
		RBHit& rchit = mw.rb[1 - fwi];
		if(!P.secondary) {
			const uint32_t numSearched = hit.numPartialSearch - hit.numUniqueSearch;
			const int64_t bestScore = mw.bestUnp;
			if(bestScore >= mw.minsc) {
				const uint32_t maxmm = (uint32_t)((-bestScore + sc.mmpMax - 1) / sc.mmpMax);
				if(numSearched > maxmm + best_spliced_unp(mw) + 1) {
					hit.done = 1;
					if(gv.paired) { if(ow.bestUnp >= ow.minsc && ws->npairs > 0) M_GOTO(PC_GO_AFTER_LOOP); else M_GOTO(PC_NB_PICK); }
					else M_GOTO(PC_GO_AFTER_LOOP);
				}
			}
			if(rchit.done && bestScore < mw.minsc) {
				if(numSearched > (rchit.numPartialSearch - rchit.numUniqueSearch) + (P.anchorStop ? 1u : 0u)) { hit.done = 1; M_GOTO(PC_GO_AFTER_LOOP); }
			}
		}
		gv.nb_rdi = rdi; gv.nb_fwi = fwi;
		gv.sv_rdi = (uint32_t)rdi; gv.sv_fw = fwi == 0; gv.mw_slot = (uint32_t)rdi ^ gv.slot0;
		L.a0 = hit.cur;
		M_OP(OP_PSEARCH, PC_NB_AFTER_PS);
	}
	case PC_NB_AFTER_PS: {
		const int rdi = gv.nb_rdi, fwi = gv.nb_fwi;
		MateWS& mw = ws->m[gv.mw_slot];
		RBHit& hit = mw.rb[fwi];
		const h2g_fm_hit& fh = ws->fh;
		if(C.graph && hit.npartial < AL_MAX_PARTIAL) {
			GraphPNode& pn = C.gsl->pnode[gv.mw_slot][fwi][hit.npartial];
			pn.node_top = fh.node_top; pn.node_bot = fh.node_bot; pn.ie = C.gsl->ie;
			if(pn.ie.n > H2G_IEDGE_CAP) { ws->overflow |= 512; AL_TRACE("  cap512 at %s:%d\n", __FILE__, __LINE__); }
		}
		AL_TRACE("  psearch rdi %d fwi %d cur %u -> top %u bot %u len %u type %u cur %u done %u anchor %u\n", rdi, fwi, hit.cur, fh.top, fh.bot, fh.len, fh.hit_type, fh.cur, fh.done, fh.anchorStop);
		ws->nrank += fh.nrank; ws->nside += fh.nside;
		hit.numPartialSearch += 1; hit.numUniqueSearch += fh.numUniqueSearch; hit.cur = fh.cur;
		if(hit.npartial < AL_MAX_PARTIAL) {
			PartialHit& p = hit.partial[hit.npartial++];
			p.top = fh.top; p.bot = fh.bot; p.bwoff = fh.bwoff; p.len = fh.len; p.hit_type = fh.hit_type; p.ncoords = 0;
		} else { ws->overflow |= 32; hit.done = 1; M_GOTO(PC_GO_AFTER_LOOP); }
		if(fh.done) { hit.done = 1; gv.sel_r = rdi; gv.sel_f = fwi; M_GOTO(PC_ALIGN); }
		if(!fh.pseudogeneStop) { if(hit.cur + 1 < hit.len) hit.cur++; }
		if(fh.anchorStop) { hit.done = 1; gv.sel_r = rdi; gv.sel_f = fwi; M_GOTO(PC_ALIGN); }
		M_GOTO(PC_NB_PICK);
	}
	// ======================================================================== align() :5484-5573
	case PC_ALIGN: {
		gv.sv_rdi = (uint32_t)gv.sel_r; gv.sv_fw = gv.sel_f == 0; gv.mw_slot = (uint32_t)gv.sel_r ^ gv.slot0;
		MateWS* mw = &ws->m[gv.mw_slot];
		RBHit& hit = mw->rb[gv.sel_f];
		bool any = false;
		for(uint32_t i = 0; i < hit.npartial; i++) if(!ph_empty(hit.partial[i])) { any = true; break; }
		if(!any) { gv.hs_found = 0; M_GOTO(PC_AFTER_ALIGN); }        // minWidth() == max
		int64_t bestScore = mw->bestUnp;
		if(bestScore < mw->minsc) bestScore = mw->minsc;
		const uint32_t maxmm = (uint32_t)((-bestScore + sc.mmpMax - 1) / sc.mmpMax);
		const uint32_t nact = hit.numPartialSearch - hit.numUniqueSearch;
		if(!P.secondary && nact > maxmm + best_spliced_unp(*mw) + 1) { gv.hs_found = 1; M_GOTO(PC_AFTER_ALIGN); }
		M_GOTO(PC_GAH_BEGIN);
	}
	case PC_AFTER_ALIGN: {
		gv.found[gv.sel_r][gv.sel_f] = (uint8_t)gv.hs_found;
		AL_TRACE(" align rdi %d fwi %d -> found %d nghits %u\n", gv.sel_r, gv.sel_f, (int)gv.hs_found, ws->nghits);
		if(!gv.found[0][0] && !gv.found[0][1] && !gv.found[1][0] && !gv.found[1][1]) M_GOTO(PC_GO_AFTER_LOOP);
		if(gv.paired) { gv.pr_ret_pc = PC_NB_PICK; M_GOTO(PC_PAIR_READS); }
		M_GOTO(PC_NB_PICK);
	}
	case PC_PAIR_READS: {
#if H2G_EXT_OPTS
		al_pair_reads(P, ws, gv.rdlens[0], gv.rdlens[1], C.pe_flags, C.min_frag_len);
#else
		al_pair_reads(P, ws, gv.rdlens[0], gv.rdlens[1]);
#endif
		M_GOTO(gv.pr_ret_pc);
	}
	// no concordant pair: use each mate's alignments as anchors for the other mate (hi_aligner.h:4092-4148)
	case PC_GO_AFTER_LOOP: {
		if(gv.paired && ws->npairs == 0 && (ws->m[0].bestUnp >= ws->m[0].minsc || ws->m[1].bestUnp >= ws->m[1].minsc)) {
			gv.mate_found = 0; gv.mp_rs[0] = ws->m[0].nres; gv.mp_rs[1] = ws->m[1].nres; gv.mp_i = 0; gv.mp_j = 0;
			M_GOTO(PC_MP_LOOP);
		}
		M_GOTO(PC_FINISH);
	}
	case PC_MP_LOOP: {
		if(gv.mp_i >= 2) {
			if(gv.mate_found) { gv.pr_ret_pc = PC_FINISH; M_GOTO(PC_PAIR_READS); }
			M_GOTO(PC_FINISH);
		}
		if(gv.mp_j >= gv.mp_rs[gv.mp_i]) { gv.mp_i++; gv.mp_j = 0; M_GOTO(PC_MP_LOOP); }
		const AlnRec& r = ws->m[gv.mp_i].res[gv.mp_j];
		const bool fw = r.fw != 0;
		AL_TRACE(" alignMate anchor mate %u res %u fw %d toff %u\n", gv.mp_i, gv.mp_j, (int)fw, r.toff);
		// alignMate hi_aligner.h:5579-5770: anchor the OTHER mate near (tidx, toff) through the local index
		gv.am_fw = fw; gv.am_tidx = r.tidx; gv.am_toff = r.toff;
#if H2G_EXT_OPTS
		gv.sv_rdi = 1 - gv.mp_i; gv.sv_fw = (fw == ((C.pe_flags & 2u) != 0)) ? (C.pe_flags & 1u) != 0 : (C.pe_flags & 2u) != 0;   // ofw = (fw == gMate2fw ? gMate1fw : gMate2fw) hi_aligner.h:5605
#else
		gv.sv_rdi = 1 - gv.mp_i; gv.sv_fw = !fw;                     // ofw = (fw == gMate2fw ? gMate1fw : gMate2fw) = !fw
#endif
		gv.mw_slot = 1 - gv.mp_i;
		ws->nghits = 0;
		gv.am_lidx = local_index_of(*C.ls, r.tidx, r.toff);
		gv.am_first = 1; gv.am_count = 0; gv.am_maxhitlen = 0;
		M_GOTO(PC_AM_WHILE);
	}
	case PC_AM_WHILE: {
		if(!(gv.am_count++ < 2)) { gv.am_hi = 0; M_GOTO(PC_AM_EXT_LOOP); }
		if(gv.am_first) gv.am_first = 0;
		else {
			if(ws->nghits > 0) { gv.am_hi = 0; M_GOTO(PC_AM_EXT_LOOP); }
			if(gv.am_lidx != H2G_MAX) gv.am_lidx = gv.am_fw ? local_index_next(*C.ls, gv.am_lidx) : local_index_prev(*C.ls, gv.am_lidx);
			if(gv.am_lidx == H2G_MAX || C.ls->desc[gv.am_lidx].len == 0) { gv.am_hi = 0; M_GOTO(PC_AM_EXT_LOOP); }
		}
		if(gv.am_lidx == H2G_MAX) { gv.am_hi = 0; M_GOTO(PC_AM_EXT_LOOP); }
		gv.am_hitoff = mach_sv(M).len - 1;
		M_GOTO(PC_AM_INNER);
	}
	case PC_AM_INNER: {
		if(!(gv.am_hitoff >= minK_local - 1)) M_GOTO(PC_AM_WHILE);
		if(C.ls->desc[gv.am_lidx].len == 0) { L.a0 = 0; L.a1 = 0; L.a2 = H2G_MAX; L.a3 = H2G_MAX; L.a4 = 0; M_GOTO(PC_AM_AFTER_LS); }
		L.a0 = gv.am_lidx; L.a1 = gv.am_hitoff; L.a2 = 0xffffu; L.a3 = 0; L.a4 = H2G_MAX; L.a5 = H2G_MAX;
		M_OP(OP_LSEARCH, PC_AM_AFTER_LS);
	}
	case PC_AM_AFTER_LS: {
		const uint32_t nelt = L.a0, hitlen = L.a1, top = L.a2, bot = L.a3;
		gv.am_hitlen = hitlen;
		if(nelt > 0 && nelt <= P.kseeds && hitlen > gv.am_maxhitlen) {
			L.a0 = gv.am_lidx; L.a1 = top; L.a2 = bot; L.a3 = gv.am_hitoff - hitlen + 1; L.a4 = hitlen; L.a5 = AL_MAX_GHITS; L.p0 = ws->am_co;
			M_OP(OP_LCOORDS, PC_AM_AFTER_LC);
		}
		M_GOTO(PC_AM_ADV);
	}
	case PC_AM_AFTER_LC: {
		gv.am_nco = L.a0;
		ws->nghits = 0;
		gv.am_ri = 0;
		M_GOTO(PC_AM_RI_LOOP);
	}
	case PC_AM_RI_LOOP: {
		const uint32_t hitoff = gv.am_hitoff, hitlen = gv.am_hitlen, toff = gv.am_toff;
		const h2g_coord* co = ws->am_co;
		for(; gv.am_ri < gv.am_nco; gv.am_ri++) {
			const uint32_t ri = gv.am_ri;
			if(no_spliced) {
				if((uint64_t)co[ri].toff + (uint64_t)P.maxFragLen * 2 < toff || (uint64_t)toff + (uint64_t)P.maxFragLen * 2 < co[ri].toff) continue;
			}
			if(C.graph) {                                            // adjustWithALT (:5692)
				L.a0 = hitoff - hitlen + 1; L.a1 = hitlen; L.a2 = co[ri].tidx; L.a3 = co[ri].toff; L.a4 = co[ri].joinedOff;
				M_OP(OP_ADJUST, PC_AM_RI_AFTER);
			} else if(ws->nghits < AL_MAX_GHITS) hit_init(&ws->ghits[ws->nghits++], gv.sv_fw != 0, hitoff - hitlen + 1, hitlen, co[ri].tidx, co[ri].toff, co[ri].joinedOff);
			else ws->overflow |= 64;
		}
		gv.am_maxhitlen = hitlen;
		M_GOTO(PC_AM_ADV);
	}
	case PC_AM_RI_AFTER: {
		if(L.a0) ws->overflow |= 64;
		gv.am_ri++;
		M_GOTO(PC_AM_RI_LOOP);
	}
	case PC_AM_ADV: {
		if(gv.am_hitlen > 0) gv.am_hitoff -= (gv.am_hitlen - 1);
		if(gv.am_hitoff > 0) gv.am_hitoff -= 1;
		M_GOTO(PC_AM_INNER);
	}
	case PC_AM_EXT_LOOP: {                                 // (genomeHits never exceeds kseeds here: nelt <= kseeds)
		if(gv.am_hi >= ws->nghits) { gv.mate_found = 1; gv.mp_j++; M_GOTO(PC_MP_LOOP); }
		L.p0 = &ws->ghits[gv.am_hi]; L.a0 = 0; L.a1 = H2G_MAX; L.a2 = H2G_MAX;
		M_OP(OP_EXTEND, PC_AM_EXT_AFTER);
	}
	case PC_AM_EXT_AFTER: {
		hit_copy(&ws->tmp2, &ws->ghits[gv.am_hi]);
		RC_START(&ws->tmp2, ws->tmp2.rdoff, ws->tmp2.len, ws->m[gv.mw_slot].minsc, true, PC_AM_REC_AFTER);
	}
	case PC_AM_REC_AFTER: { gv.am_hi++; M_GOTO(PC_AM_EXT_LOOP); }
	// ======================================================================== getAnchorHits :5007-5193
	case PC_GAH_BEGIN: { ws->nghits = 0; gv.gh_hi = 0; M_GOTO(PC_GAH_LOOP); }
	case PC_GAH_LOOP: {
		MateWS* mw = &ws->m[gv.mw_slot];
		const int fwi = gv.sel_f;
		RBHit& hit = mw->rb[fwi];
		const uint32_t maxsz = P.khits > P.kseeds ? P.khits : P.kseeds;
		const uint32_t offsetSize = hit.npartial;
		if(gv.gh_hi >= offsetSize) M_GOTO(PC_GAH_END);
		uint32_t hj = 0;
		for(; hj < offsetSize; hj++) {
			const PartialHit& pj = hit.partial[hj];
			if(ph_empty(pj) || pj.ncoords > 0 || pj.len <= minK + 2) continue;
			else break;
		}
		if(hj >= offsetSize) M_GOTO(PC_GAH_END);
		for(uint32_t hk = hj + 1; hk < offsetSize; hk++) {
			const PartialHit& pk = hit.partial[hk];
			if(ph_empty(pk) || pk.ncoords > 0 || pk.len <= minK + 2) continue;
			const uint32_t tj = hit.partial[hj].hit_type, tk = pk.hit_type, lj = hit.partial[hj].len, lk = pk.len;
			const uint32_t sj = hit.partial[hj].bot - hit.partial[hj].top, sk = pk.bot - pk.top;
			const bool better = tj == tk ? (sj > sk || (sj == sk && lj < lk)) : (tk > tj);
			if(better) hj = hk;
		}
		PartialHit& ph = hit.partial[hj];
		const uint32_t remained = maxsz - ws->nghits;
		if(remained == 0) M_GOTO(PC_GAH_END);
		const GraphPNode* pn = C.graph ? &C.gsl->pnode[gv.mw_slot][fwi][hj] : nullptr;
		const uint32_t expected = C.graph ? pn->node_bot - pn->node_top : ph.bot - ph.top;
		gv.gh_hj = hj; gv.gh_nco = 0; gv.gh_remained = remained;
		gv.gh_rdoff = hit.len - ph.bwoff - ph.len;
		if(expected <= remained) {
			L.a0 = ph.top; L.a1 = ph.bot; L.a2 = ph.bot - ph.top; L.a3 = ph.len; L.a4 = 0; L.a5 = AL_MAX_GHITS; L.p0 = ph.coords;
			if(C.graph) { L.a6 = pn->node_top; L.a7 = pn->node_bot; L.p1 = (void*)&pn->ie; }
			M_OP(OP_GCOORDS, PC_GAH_FULL_AFTER);
		}
		// random sub-sample of `remained` rows (linear) / NODES, each with its own rows and extra in-edges (graph) (:5096-5136)
		gv.gh_expected = expected; gv.gh_top = ph.top; gv.gh_added = 0; gv.gh_edgeIdx = 0;
		gv.gh_node = C.graph ? pn->node_top : ph.top; gv.gh_node_end = C.graph ? pn->node_bot : ph.bot;
		M_GOTO(PC_GAH_SUB_LOOP);
	}
	case PC_GAH_FULL_AFTER: {
		if(L.a1 == H2G_MAX) { { ws->overflow |= 512; AL_TRACE("  cap512 at %s:%d\n", __FILE__, __LINE__); } L.a1 = 0; }
		gv.gh_nco = L.a0;
		ws->nsteps += L.a1;
		M_GOTO(PC_GAH_HAVE);
	}
	case PC_GAH_SUB_LOOP: {
		MateWS* mw = &ws->m[gv.mw_slot];
		PartialHit& ph = mw->rb[gv.sel_f].partial[gv.gh_hj];
		const GraphPNode* pn = C.graph ? &C.gsl->pnode[gv.mw_slot][gv.sel_f][gv.gh_hj] : nullptr;
		Rng rnd; rnd.last = gv.rnd;
		for(; gv.gh_node < gv.gh_node_end; gv.gh_node++, gv.gh_expected--) {
			uint32_t bot = gv.gh_top + 1;
			if(C.graph) {
				IEdges& t = C.gsl->ie;
				t.n = 0;
				if(gv.gh_edgeIdx < pn->ie.n && gv.gh_edgeIdx < H2G_IEDGE_CAP) {
					if(gv.gh_node - pn->node_top == pn->ie.e[gv.gh_edgeIdx][0]) {
						bot += pn->ie.e[gv.gh_edgeIdx][1];
						t.n = 1; t.e[0][0] = 0; t.e[0][1] = pn->ie.e[gv.gh_edgeIdx][1];
						gv.gh_edgeIdx++;
					}
				}
			}
			const uint32_t rndi = rnd.nextU32() % gv.gh_expected;
			if(rndi < gv.gh_remained - gv.gh_added) {
				if(gv.gh_nco < AL_MAX_GHITS) {
					gv.rnd = rnd.last; gv.gh_bot = bot;
					L.a0 = gv.gh_top; L.a1 = bot; L.a2 = ph.bot - ph.top; L.a3 = ph.len; L.a4 = 0; L.a5 = AL_MAX_GHITS - gv.gh_nco; L.p0 = ph.coords + gv.gh_nco;
					if(C.graph) { L.a6 = gv.gh_node; L.a7 = gv.gh_node + 1; L.p1 = (void*)&C.gsl->ie; }
					M_OP(OP_GCOORDS, PC_GAH_SUB_AFTER);
				} else ws->overflow |= 64;
				gv.gh_added++;
				if(gv.gh_added >= gv.gh_remained) break;
			}
			gv.gh_top = bot;
		}
		gv.rnd = rnd.last;
		M_GOTO(PC_GAH_HAVE);
	}
	case PC_GAH_SUB_AFTER: {
		if(L.a1 == H2G_MAX) { { ws->overflow |= 512; AL_TRACE("  cap512 at %s:%d\n", __FILE__, __LINE__); } L.a1 = 0; }
		gv.gh_nco += L.a0;
		ws->nsteps += L.a1;
		gv.gh_added++;
		if(gv.gh_added >= gv.gh_remained) M_GOTO(PC_GAH_HAVE);
		gv.gh_top = gv.gh_bot;
		gv.gh_node++; gv.gh_expected--;
		M_GOTO(PC_GAH_SUB_LOOP);
	}
	case PC_GAH_HAVE: {
		MateWS* mw = &ws->m[gv.mw_slot];
		PartialHit& ph = mw->rb[gv.sel_f].partial[gv.gh_hj];
		const uint32_t maxsz = P.khits > P.kseeds ? P.khits : P.kseeds;
		const uint32_t nco = gv.gh_nco;
		h2g_coord* co = ph.coords;
		AL_TRACE("   anchor hj %u nco %u expected %u remained %u\n", gv.gh_hj, nco, gv.gh_expected, gv.gh_remained);
		ph.ncoords = nco;
		if(nco == 0) { gv.gh_hi++; M_GOTO(PC_GAH_LOOP); }          // !hasGenomeCoords()
		gv.gh_gsize = ws->nghits;
		if(gv.gh_gsize + nco > maxsz) {                              // coords.shufflePortion(0, size, rnd) ds.h:836
			Rng rnd; rnd.last = gv.rnd;
			uint32_t left = nco;
			for(uint32_t i = 0; i + 1 < nco; i++) {
				uint32_t r = rnd.nextU32() % left;
				if(r > 0) { h2g_coord t = co[i]; co[i] = co[i + r]; co[i + r] = t; }
				left--;
			}
			gv.rnd = rnd.last;
		}
		gv.gh_k = 0;
		M_GOTO(PC_GAH_K_LOOP);
	}
	case PC_GAH_K_LOOP: {
		MateWS* mw = &ws->m[gv.mw_slot];
		RBHit& hit = mw->rb[gv.sel_f];
		PartialHit& ph = hit.partial[gv.gh_hj];
		const uint32_t maxsz = P.khits > P.kseeds ? P.khits : P.kseeds;
		const h2g_coord* co = ph.coords;
		const uint32_t rdoff = gv.gh_rdoff;
		const bool svfw = gv.sv_fw != 0;
		for(; gv.gh_k < gv.gh_nco; gv.gh_k++) {
			const uint32_t k = gv.gh_k;
			if(co[k].tidx == H2G_MAX) continue;
			const uint32_t len = ph.len;
			bool overlapped = false;
			for(uint32_t l = 0; l < gv.gh_gsize; l++) {
				h2g_ghit& gh = ws->ghits[l];
				if(gh.tidx != co[k].tidx || (gh.fw != 0) != svfw) continue;
				const uint32_t hitoff = gh.toff + hit.len - gh.rdoff;
				const uint32_t hitoff2 = co[k].toff + hit.len - rdoff;
				const int64_t diff = no_spliced ? 0 : (int64_t)P.maxIntronLen;
				int64_t d = (int64_t)hitoff - (int64_t)hitoff2;
				if(d < 0) d = -d;
				if(d <= diff) { overlapped = true; gh.read++; break; }   // _hitcount++
			}
			if(!overlapped) {
				if(C.graph) {                                        // adjustWithALT may add several (or no) hits (:5175)
					L.a0 = rdoff; L.a1 = len; L.a2 = co[k].tidx; L.a3 = co[k].toff; L.a4 = co[k].joinedOff;
					M_OP(OP_ADJUST, PC_GAH_K_AFTER);
				} else if(ws->nghits < AL_MAX_GHITS) hit_init(&ws->ghits[ws->nghits++], svfw, rdoff, len, co[k].tidx, co[k].toff, co[k].joinedOff);
				else ws->overflow |= 64;
			}
			if(ph.hit_type == H2G_CANDIDATE_HIT && ws->nghits >= maxsz) break;
		}
		if(ph.hit_type == H2G_CANDIDATE_HIT && ws->nghits >= maxsz) M_GOTO(PC_GAH_END);
		gv.gh_hi++;
		M_GOTO(PC_GAH_LOOP);
	}
	case PC_GAH_K_AFTER: {
		const PartialHit& ph = ws->m[gv.mw_slot].rb[gv.sel_f].partial[gv.gh_hj];
		const uint32_t maxsz = P.khits > P.kseeds ? P.khits : P.kseeds;
		if(L.a0) ws->overflow |= 64;
		if(ph.hit_type == H2G_CANDIDATE_HIT && ws->nghits >= maxsz) M_GOTO(PC_GAH_END);
		gv.gh_k++;
		M_GOTO(PC_GAH_K_LOOP);
	}
	case PC_GAH_END: {
		MateWS* mw = &ws->m[gv.mw_slot];
		const uint32_t numHits = ws->nghits;
		if(numHits == 0) { gv.hs_found = 0; M_GOTO(PC_AFTER_ALIGN); }
		const uint64_t add = (uint64_t)((-mw->minsc) / sc.mmpMax) * numHits * (P.secondary ? 2 : 1);
		ws->max_localindexatts = ws->localindexatts + (add > 10 ? add : 10);
		M_GOTO(PC_HS_BEGIN);
	}
	// ======================================================================== hybridSearch spliced_aligner.h:112-322
	case PC_HS_BEGIN: { gv.hs_hi = 0; M_GOTO(PC_HS_EXT_LOOP); }
	case PC_HS_EXT_LOOP: {
		if(gv.hs_hi >= ws->nghits) { gv.hs_hi = 0; M_GOTO(PC_HS_LOOP); }
		L.p0 = &ws->ghits[gv.hs_hi]; L.a0 = 0; L.a1 = H2G_MAX; L.a2 = H2G_MAX;
		M_OP(OP_EXTEND, PC_HS_EXT_AFTER);
	}
	case PC_HS_EXT_AFTER: { ws->ghit_done[gv.hs_hi] = 0; gv.hs_hi++; M_GOTO(PC_HS_EXT_LOOP); }
	case PC_HS_LOOP: {
		if(gv.hs_hi >= ws->nghits) { gv.hs_found = 1; M_GOTO(PC_AFTER_ALIGN); }
		uint32_t hj = 0;
		for(; hj < ws->nghits; hj++) if(!ws->ghit_done[hj]) break;
		if(hj >= ws->nghits) { gv.hs_found = 1; M_GOTO(PC_AFTER_ALIGN); }
		for(uint32_t hk = hj + 1; hk < ws->nghits; hk++) {
			if(ws->ghit_done[hk]) continue;
			// (values into locals, then one bool: the reference-to-element form of this test was miscompiled by hipcc 7.2 -O3 for
			// gfx950 inside this divergent loop — the update of hj was dropped; tests/test_gpu_align.py::test_live_reference[case2])
			const uint32_t ar = ws->ghits[hj].read, al = ws->ghits[hj].len, br = ws->ghits[hk].read, bl = ws->ghits[hk].len;
			const bool better = br > ar || (br == ar && bl > al);
			if(better) hj = hk;
		}
		gv.hs_hj = hj;
		h2g_ghit* gh = &ws->ghits[hj];
		RC_START(gh, gh->rdoff, gh->len, ws->m[gv.mw_slot].minsc, false, PC_HS_AFTER_REC1);
	}
	case PC_HS_AFTER_REC1: {
		// spliced_aligner.h:209-317: the opt-in SwAligner pass (--bowtie2-dp 1: only when nothing reached minsc; 2: always)
		MateWS* mw = &ws->m[gv.mw_slot];
		const int64_t maxsc = gv.ret;
		if(P.bowtie2_dp == 2 || (P.bowtie2_dp == 1 && maxsc < mw->minsc)) {
			h2g_ghit* gh = &ws->ghits[gv.hs_hj];
			const uint32_t svlen = mach_sv(M).len;
			if(gh->len >= svlen) RC_START(gh, gh->rdoff, gh->len, mw->minsc, false, PC_HS_ONE_DONE);
			if(C.sw == nullptr || svlen > H2G_SW_MAX_ROWS) { ws->overflow |= 256; M_GOTO(PC_HS_ONE_DONE); }   // no SW scratch / read longer than the DP path holds
			L.p0 = gh;
			M_OP(OP_SW, PC_HS_AFTER_SW);
		}
		M_GOTO(PC_HS_ONE_DONE);
	}
	case PC_HS_AFTER_SW: {
		if(L.a0) { h2g_ghit* gh = &ws->ghits[gv.hs_hj]; RC_START(gh, gh->rdoff, gh->len, ws->m[gv.mw_slot].minsc, false, PC_HS_ONE_DONE); }
		M_GOTO(PC_HS_ONE_DONE);
	}
	case PC_HS_ONE_DONE: { ws->ghit_done[gv.hs_hj] = 1; gv.hs_hi++; M_GOTO(PC_HS_LOOP); }
	// ======================================================================== hybridSearch_recur spliced_aligner.h:331-2052
	case PC_RC_ENTRY: {
		Frame& f = FR;
		MateWS* mw = &ws->m[gv.mw_slot];
		const h2g_ghit& hit = f.hit;
		const uint32_t hitoff = f.hitoff, hitlen = f.hitlen, dep = (uint32_t)gv.sp, rdlen = mach_sv(M).len;
		const int64_t minsc = gv.rc_minsc;
		AL_TRACE("   recur dep %u fw %u hitoff %u hitlen %u (rdoff %u len %u) toff %u score %lld nedits %u mate %d\n", dep, hit.fw, hitoff, hitlen, hit.rdoff, hit.len, hit.toff, (long long)hit.score, hit.nedits, (int)gv.rc_alignMate);
		f.maxsc = INT64_MIN;
		if(hit.score + gv.rc_cushion < minsc) RC_RET(f.maxsc);
		if(dep >= 128) RC_RET(f.maxsc);
		if(hitoff == hit.rdoff - hit.trim5 && hitlen == hit.len + hit.trim5 + hit.trim3) {
			if(al_is_searched(mw, &hit)) RC_RET(f.maxsc);
			al_add_searched(ws, mw, &hit);
		}
		const bool have_db = H2G_SPLICE_DB && C.ssdb != nullptr && C.ssdb->n != 0;     // !ssdb.empty()
		if(hitoff == 0 && hitlen == rdlen) {
			if(!al_redundant(mw, &hit, rdlen)) {
#if H2G_SPLICE_DB
				if(have_db) {
					// a full alignment: look for the same read joined through database sites near its ends (:409-676);
					// _local_genomeHits[dep] = f.local_hits, best_score = f.prev_score
					f.prev_score = hit.score; f.nlocal = 1; hit_copy(&f.local_hits[0], &hit);
					f.ncoords = 0; f.ri = 0;
					uint32_t fragoff, fraglen, left;
					hit_get_left(&hit, nullptr, nullptr, &fragoff, &fraglen, &left, nullptr);
					if(fraglen >= minK && left >= minK && hit.trim5 == 0 && !no_spliced) {
						f.ncoords = ss_left_sites(*C.ssdb, hit.tidx, left + minK, minK, C.rdid_base + M.read, f.coords, AL_MAX_COORDS);
						if(f.ncoords > AL_MAX_COORDS) { ws->overflow |= 2048; f.ncoords = AL_MAX_COORDS; }
					}
					M_GOTO(PC_FS_L_LOOP);
				}
#endif
				al_report(ws, mw, &hit, rdlen, minsc, H2G_XS_ONLY(P));
				if(hit.score > f.maxsc) f.maxsc = hit.score;
			}
			RC_RET(f.maxsc);
		} else if(hitoff > 0 && (hitoff + hitlen == rdlen || hitoff + hitoff < rdlen - hitlen)) {
			// ---------------- extend to the left: first through database sites (spliced_aligner.h:685-811) ----------------
#if H2G_SPLICE_DB
			f.ncoords = 0; f.ri = 0;
			if(have_db && !no_spliced) {
				uint32_t fragoff, fraglen, left;
				hit_get_left(&hit, nullptr, nullptr, &fragoff, &fraglen, &left, nullptr);
				if(fraglen >= minK_local && left >= minK_local) {
					f.ncoords = ss_left_sites(*C.ssdb, hit.tidx, left + minK_local, minK_local + (minK_local < fragoff ? minK_local : fragoff), C.rdid_base + M.read, f.coords, AL_MAX_COORDS);
					if(f.ncoords > AL_MAX_COORDS) { ws->overflow |= 2048; f.ncoords = AL_MAX_COORDS; }
				}
			}
			M_GOTO(PC_LSS_LOOP);
#else
			M_GOTO(PC_RC_ENTRY_LX);
#endif
		} else {
			// ---------------- extend to the right: first through database sites (:1365-1496) ----------------
#if H2G_SPLICE_DB
			f.ncoords = 0; f.ri = 0;
			if(have_db && !no_spliced) {
				uint32_t fragoff, fraglen, right;
				hit_get_right(&hit, &fragoff, &fraglen, &right);
				if(fraglen >= minK_local) {
					const uint32_t unmapped = rdlen - fragoff - fraglen;
					f.ncoords = ss_right_sites(*C.ssdb, hit.tidx, right + fraglen - minK_local, minK_local + (minK_local < unmapped ? minK_local : unmapped), C.rdid_base + M.read, f.coords, AL_MAX_COORDS);
					if(f.ncoords > AL_MAX_COORDS) { ws->overflow |= 2048; f.ncoords = AL_MAX_COORDS; }
				}
			}
			M_GOTO(PC_RSS_LOOP);
#else
			M_GOTO(PC_RC_ENTRY_RX);
#endif
		}
	}
#if H2G_SPLICE_DB
	// ---- full alignment, left end through a database site (:428-537)
	case PC_FS_L_LOOP: {
		Frame& f = FR;
		const h2g_ghit& hit = f.hit;
		for(; (uint32_t)f.ri < f.ncoords; f.ri++) {
			const h2g_coord ss = f.coords[f.ri];                     // {left, right, dir}
			uint32_t fragoff, fraglen, left;
			hit_get_left(&hit, nullptr, nullptr, &fragoff, &fraglen, &left, nullptr);
			if(left + fraglen - 1 < ss.toff) continue;
			const uint32_t frag2off = ss.tidx - (ss.toff - left);
			if(frag2off + 1 < f.hitoff) continue;
			if(fragoff + ss.toff < left + 1) continue;
			const uint32_t readoff = fragoff + ss.toff - left - 1;
			uint32_t joff = 0;
			if(!text_off_to_joined(*C.g, hit.tidx, ss.tidx, &joff)) continue;
			hit_init(&ws->tmp, hit.fw, readoff + 1, 0, hit.tidx, ss.tidx + 1, joff + 1);
			L.p0 = &ws->tmp; L.a0 = 0; L.a1 = readoff + 1; L.a2 = 0;
			M_OP(OP_EXTEND, PC_FS_L_EXT);
		}
		f.count = f.nlocal; f.ti = 0;                                // num_local_genomeHits (:539)
		M_GOTO(PC_FS_R_I);
	}
	case PC_FS_L_EXT: {
		Frame& f = FR;
		h2g_ghit* t = &ws->tmp;
		if(t->len == 0 || !hit_compatible(t, &f.hit, P.maxIntronLen, no_spliced)) { f.ri++; M_GOTO(PC_FS_L_LOOP); }
		const h2g_coord ss = f.coords[f.ri];
		const int64_t m = gv.rc_minsc > f.prev_score ? gv.rc_minsc : f.prev_score;
		L.p0 = t; L.p1 = &f.hit; L.a1 = 3; L.a2 = ss.tidx; L.a3 = ss.toff; L.a4 = ss.joinedOff; L.a5 = (uint32_t)(uint64_t)m; L.a6 = (uint32_t)((uint64_t)m >> 32);
		M_OP(OP_COMBINE, PC_FS_L_COMB);
	}
	case PC_FS_L_COMB: {
		Frame& f = FR;
		MateWS* mw = &ws->m[gv.mw_slot];
		h2g_ghit* t = &ws->tmp;
		const bool combined = L.a0 != 0;
		if(t->overflow) ws->overflow |= 1;
		int64_t m = gv.rc_minsc > f.prev_score ? gv.rc_minsc : f.prev_score;
		if(mw->bestUnp > m) m = mw->bestUnp;                         // :515-516 (no cushion, whatever --secondary says)
		uint32_t anchor = t->len, ned = 0;                            // getLeftAnchor hi_aligner.h:1040
		for(uint32_t i = 0; i < t->nedits; i++) {
			const h2g_edit e = t->edits[i];
			if(e.type == H2G_EDIT_SPL) { anchor = e.pos; break; }
			if(e.type == H2G_EDIT_MM || is_gap(e.type)) ned++;
		}
		f.ri++;
		if(combined && t->score >= m && ned <= anchor / 4 && !al_is_searched(mw, t) && !al_redundant(mw, t, mach_sv(M).len)) {
			if(t->score > f.prev_score) f.prev_score = t->score;
			if(f.nlocal < AL_MAX_LOCALHITS) hit_copy(&f.local_hits[f.nlocal++], t); else ws->overflow |= 16;
		}
		M_GOTO(PC_FS_L_LOOP);
	}
	// ---- every candidate so far, right end through a database site (:540-656)
	case PC_FS_R_I: {
		Frame& f = FR;
		for(; f.ti < f.count; f.ti++) {
			const h2g_ghit& can = f.local_hits[f.ti];
			if(can.score < f.prev_score) continue;
			uint32_t fragoff, fraglen, right;
			hit_get_right(&can, &fragoff, &fraglen, &right);
			if(!(fraglen >= minK && can.trim3 == 0 && !no_spliced)) continue;
			f.ncoords = ss_right_sites(*C.ssdb, can.tidx, right + fraglen - minK, minK, C.rdid_base + M.read, f.coords, AL_MAX_COORDS);
			if(f.ncoords > AL_MAX_COORDS) { ws->overflow |= 2048; f.ncoords = AL_MAX_COORDS; }
			f.ri = 0;
			M_GOTO(PC_FS_R_LOOP);
		}
		f.ti = 0;
		M_GOTO(PC_FS_REPORT);
	}
	case PC_FS_R_LOOP: {
		Frame& f = FR;
		const h2g_ghit& can = f.local_hits[f.ti];
		const uint32_t rdlen = mach_sv(M).len;
		for(; (uint32_t)f.ri < f.ncoords; f.ri++) {
			const h2g_coord ss = f.coords[f.ri];
			uint32_t fragoff, fraglen, right;
			hit_get_right(&can, &fragoff, &fraglen, &right);
			if(right > ss.tidx) continue;
			const uint32_t readoff = fragoff + ss.tidx - right + 1;
			if(readoff >= rdlen) continue;
			uint32_t joff = 0;
			if(!text_off_to_joined(*C.g, can.tidx, ss.toff, &joff)) continue;
			hit_init(&ws->tmp, can.fw, readoff, 0, can.tidx, ss.toff, joff);
			L.p0 = &ws->tmp; L.a0 = 0; L.a1 = 0; L.a2 = rdlen - readoff;
			M_OP(OP_EXTEND, PC_FS_R_EXT);
		}
		f.ti++;
		M_GOTO(PC_FS_R_I);
	}
	case PC_FS_R_EXT: {
		Frame& f = FR;
		h2g_ghit* t = &ws->tmp;
		const h2g_ghit& can = f.local_hits[f.ti];
		if(t->len == 0 || !hit_compatible(&can, t, P.maxIntronLen, no_spliced)) { f.ri++; M_GOTO(PC_FS_R_LOOP); }
		hit_copy(&ws->tmp2, &can);                                   // combinedHit = canHit
		gv.fs_tscore = t->score;                                     // :643 raises best_score to tempHit's score, not the combined one
		const h2g_coord ss = f.coords[f.ri];
		const int64_t m = gv.rc_minsc > f.prev_score ? gv.rc_minsc : f.prev_score;
		L.p0 = &ws->tmp2; L.p1 = t; L.a1 = 3; L.a2 = ss.tidx; L.a3 = ss.toff; L.a4 = ss.joinedOff; L.a5 = (uint32_t)(uint64_t)m; L.a6 = (uint32_t)((uint64_t)m >> 32);
		M_OP(OP_COMBINE, PC_FS_R_COMB);
	}
	case PC_FS_R_COMB: {
		Frame& f = FR;
		MateWS* mw = &ws->m[gv.mw_slot];
		h2g_ghit* t = &ws->tmp2;
		const bool combined = L.a0 != 0;
		if(t->overflow) ws->overflow |= 1;
		int64_t m = gv.rc_minsc > f.prev_score ? gv.rc_minsc : f.prev_score;
		if(mw->bestUnp > m) m = mw->bestUnp;
		uint32_t anchor = t->len, ned = 0;                            // getRightAnchor :1062
		for(int i = (int)t->nedits - 1; i >= 0; i--) {
			const h2g_edit e = t->edits[i];
			if(e.type == H2G_EDIT_SPL) { anchor = t->len - e.pos - 1; break; }
			if(e.type == H2G_EDIT_MM || is_gap(e.type)) ned++;
		}
		f.ri++;
		if(combined && t->score >= m && ned <= anchor / 4 && !al_is_searched(mw, t) && !al_redundant(mw, t, mach_sv(M).len)) {
			if(t->score > f.prev_score) f.prev_score = gv.fs_tscore;
			if(f.nlocal < AL_MAX_LOCALHITS) hit_copy(&f.local_hits[f.nlocal++], t); else ws->overflow |= 16;
		}
		M_GOTO(PC_FS_R_LOOP);
	}
	case PC_FS_REPORT: {                                         // :658-676
		Frame& f = FR;
		MateWS* mw = &ws->m[gv.mw_slot];
		const uint32_t rdlen = mach_sv(M).len;
		for(uint32_t i = 0; i < f.nlocal; i++) {
			const h2g_ghit* can = &f.local_hits[i];
			if(!P.secondary && can->score < f.prev_score) continue;
			if(i > 0 && !al_is_searched(mw, can)) al_add_searched(ws, mw, can);
			if(!al_redundant(mw, can, rdlen)) {
				al_report(ws, mw, can, rdlen, gv.rc_minsc, H2G_XS_ONLY(P));
				if(can->score > f.maxsc) f.maxsc = can->score;
			}
		}
		RC_RET(f.maxsc);
	}
	// ---- partial alignment, left end through a database site (:697-811)
	case PC_LSS_LOOP: {
		Frame& f = FR;
		const h2g_ghit& hit = f.hit;
		for(; (uint32_t)f.ri < f.ncoords; f.ri++) {
			const h2g_coord ss = f.coords[f.ri];
			uint32_t fragoff, fraglen, left;
			hit_get_left(&hit, nullptr, nullptr, &fragoff, &fraglen, &left, nullptr);
			if(left + fraglen - 1 < ss.toff) continue;
			if(fragoff + ss.toff < left + 1) continue;
			const uint32_t readoff = fragoff + ss.toff - left - 1;
			uint32_t joff = 0;
			if(!text_off_to_joined(*C.g, hit.tidx, ss.tidx, &joff)) continue;
			hit_init(&ws->tmp, hit.fw, readoff + 1, 0, hit.tidx, ss.tidx + 1, joff + 1);
			L.p0 = &ws->tmp; L.a0 = 0; L.a1 = readoff + 1; L.a2 = 0;
			M_OP(OP_EXTEND, PC_LSS_EXT);
		}
		M_GOTO(PC_RC_ENTRY_LX);
	}
	case PC_LSS_EXT: {
		Frame& f = FR;
		h2g_ghit* t = &ws->tmp;
		if(t->len == 0 || !hit_compatible(t, &f.hit, P.maxIntronLen, no_spliced)) { f.ri++; M_GOTO(PC_LSS_LOOP); }
		const h2g_coord ss = f.coords[f.ri];
		L.p0 = t; L.p1 = &f.hit; L.a1 = 1; L.a2 = ss.tidx; L.a3 = ss.toff; L.a4 = ss.joinedOff;
		M_OP(OP_COMBINE, PC_LSS_COMB);
	}
	case PC_LSS_COMB: {
		Frame& f = FR;
		MateWS* mw = &ws->m[gv.mw_slot];
		h2g_ghit* t = &ws->tmp;
		const bool combined = L.a0 != 0;
		if(t->overflow) ws->overflow |= 1;
		int64_t m = gv.rc_minsc;
		MINSC_LIVE(m);
		f.ri++;
		// "soft-clipping might be better" :782
		if(combined && t->score >= m && t->score + (int64_t)sc_penalty(sc, 0) * (int64_t)f.hit.rdoff >= f.hit.score)
			RC_CALL(t, t->rdoff, t->len + t->trim3, PC_LSS_RET);
		M_GOTO(PC_LSS_LOOP);
	}
	case PC_LSS_RET: { Frame& f = FR; if(gv.ret > f.maxsc) f.maxsc = gv.ret; M_GOTO(PC_LSS_LOOP); }
#endif
	case PC_RC_ENTRY_LX: {
		Frame& f = FR;
		const h2g_ghit& hit = f.hit;
		// ---------------- extend to the left (spliced_aligner.h:813-1360) ----------------
		f.use_localindex = 1;
		if(f.hitoff == hit.rdoff && f.hitoff <= minK) {
			hit_copy(&ws->tmp, &hit);
			L.p0 = &ws->tmp; L.a0 = 1; L.a1 = H2G_MAX; L.a2 = 0;
			M_OP(OP_EXTEND, PC_RC_ENTRY_L2);
		}
		M_GOTO(PC_RC_ENTRY_L3);
	}
#if H2G_SPLICE_DB
	// ---- partial alignment, right end through a database site (:1377-1496)
	case PC_RSS_LOOP: {
		Frame& f = FR;
		const h2g_ghit& hit = f.hit;
		const uint32_t rdlen = mach_sv(M).len;
		for(; (uint32_t)f.ri < f.ncoords; f.ri++) {
			const h2g_coord ss = f.coords[f.ri];
			uint32_t fragoff, fraglen, right;
			hit_get_right(&hit, &fragoff, &fraglen, &right);
			if(right > ss.tidx) continue;
			const uint32_t readoff = fragoff + ss.tidx - right + 1;
			if(readoff >= rdlen) continue;
			uint32_t joff = 0;
			if(!text_off_to_joined(*C.g, hit.tidx, ss.toff, &joff)) continue;
			hit_init(&ws->tmp, hit.fw, readoff, 0, hit.tidx, ss.toff, joff);
			L.p0 = &ws->tmp; L.a0 = 0; L.a1 = 0; L.a2 = rdlen - readoff;
			M_OP(OP_EXTEND, PC_RSS_EXT);
		}
		M_GOTO(PC_RC_ENTRY_RX);
	}
	case PC_RSS_EXT: {
		Frame& f = FR;
		h2g_ghit* t = &ws->tmp;
		if(t->len == 0 || !hit_compatible(&f.hit, t, P.maxIntronLen, no_spliced)) { f.ri++; M_GOTO(PC_RSS_LOOP); }
		hit_copy(&ws->tmp2, &f.hit);                                  // combinedHit = hit
		const h2g_coord ss = f.coords[f.ri];
		L.p0 = &ws->tmp2; L.p1 = t; L.a1 = 1; L.a2 = ss.tidx; L.a3 = ss.toff; L.a4 = ss.joinedOff;
		M_OP(OP_COMBINE, PC_RSS_COMB);
	}
	case PC_RSS_COMB: {
		Frame& f = FR;
		MateWS* mw = &ws->m[gv.mw_slot];
		h2g_ghit* t = &ws->tmp2;
		const bool combined = L.a0 != 0;
		if(t->overflow) ws->overflow |= 1;
		int64_t m = gv.rc_minsc;
		MINSC_LIVE(m);
		f.ri++;
		const uint32_t rdlen = mach_sv(M).len;
		if(combined && t->score >= m &&
		   t->score + (int64_t)sc_penalty(sc, 0) * (int64_t)(rdlen - f.hit.rdoff - f.hit.len - f.hit.trim5) >= f.hit.score)
			RC_CALL(t, t->rdoff - t->trim5, t->len + t->trim5, PC_RSS_RET);
		M_GOTO(PC_RSS_LOOP);
	}
	case PC_RSS_RET: { Frame& f = FR; if(gv.ret > f.maxsc) f.maxsc = gv.ret; M_GOTO(PC_RSS_LOOP); }
#endif
	case PC_RC_ENTRY_RX: {
		Frame& f = FR;
		const h2g_ghit& hit = f.hit;
		const uint32_t rdlen = mach_sv(M).len;
		// ---------------- extend to the right (spliced_aligner.h:1496-2050) ----------------
		f.use_localindex = 1;
		if(hit.len == f.hitlen && f.hitoff + f.hitlen + minK > rdlen) {
			hit_copy(&ws->tmp, &hit);
			L.p0 = &ws->tmp; L.a0 = 1; L.a1 = 0; L.a2 = H2G_MAX;
			M_OP(OP_EXTEND, PC_RC_ENTRY_R2);
		}
		M_GOTO(PC_RC_ENTRY_R3);
	}
	case PC_RC_ENTRY_L2: { if(ws->tmp.rdoff == 0) FR.use_localindex = 0; M_GOTO(PC_RC_ENTRY_L3); }
	case PC_RC_ENTRY_L3: {
		Frame& f = FR;
		f.lidx = local_index_of(*C.ls, f.hit.tidx, f.hit.toff);
		f.success = 0; f.first = 1; f.count = 0; f.prev_score = f.hit.score; f.nlocal = 0;
		M_GOTO(PC_L_WHILE);
	}
	case PC_RC_ENTRY_R2: { if(ws->tmp.rdoff + ws->tmp.len == mach_sv(M).len) FR.use_localindex = 0; M_GOTO(PC_RC_ENTRY_R3); }
	case PC_RC_ENTRY_R3: {
		Frame& f = FR;
		f.lidx = local_index_of(*C.ls, f.hit.tidx, f.hit.toff);
		f.success = 0; f.first = 1; f.count = 0; f.prev_score = f.hit.score; f.nlocal = 0;
		M_GOTO(PC_R_WHILE);
	}
	// =============================== LEFT ===============================
	case PC_L_WHILE: {
		Frame& f = FR;
		if(f.success) M_GOTO(PC_L_AFTER_WHILE);
		if(!(f.count++ < 2)) M_GOTO(PC_L_AFTER_WHILE);
		if(!f.use_localindex) M_GOTO(PC_L_AFTER_WHILE);
		if(ws->localindexatts >= ws->max_localindexatts) M_GOTO(PC_L_AFTER_WHILE);
		if(f.first) f.first = 0;
		else {
			f.lidx = f.lidx == H2G_MAX ? H2G_MAX : local_index_prev(*C.ls, f.lidx);
			if(f.lidx == H2G_MAX || C.ls->desc[f.lidx].len == 0) M_GOTO(PC_L_AFTER_WHILE);
		}
		if(f.lidx == H2G_MAX) M_GOTO(PC_L_AFTER_WHILE);
		uint32_t extoff = f.hitoff - 1;
		if(extoff > 0) extoff -= 1;
		if(extoff < P.minAnchorLen) extoff = P.minAnchorLen;
		f.extoff = extoff; f.extlen = 0; f.top = H2G_MAX; f.bot = H2G_MAX; f.nelt = H2G_MAX; f.noext = 0; f.uniqueStop = 0;
		M_GOTO(PC_L_LS_LOOP);
	}
	case PC_L_LS_LOOP: {
		Frame& f = FR;
		if(!(f.extoff < mach_sv(M).len)) M_GOTO(PC_L_LS_DONE);
		f.extlen = 0; f.uniqueStop = 1;
		ws->localindexatts++;
		if(C.ls->desc[f.lidx].len == 0) { L.a0 = 0; L.a1 = 0; L.a2 = f.top; L.a3 = f.bot; L.a4 = 1; M_GOTO(PC_L_LS_AFTER); }
		L.a0 = f.lidx; L.a1 = f.extoff; L.a2 = 0xffffu; L.a3 = 1; L.a4 = f.top; L.a5 = f.bot;
		M_OP(OP_LSEARCH, PC_L_LS_AFTER);
	}
	case PC_L_LS_AFTER: {
		Frame& f = FR;
		f.nelt = L.a0; f.extlen = L.a1; f.top = L.a2; f.bot = L.a3; f.uniqueStop = (uint8_t)L.a4;
		if(f.extoff + 1 - f.extlen >= f.hitoff) { f.noext = 1; M_GOTO(PC_L_LS_DONE); }
		if(f.nelt <= 5) M_GOTO(PC_L_LS_DONE);
		f.extoff++;
		M_GOTO(PC_L_LS_LOOP);
	}
	case PC_L_LS_DONE: {
		Frame& f = FR;
		f.ncoords = 0; f.ri = -1;
		AL_TRACE("    L local lidx %u extoff %u extlen %u nelt %u top %u bot %u unique %d noext %d atts %llu/%llu\n", f.lidx, f.extoff, f.extlen, f.nelt, f.top, f.bot, (int)f.uniqueStop, (int)f.noext, (unsigned long long)ws->localindexatts, (unsigned long long)ws->max_localindexatts);
		if(f.nelt > 0 && f.nelt <= 5 && f.extlen >= P.minAnchorLen && !f.noext) {
			L.a0 = f.lidx; L.a1 = f.top; L.a2 = f.bot; L.a3 = f.extoff + 1 - f.extlen; L.a4 = f.extlen; L.a5 = AL_MAX_COORDS; L.p0 = f.coords;
			M_OP(OP_LCOORDS, PC_L_LC_AFTER);
		}
		M_GOTO(PC_L_FOR_RI);
	}
	case PC_L_LC_AFTER: {
		Frame& f = FR;
		f.ncoords = L.a0;
		sort_coords(f.coords, f.ncoords);
		f.ri = (int)f.ncoords - 1;
		M_GOTO(PC_L_FOR_RI);
	}
	case PC_L_FOR_RI: {
		Frame& f = FR;
		if(f.ri < 0) M_GOTO(PC_L_AFTER_FOR);
		const h2g_coord co = f.coords[f.ri];
		h2g_ghit* t = &ws->tmp;
		hit_init(t, f.hit.fw, f.extoff + 1 - f.extlen, f.extlen, co.tidx, co.toff, co.joinedOff);
		if(C.graph) { L.p0 = t; M_OP(OP_ADJMEMBER, PC_L_RI_A); }
		L.a0 = 1;
		M_GOTO(PC_L_RI_A);
	}
	case PC_L_RI_A: {
		Frame& f = FR;
		h2g_ghit* t = &ws->tmp;
		if(!L.a0) { f.ri--; M_GOTO(PC_L_FOR_RI); }
		if(!hit_compatible(t, &f.hit, P.maxIntronLen, no_spliced)) {
			if(f.count == 1) { f.ri--; M_GOTO(PC_L_FOR_RI); }
			M_GOTO(PC_L_AFTER_FOR);
		}
		if(f.uniqueStop) { L.p0 = t; L.a0 = 0; L.a1 = H2G_MAX; L.a2 = 0; M_OP(OP_EXTEND, PC_L_RI_B); }
		M_GOTO(PC_L_RI_B);
	}
	case PC_L_RI_B: { L.p0 = &ws->tmp; L.p1 = &FR.hit; L.a1 = 0; M_OP(OP_COMBINE, PC_L_RI_C); }
	case PC_L_RI_C: {
		Frame& f = FR;
		MateWS* mw = &ws->m[gv.mw_slot];
		h2g_ghit* t = &ws->tmp;
		const bool combined = L.a0 != 0;
		AL_TRACE("    L combined %d rdoff %u len %u score %lld nedits %u\n", (int)combined, t->rdoff, t->len, (long long)t->score, t->nedits);
		int64_t m = gv.rc_minsc;
		if(t->overflow) ws->overflow |= 1;
		MINSC_LIVE(m);
		f.ri--;
		if(combined && t->score >= m) {
			if(t->score >= f.prev_score - sc.mmpMax) RC_CALL(t, t->rdoff, t->len + t->trim3, PC_L_R1);
			else if(f.nlocal < AL_MAX_LOCALHITS) hit_copy(&f.local_hits[f.nlocal++], t);
			else ws->overflow |= 16;
		}
		M_GOTO(PC_L_FOR_RI);
	}
	case PC_L_R1: { Frame& f = FR; if(gv.ret > f.maxsc) f.maxsc = gv.ret; M_GOTO(PC_L_FOR_RI); }
	case PC_L_AFTER_FOR: {
		Frame& f = FR;
		if(f.maxsc >= f.prev_score - sc.mmpMax) f.success = 1;
		f.ti = 0;
		if(!f.success && (ws->localindexatts >= ws->max_localindexatts || f.count == 2 ||
		                  (f.lidx == H2G_MAX || local_index_prev(*C.ls, f.lidx) == H2G_MAX)))
			M_GOTO(PC_L_FOR_TI);
		M_GOTO(PC_L_WHILE);
	}
	case PC_L_FOR_TI: {
		Frame& f = FR;
		MateWS* mw = &ws->m[gv.mw_slot];
		if(f.ti >= f.nlocal) M_GOTO(PC_L_WHILE);
		h2g_ghit* t = &f.local_hits[f.ti++];
		int64_t m = gv.rc_minsc;
		MINSC_LIVE(m);
		if(t->score >= m) RC_CALL(t, t->rdoff, t->len + t->trim3, PC_L_R2);
		M_GOTO(PC_L_FOR_TI);
	}
	case PC_L_R2: { Frame& f = FR; if(gv.ret > f.maxsc) f.maxsc = gv.ret; M_GOTO(PC_L_FOR_TI); }
	case PC_L_AFTER_WHILE: {
		Frame& f = FR;
		if(f.success) RC_RET(f.maxsc);
		f.ncoords = 0; f.ri = -1;
		if(f.hitoff > minK && ws->localindexatts < ws->max_localindexatts) {   // global search for long introns (:1085)
			f.extoff = f.hitoff - 1; f.extlen = 0;
			L.a1 = f.extoff; L.a3 = 1; L.a4 = H2G_MAX; L.a5 = H2G_MAX;
			M_OP(OP_GSEARCH, PC_L_GS_AFTER);
		}
		M_GOTO(PC_L_FOR_G);
	}
	case PC_L_GS_AFTER: {
		Frame& f = FR;
		const uint32_t nelt = L.a0, top = L.a2, bot = L.a3;
		f.extlen = L.a1; f.uniqueStop = (uint8_t)L.a4;
		AL_TRACE("    L global extoff %u extlen %u nelt %u top %u bot %u unique %d\n", f.extoff, f.extlen, nelt, top, bot, (int)f.uniqueStop);
		if(nelt > 0 && nelt <= 5 && f.extlen >= minK) {
			L.a0 = top; L.a1 = bot; L.a2 = bot - top; L.a3 = f.extlen; L.a4 = 1; L.a5 = AL_MAX_COORDS; L.p0 = f.coords;
			if(C.graph) { L.a6 = C.gsl->node_top; L.a7 = C.gsl->node_bot; L.p1 = (void*)&C.gsl->ie; }
			M_OP(OP_GCOORDS, PC_L_GC_AFTER);
		}
		M_GOTO(PC_L_FOR_G);
	}
	case PC_L_GC_AFTER: {
		Frame& f = FR;
		if(L.a1 == H2G_MAX) { { ws->overflow |= 512; AL_TRACE("  cap512 at %s:%d\n", __FILE__, __LINE__); } L.a1 = 0; }
		ws->nsteps += L.a1;
		f.ncoords = L.a0;
		if(f.ncoords > 1) sort_coords(f.coords, f.ncoords);
		f.ri = (int)f.ncoords - 1;
		M_GOTO(PC_L_FOR_G);
	}
	case PC_L_FOR_G: {
		Frame& f = FR;
		if(f.ri < 0) M_GOTO(PC_L_TRIM);
		const h2g_coord co = f.coords[f.ri];
		f.ri--;
		h2g_ghit* t = &ws->tmp;
		hit_init(t, f.hit.fw, f.extoff + 1 - f.extlen, f.extlen, co.tidx, co.toff, co.joinedOff);
		if(C.graph) { L.p0 = t; M_OP(OP_ADJMEMBER, PC_L_G_A); }
		L.a0 = 1;
		M_GOTO(PC_L_G_A);
	}
	case PC_L_G_A: {
		Frame& f = FR;
		h2g_ghit* t = &ws->tmp;
		if(!L.a0) M_GOTO(PC_L_FOR_G);
		if(!hit_compatible(t, &f.hit, P.maxIntronLen, no_spliced)) M_GOTO(PC_L_FOR_G);
		if(f.uniqueStop) { L.p0 = t; L.a0 = 0; L.a1 = H2G_MAX; L.a2 = 0; M_OP(OP_EXTEND, PC_L_G_B); }
		M_GOTO(PC_L_G_B);
	}
	case PC_L_G_B: { L.p0 = &ws->tmp; L.p1 = &FR.hit; L.a1 = 0; M_OP(OP_COMBINE, PC_L_G_C); }
	case PC_L_G_C: {
		MateWS* mw = &ws->m[gv.mw_slot];
		h2g_ghit* t = &ws->tmp;
		const bool combined = L.a0 != 0;
		int64_t m = gv.rc_minsc;
		if(t->overflow) ws->overflow |= 1;
		MINSC_LIVE(m);
		if(combined && t->score >= m) RC_CALL(t, t->rdoff, t->len + t->trim3, PC_L_R3);
		M_GOTO(PC_L_FOR_G);
	}
	case PC_L_R3: { Frame& f = FR; if(gv.ret > f.maxsc) f.maxsc = gv.ret; M_GOTO(PC_L_FOR_G); }
	case PC_L_TRIM: {
		Frame& f = FR;
		const h2g_ghit& hit = f.hit;
		const int64_t minsc = gv.rc_minsc;
		const int64_t floor_ = f.maxsc > minsc ? f.maxsc : minsc;
		const int64_t tm = (hit.score - floor_) / sc_penalty(sc, 0);
		const uint32_t trimMax = (uint32_t)tm;
		if(hit.rdoff < trimMax) {
			h2g_ghit* t = &ws->tmp;
			hit_copy(t, &hit);
			t->trim5 = hit.rdoff;                         // GenomeHit::trim5 hi_aligner.h:831
			calculate_score(sc, mach_sv(M), t);
			if(t->score > f.maxsc && t->score >= minsc) RC_CALL(t, 0, t->len + t->trim5 + t->trim3, PC_L_R4);
		}
		M_GOTO(PC_L_EXT);
	}
	case PC_L_R4: { Frame& f = FR; if(gv.ret > f.maxsc) f.maxsc = gv.ret; M_GOTO(PC_L_EXT); }
	case PC_L_EXT: {
		Frame& f = FR;
		h2g_ghit* t = &ws->tmp;
		hit_copy(t, &f.hit);
		const uint32_t mm = (uint32_t)((t->score - gv.rc_minsc) / sc.mmpMax);
		uint32_t nmm = 1;
		if(f.hitoff <= minK_local) nmm = t->rdoff < mm ? t->rdoff : mm;
		AL_TRACE("    L ext from rdoff %u len %u toff %u joff %u nmm %u\n", t->rdoff, t->len, t->toff, t->joinedOff, nmm);
		L.p0 = t; L.a0 = nmm; L.a1 = H2G_MAX; L.a2 = 0;
		M_OP(OP_EXTEND, PC_L_EXT_A);
	}
	case PC_L_EXT_A: {
		Frame& f = FR;
		MateWS* mw = &ws->m[gv.mw_slot];
		const h2g_ghit& hit = f.hit;
		h2g_ghit* t = &ws->tmp;
		const uint32_t le = L.a0, hitoff = f.hitoff, hitlen = f.hitlen;
		int64_t m = gv.rc_minsc;
		AL_TRACE("    L ext -> rdoff %u len %u toff %u joff %u score %lld nedits %u le %u\n", t->rdoff, t->len, t->toff, t->joinedOff, (long long)t->score, t->nedits, le);
		if(t->overflow) ws->overflow |= 1;
		MINSC_LIVE(m);
		const uint32_t need = minK_local < hit.rdoff ? minK_local : hit.rdoff;
		if(t->score >= m && le >= need) RC_CALL(t, t->rdoff, t->len + t->trim3, PC_L_R5);
		else if(hitoff > minK_local) {
			const uint32_t jumplen = hitoff > minK ? minK : minK_local;
			const int64_t expected = hit.score - (int64_t)((hit.rdoff - hitoff) / jumplen) * sc.mmpMax - sc.mmpMax;
			if(expected >= m) RC_CALL(&hit, hitoff - jumplen, hitlen + jumplen, PC_L_R5);
		}
		RC_RET(f.maxsc);
	}
	case PC_L_R5: { Frame& f = FR; if(gv.ret > f.maxsc) f.maxsc = gv.ret; RC_RET(f.maxsc); }
	// =============================== RIGHT ===============================
	case PC_R_WHILE: {
		Frame& f = FR;
		const uint32_t rdlen = mach_sv(M).len;
		if(f.success) M_GOTO(PC_R_AFTER_WHILE);
		if(!(f.count++ < 2)) M_GOTO(PC_R_AFTER_WHILE);
		if(!f.use_localindex) M_GOTO(PC_R_AFTER_WHILE);
		if(ws->localindexatts >= ws->max_localindexatts) M_GOTO(PC_R_AFTER_WHILE);
		if(f.first) f.first = 0;
		else {
			f.lidx = f.lidx == H2G_MAX ? H2G_MAX : local_index_next(*C.ls, f.lidx);
			if(f.lidx == H2G_MAX || C.ls->desc[f.lidx].len == 0) M_GOTO(PC_R_AFTER_WHILE);
		}
		if(f.lidx == H2G_MAX) M_GOTO(PC_R_AFTER_WHILE);
		uint32_t extoff = f.hitoff + f.hitlen + minK_local;
		if(extoff + 1 < rdlen) extoff += 1;
		if(extoff >= rdlen) extoff = rdlen - 1;
		uint32_t maxHitLen = extoff - f.hitoff - f.hitlen;
		if(maxHitLen < minK_local) maxHitLen = minK_local;
		f.extoff = extoff; f.extlen = 0; f.top = H2G_MAX; f.bot = H2G_MAX; f.nelt = H2G_MAX; f.noext = 0; f.uniqueStop = 0; f.maxHitLen = maxHitLen;
		M_GOTO(PC_R_LS_LOOP);
	}
	case PC_R_LS_LOOP: {
		Frame& f = FR;
		if(!(f.maxHitLen < f.extoff + 1 && f.extoff < mach_sv(M).len)) M_GOTO(PC_R_LS_DONE);
		f.extlen = 0; f.uniqueStop = 0;
		ws->localindexatts++;
		if(C.ls->desc[f.lidx].len == 0) { L.a0 = 0; L.a1 = 0; L.a2 = f.top; L.a3 = f.bot; L.a4 = 0; M_GOTO(PC_R_LS_AFTER); }
		L.a0 = f.lidx; L.a1 = f.extoff; L.a2 = f.maxHitLen; L.a3 = 0; L.a4 = f.top; L.a5 = f.bot;
		M_OP(OP_LSEARCH, PC_R_LS_AFTER);
	}
	case PC_R_LS_AFTER: {
		Frame& f = FR;
		const uint32_t rdlen = mach_sv(M).len;
		f.nelt = L.a0; f.extlen = L.a1; f.top = L.a2; f.bot = L.a3; f.uniqueStop = (uint8_t)L.a4;
		if(f.extoff < f.hitoff + f.hitlen) { f.noext = 1; M_GOTO(PC_R_LS_DONE); }
		if(f.nelt <= 5) M_GOTO(PC_R_LS_DONE);
		if(f.extoff + 1 < rdlen) f.extoff++;
		else { if(f.extlen < f.maxHitLen) M_GOTO(PC_R_LS_DONE); else f.maxHitLen++; }
		M_GOTO(PC_R_LS_LOOP);
	}
	case PC_R_LS_DONE: {
		Frame& f = FR;
		f.ncoords = 0; f.ri = 0;
		AL_TRACE("    R local lidx %u extoff %u extlen %u nelt %u top %u bot %u unique %d noext %d atts %llu/%llu\n", f.lidx, f.extoff, f.extlen, f.nelt, f.top, f.bot, (int)f.uniqueStop, (int)f.noext, (unsigned long long)ws->localindexatts, (unsigned long long)ws->max_localindexatts);
		if(f.nelt > 0 && f.nelt <= 5 && f.extlen >= P.minAnchorLen && !f.noext) {
			L.a0 = f.lidx; L.a1 = f.top; L.a2 = f.bot; L.a3 = f.extoff + 1 - f.extlen; L.a4 = f.extlen; L.a5 = AL_MAX_COORDS; L.p0 = f.coords;
			M_OP(OP_LCOORDS, PC_R_LC_AFTER);
		}
		M_GOTO(PC_R_FOR_RI);
	}
	case PC_R_LC_AFTER: {
		Frame& f = FR;
		f.ncoords = L.a0;
		if(f.ncoords > 1) sort_coords(f.coords, f.ncoords);
		M_GOTO(PC_R_FOR_RI);
	}
	case PC_R_FOR_RI: {
		Frame& f = FR;
		if(f.ri >= (int)f.ncoords) M_GOTO(PC_R_AFTER_FOR);
		const h2g_coord co = f.coords[f.ri];
		h2g_ghit* t = &ws->tmp;
		hit_init(t, f.hit.fw, f.extoff + 1 - f.extlen, f.extlen, co.tidx, co.toff, co.joinedOff);
		if(C.graph) { L.p0 = t; M_OP(OP_ADJMEMBER, PC_R_RI_A); }
		L.a0 = 1;
		M_GOTO(PC_R_RI_A);
	}
	case PC_R_RI_A: {
		Frame& f = FR;
		h2g_ghit* t = &ws->tmp;
		if(!L.a0) { f.ri++; M_GOTO(PC_R_FOR_RI); }
		if(!hit_compatible(&f.hit, t, P.maxIntronLen, no_spliced)) {
			if(f.count == 1) { f.ri++; M_GOTO(PC_R_FOR_RI); }
			M_GOTO(PC_R_AFTER_FOR);
		}
		L.p0 = t; L.a0 = 0; L.a1 = 0; L.a2 = H2G_MAX;
		M_OP(OP_EXTEND, PC_R_RI_B);
	}
	case PC_R_RI_B: {
		hit_copy(&ws->tmp2, &FR.hit);
		L.p0 = &ws->tmp2; L.p1 = &ws->tmp;
		L.a1 = 0;
		M_OP(OP_COMBINE, PC_R_RI_C);
	}
	case PC_R_RI_C: {
		Frame& f = FR;
		MateWS* mw = &ws->m[gv.mw_slot];
		h2g_ghit* cmb = &ws->tmp2;
		const bool combined = L.a0 != 0;
		int64_t m = gv.rc_minsc;
		if(cmb->overflow) ws->overflow |= 1;
		MINSC_LIVE(m);
		f.ri++;
		if(combined && cmb->score >= m) {
			if(cmb->score >= f.prev_score - sc.mmpMax) RC_CALL(cmb, cmb->rdoff - cmb->trim5, cmb->len + cmb->trim5, PC_R_R1);
			else if(f.nlocal < AL_MAX_LOCALHITS) hit_copy(&f.local_hits[f.nlocal++], cmb);
			else ws->overflow |= 16;
		}
		M_GOTO(PC_R_FOR_RI);
	}
	case PC_R_R1: { Frame& f = FR; if(gv.ret > f.maxsc) f.maxsc = gv.ret; M_GOTO(PC_R_FOR_RI); }
	case PC_R_AFTER_FOR: {
		Frame& f = FR;
		if(f.maxsc >= f.prev_score - sc.mmpMax) f.success = 1;
		f.ti = 0;
		if(!f.success && (ws->localindexatts >= ws->max_localindexatts || f.count == 2 ||
		                  (f.lidx == H2G_MAX || local_index_next(*C.ls, f.lidx) == H2G_MAX)))
			M_GOTO(PC_R_FOR_TI);
		M_GOTO(PC_R_WHILE);
	}
	case PC_R_FOR_TI: {
		Frame& f = FR;
		MateWS* mw = &ws->m[gv.mw_slot];
		if(f.ti >= f.nlocal) M_GOTO(PC_R_WHILE);
		h2g_ghit* t = &f.local_hits[f.ti++];
		int64_t m = gv.rc_minsc;
		MINSC_LIVE(m);
		if(t->score >= m) RC_CALL(t, t->rdoff - t->trim5, t->len + t->trim5, PC_R_R2);
		M_GOTO(PC_R_FOR_TI);
	}
	case PC_R_R2: { Frame& f = FR; if(gv.ret > f.maxsc) f.maxsc = gv.ret; M_GOTO(PC_R_FOR_TI); }
	case PC_R_AFTER_WHILE: {
		Frame& f = FR;
		if(f.success) RC_RET(f.maxsc);
		f.ncoords = 0; f.ri = 0;
		if(f.hitoff + f.hitlen + minK + 1 < mach_sv(M).len && ws->localindexatts < ws->max_localindexatts) {
			f.extoff = f.hitoff + f.hitlen + minK + 1; f.extlen = 0;
			L.a1 = f.extoff; L.a3 = 1; L.a4 = H2G_MAX; L.a5 = H2G_MAX;
			M_OP(OP_GSEARCH, PC_R_GS_AFTER);
		}
		M_GOTO(PC_R_FOR_G);
	}
	case PC_R_GS_AFTER: {
		Frame& f = FR;
		const uint32_t nelt = L.a0, top = L.a2, bot = L.a3;
		f.extlen = L.a1; f.uniqueStop = (uint8_t)L.a4;
		if(nelt > 0 && nelt <= 5 && f.extlen >= minK) {
			L.a0 = top; L.a1 = bot; L.a2 = bot - top; L.a3 = f.extlen; L.a4 = 1; L.a5 = AL_MAX_COORDS; L.p0 = f.coords;
			if(C.graph) { L.a6 = C.gsl->node_top; L.a7 = C.gsl->node_bot; L.p1 = (void*)&C.gsl->ie; }
			M_OP(OP_GCOORDS, PC_R_GC_AFTER);
		}
		M_GOTO(PC_R_FOR_G);
	}
	case PC_R_GC_AFTER: {
		Frame& f = FR;
		if(L.a1 == H2G_MAX) { { ws->overflow |= 512; AL_TRACE("  cap512 at %s:%d\n", __FILE__, __LINE__); } L.a1 = 0; }
		ws->nsteps += L.a1;
		f.ncoords = L.a0;
		sort_coords(f.coords, f.ncoords);
		M_GOTO(PC_R_FOR_G);
	}
	case PC_R_FOR_G: {
		Frame& f = FR;
		if(f.ri >= (int)f.ncoords) M_GOTO(PC_R_TRIM);
		const h2g_coord co = f.coords[f.ri];
		f.ri++;
		h2g_ghit* t = &ws->tmp;
		hit_init(t, f.hit.fw, f.extoff + 1 - f.extlen, f.extlen, co.tidx, co.toff, co.joinedOff);
		if(C.graph) { L.p0 = t; M_OP(OP_ADJMEMBER, PC_R_G_A); }
		L.a0 = 1;
		M_GOTO(PC_R_G_A);
	}
	case PC_R_G_A: {
		Frame& f = FR;
		h2g_ghit* t = &ws->tmp;
		if(!L.a0) M_GOTO(PC_R_FOR_G);
		if(!hit_compatible(&f.hit, t, P.maxIntronLen, no_spliced)) M_GOTO(PC_R_FOR_G);
		L.p0 = t; L.a0 = 0; L.a1 = 0; L.a2 = H2G_MAX;
		M_OP(OP_EXTEND, PC_R_G_B);
	}
	case PC_R_G_B: {
		hit_copy(&ws->tmp2, &FR.hit);
		L.p0 = &ws->tmp2; L.p1 = &ws->tmp;
		L.a1 = 0;
		M_OP(OP_COMBINE, PC_R_G_C);
	}
	case PC_R_G_C: {
		MateWS* mw = &ws->m[gv.mw_slot];
		h2g_ghit* cmb = &ws->tmp2;
		const bool combined = L.a0 != 0;
		int64_t m = gv.rc_minsc;
		if(cmb->overflow) ws->overflow |= 1;
		MINSC_LIVE(m);
		if(combined && cmb->score >= m) RC_CALL(cmb, cmb->rdoff - cmb->trim5, cmb->len + cmb->trim5, PC_R_R3);
		M_GOTO(PC_R_FOR_G);
	}
	case PC_R_R3: { Frame& f = FR; if(gv.ret > f.maxsc) f.maxsc = gv.ret; M_GOTO(PC_R_FOR_G); }
	case PC_R_TRIM: {
		Frame& f = FR;
		const h2g_ghit& hit = f.hit;
		const int64_t minsc = gv.rc_minsc;
		const uint32_t trimLen = mach_sv(M).len - f.hitoff - hit.len - hit.trim5;
		const int64_t floor_ = f.maxsc > minsc ? f.maxsc : minsc;
		const uint32_t trimMax = (uint32_t)((hit.score - floor_) / sc_penalty(sc, 0));
		if(trimLen < trimMax) {
			h2g_ghit* t = &ws->tmp;
			hit_copy(t, &hit);
			t->trim3 = trimLen;                           // GenomeHit::trim3 hi_aligner.h:855
			calculate_score(sc, mach_sv(M), t);
			if(t->score > f.maxsc && t->score >= minsc) RC_CALL(t, t->rdoff - t->trim5, t->len + t->trim5 + t->trim3, PC_R_R4);
		}
		M_GOTO(PC_R_EXT);
	}
	case PC_R_R4: { Frame& f = FR; if(gv.ret > f.maxsc) f.maxsc = gv.ret; M_GOTO(PC_R_EXT); }
	case PC_R_EXT: {
		Frame& f = FR;
		h2g_ghit* t = &ws->tmp;
		hit_copy(t, &f.hit);
		const uint32_t rdlen = mach_sv(M).len;
		const uint32_t mm = (uint32_t)((t->score - gv.rc_minsc) / sc.mmpMax);
		uint32_t nmm = 1;
		if(rdlen - f.hitoff - f.hitlen <= minK_local) {
			const uint32_t rest = rdlen - t->rdoff - t->len;
			nmm = rest < mm ? rest : mm;
		}
		L.p0 = t; L.a0 = nmm; L.a1 = 0; L.a2 = H2G_MAX;
		M_OP(OP_EXTEND, PC_R_EXT_A);
	}
	case PC_R_EXT_A: {
		Frame& f = FR;
		MateWS* mw = &ws->m[gv.mw_slot];
		const h2g_ghit& hit = f.hit;
		h2g_ghit* t = &ws->tmp;
		const uint32_t re = L.a1, hitoff = f.hitoff, hitlen = f.hitlen, rdlen = mach_sv(M).len;
		int64_t m = gv.rc_minsc;
		if(t->overflow) ws->overflow |= 1;
		MINSC_LIVE(m);
		const uint32_t rest0 = rdlen - hit.len - hit.rdoff;
		const uint32_t need = minK_local < rest0 ? minK_local : rest0;
		if(t->score >= m && re >= need) RC_CALL(t, t->rdoff - t->trim5, t->len + t->trim5, PC_R_R5);
		else if(hitoff + hitlen + minK_local < rdlen) {
			const uint32_t jumplen = hitoff + hitlen + minK < rdlen ? minK : minK_local;
			const int64_t expected = hit.score - (int64_t)((hitlen - hit.len) / jumplen) * sc.mmpMax - sc.mmpMax;
			if(expected >= m) RC_CALL(&hit, hitoff, hitlen + jumplen, PC_R_R5);
		}
		RC_RET(f.maxsc);
	}
	case PC_R_R5: { Frame& f = FR; if(gv.ret > f.maxsc) f.maxsc = gv.ret; RC_RET(f.maxsc); }
	// ========================================================================
	case PC_FINISH: {                                    // selectByScore + the result records, inline (no primitive worth a trip)
		mach_finish(C, M);
		L.pc = PC_FINISHED; L.op = OP_NONE;
		return;
	}
	case PC_FINISHED:
	default: return;
	}
}
#undef M_GOTO
#undef M_OP
#undef FR
#undef RC_CALL
#undef RC_RET
#undef RC_START
#undef MINSC_LIVE

// ---------------------------------------------------------------------------------------- the primitives
// Each executes for the lanes that requested it; the kernel calls mach_exec with a wave-uniform `op`, so every body below
// is one code site shared by all requesters whatever control state they came from.
H2G_MACH_FN void mach_op_psearch(const AlnCtx& C, Mach& M) {
	const AlnParams& P = *C.P;
	const SeqView sv = mach_sv(M);
	if(!C.graph) partial_search_item(*C.g, sv, M.L.a0, P.pseudogeneStop != 0, P.anchorStop != 0, P.khits, &M.ws->fh);
	else partial_search_graph_item(*C.g, sv, M.L.a0, P.pseudogeneStop != 0, P.anchorStop != 0, P.khits, P.kseeds, &M.ws->fh, &C.gsl->ie);
}
H2G_MACH_FN void mach_op_gcoords(const AlnCtx& C, Mach& M) {
	Lane& L = M.L;
	h2g_sa_result res;
	if(!C.graph) genome_coords_item(*C.g, L.a0, L.a1, L.a2, L.a3, L.a4 != 0, (h2g_coord*)L.p0, L.a5, &res);
	else genome_coords_graph_item(*C.g, &C.gws->gw, L.a0, L.a1, L.a6, L.a7, (const IEdges*)L.p1, L.a2, L.a3, L.a4 != 0, (h2g_coord*)L.p0, L.a5, &res);
	L.a0 = res.ncoords; L.a1 = res.nsteps;
}
H2G_MACH_FN void mach_op_extend(const AlnCtx& C, Mach& M) {
	Lane& L = M.L;
	uint32_t le = H2G_MAX, re = H2G_MAX;
	al_extend(C, mach_sv(M), (h2g_ghit*)L.p0, L.a0, L.a1, L.a2, &le, &re);
	L.a0 = le; L.a1 = re;
}
H2G_MACH_FN void mach_op_lsearch(const AlnCtx& C, Mach& M) {
	Lane& L = M.L;
	uint32_t extlen = 0, top = L.a4, bot = L.a5;
	bool uniqueStop = L.a3 != 0;
	const uint32_t nelt = al_local_search(C, M.ws, L.a0, mach_sv(M), L.a1, &extlen, &top, &bot, &uniqueStop, L.a2);
	L.a0 = nelt; L.a1 = extlen; L.a2 = top; L.a3 = bot; L.a4 = uniqueStop ? 1u : 0u;
}
H2G_MACH_FN void mach_op_lcoords(const AlnCtx& C, Mach& M) {
	Lane& L = M.L;
	uint32_t n = 0;
	al_local_coords(C, M.ws, L.a0, L.a1, L.a2, L.a3, L.a4, (h2g_coord*)L.p0, L.a5, &n);
	L.a0 = n;
}
H2G_MACH_FN void mach_op_gsearch(const AlnCtx& C, Mach& M) {
	Lane& L = M.L;
	uint32_t extlen = 0, top = L.a4, bot = L.a5;
	bool uniqueStop = L.a3 != 0;
	const uint32_t nelt = al_global_search(C, M.ws, mach_sv(M), L.a1, &extlen, &top, &bot, &uniqueStop);
	L.a0 = nelt; L.a1 = extlen; L.a2 = top; L.a3 = bot; L.a4 = uniqueStop ? 1u : 0u;
}
H2G_MACH_FN void mach_op_combine(const AlnCtx& C, Mach& M) {
	Lane& L = M.L;
	const AlnParams& P = *C.P;
	AlignWS* ws = M.ws;
	// a1 bit 0: join through the database site {a2 left, a3 right, a4 dir} with anchor minima 1, 1; bit 1: minsc in a5 / a6
	const bool has_site = (L.a1 & 1u) != 0;
	h2g_coord site; site.tidx = L.a2; site.toff = L.a3; site.joinedOff = L.a4;
	const int64_t minsc = (L.a1 & 2u) ? (int64_t)(((uint64_t)L.a6 << 32) | (uint64_t)L.a5) : ws->gv.rc_minsc;
	L.a0 = hit_combine(*C.ref, P.sc, mach_sv(M), (h2g_ghit*)L.p0, (const h2g_ghit*)L.p1, minsc, P.minIntronLen, P.no_spliced != 0,
	                   ScVec{C.sc, C.sc_stride}, ScVec{C.sc + (size_t)H2G_COMBINE_MAXLEN * C.sc_stride, C.sc_stride}, C.alts,
	                   has_site ? &site : nullptr, has_site ? 1u : 0u, has_site ? 1u : 0u) ? 1u : 0u;
}
H2G_MACH_FN void mach_op_adjust(const AlnCtx& C, Mach& M) {
	Lane& L = M.L;
	AlignWS* ws = M.ws;
	uint32_t ovf = 0;
	if(C.graph)
		adjust_with_alt(*C.g, *C.ref, *C.alts, mach_sv(M), L.a0, L.a1, L.a2, L.a3, L.a4, ws->ghits, &ws->nghits, AL_MAX_GHITS, &C.gws->awa, &ovf);
	L.a0 = ovf;
}
H2G_MACH_FN void mach_op_adjmember(const AlnCtx& C, Mach& M) {
	M.L.a0 = al_adjust_member(C, mach_sv(M), (h2g_ghit*)M.L.p0, M.ws) ? 1u : 0u;
}
H2G_MACH_FN void mach_op_sw(const AlnCtx& C, Mach& M) {
	Lane& L = M.L;
	const AlnParams& P = *C.P;
	AlignWS* ws = M.ws;
	GoVars& gv = ws->gv;
	h2g_ghit* gh = (h2g_ghit*)L.p0;
	const SeqView sv = mach_sv(M);
	SwParams SP;
	SP.sc = P.sc;
	const uint32_t refoff = gh->toff > gh->rdoff ? gh->toff - gh->rdoff : 0;
	SwOut* o = nullptr;
	sw_align_single(*C.ref, SP, sv, gh->tidx, refoff, ws->m[gv.mw_slot].minsc, &gv.rnd, C.sw, &o);
	if(o->overflow) ws->overflow |= 256;
	L.a0 = 0;
	if(o->found) {
		// res.alres edits: setShape turned them to 5'-end coordinates, `if(!fw) invertEdits()` turns them back to the aligned
		// strand's => exactly the backtrace's own coordinates.  genomeHit.init(fw, 0, rdlen, ...)
		const uint32_t joinedOff = (uint32_t)((int64_t)gh->joinedOff + o->off - (int64_t)gh->toff);
		hit_init(gh, sv.fw, 0, sv.len, gh->tidx, (uint32_t)o->off, joinedOff);
		gh->score = o->score;
		gh->nedits = o->nedits;
		for(uint32_t e = 0; e < o->nedits; e++) gh->edits[e] = o->edits[e];
		if(C.graph && C.alts->n > 0 && gh->nedits > 0) {  // replace_edits_with_alts spliced_aligner.h:282 (re-scores)
			replace_edits_with_alts(*C.alts, gh);
			calculate_score(P.sc, sv, gh);
		}
		L.a0 = 1;
	}
}

// returns false when a long record found no room in the long-edit area (the caller flags the read)
H2G_HD bool mach_copy_rec(h2g_alnres& d, const AlnRec& r, const MachOut& O) {
	d.fw = r.fw; d.tidx = r.tidx; d.toff = r.toff; d.len = r.len; d.trim5 = r.trim5; d.trim3 = r.trim3;
	d.nedits = r.nedits; d.splicescore = r.splicescore; d.score = r.score;
#if H2G_GHIT_EDITS > H2G_MAX_EDITS
	if(r.nedits > H2G_MAX_EDITS) {
		uint32_t at = H2G_MAX;
		if(O.ledits && O.ledits_cursor) {
#if defined(__HIP_DEVICE_COMPILE__)
			at = atomicAdd(O.ledits_cursor, r.nedits);
#else
			at = *O.ledits_cursor; *O.ledits_cursor += r.nedits;
#endif
			if(at + r.nedits > O.ledits_cap) at = H2G_MAX;
		}
		if(at == H2G_MAX) { d.nedits = 0; return false; }
		for(uint32_t e = 0; e < r.nedits; e++) O.ledits[(size_t)at + e] = r.edits[e];
		d.edits[0].pos = at; d.edits[0].chr = d.edits[0].qchr = d.edits[0].type = d.edits[0].pad = 0; d.edits[0].snp = H2G_LONG_EDITS_TAG;
		return true;
	}
#endif
	for(uint32_t e = 0; e < r.nedits; e++) d.edits[e] = r.edits[e];
	return true;
}

H2G_MACH_FN void mach_finish(const AlnCtx& C, Mach& M) {
	const MachOut& O = *M.out;
	const bool paired_input = M.paired_input;
	AlignWS* ws = M.ws;
	GoVars& gv = ws->gv;
	const uint32_t i = M.read;
	if(O.defer_list && ws->overflow) {           // the second pass re-runs this read from scratch with the large workspace
#if defined(__HIP_DEVICE_COMPILE__)
		O.defer_list[atomicAdd(O.defer_count, 1u)] = i;
#else
		O.defer_list[(*O.defer_count)++] = i;
#endif
		M.L.a0 = 0; M.L.a1 = 1;
		return;
	}
	if(!paired_input) {
		ReadOut o;
		Rng rnd; rnd.last = gv.rnd;
		o.nres = ws->m[0].nres; o.overflow = ws->overflow; o.nrank = ws->nrank; o.nsteps = ws->nsteps; o.depth = ws->nframes_max; o.nside = ws->nside;
		o.nselect = al_select(&ws->m[0], *C.P, &rnd, o.select);
		// AlnSetSumm::init aligner_result.cpp:1209: best / second best by AlnScore (score, then hisat2_score)
		int64_t b = INT64_MIN, sb = INT64_MIN, bh = 0, sbh = 0;
		for(uint32_t k = 0; k < ws->m[0].nres; k++) {
			const AlnRec& r = ws->m[0].res[k];
			const int64_t h = hisat2_score(r);
			if(b == INT64_MIN || r.score > b || (r.score == b && h > bh)) { sb = b; sbh = bh; b = r.score; bh = h; }
			else if(sb == INT64_MIN || r.score > sb || (r.score == sb && h > sbh)) { sb = r.score; sbh = h; }
		}
		o.best = b == INT64_MIN ? INT32_MIN : (int32_t)b; o.secbest = sb == INT64_MIN ? INT32_MIN : (int32_t)sb;
		o.best_h2 = (uint32_t)(uint64_t)bh; o.secbest_h2 = (uint32_t)(uint64_t)sbh;
		if(O.aln) for(uint32_t k = 0; k < o.nselect && k < O.aln_slots; k++) if(!mach_copy_rec(O.aln[(size_t)i * O.aln_slots + k], ws->m[0].res[o.select[k]], O)) o.overflow |= 1;
		if(O.rout) O.rout[i] = o;
		M.L.a0 = o.nselect > 0; M.L.a1 = o.overflow != 0;
	} else {
		PairOut o;
		o.nres[0] = ws->m[0].nres; o.nres[1] = ws->m[1].nres; o.npairs = ws->npairs; o.overflow = ws->overflow;
		// more records than the fixed rows hold: the whole pair goes to a block of the overflow area; without room there the pair is
		// flagged, so that no caller indexes past its rows
		uint32_t blk = H2G_MAX;
		if(o.nres[0] > O.pair_slots || o.nres[1] > O.pair_slots) {
			const uint32_t need = o.nres[0] + o.nres[1];
			if(O.ovf && O.ovf_cursor) {
#if defined(__HIP_DEVICE_COMPILE__)
				const uint32_t at = atomicAdd(O.ovf_cursor, need);
#else
				const uint32_t at = *O.ovf_cursor; *O.ovf_cursor += need;
#endif
				if(at + need <= O.ovf_cap) blk = at;
			}
			if(blk == H2G_MAX) o.overflow |= 4;
		}
		o.nrank = ws->nrank; o.nsteps = ws->nsteps; o.depth = ws->nframes_max; o.nside = ws->nside; o.rnd_state = gv.rnd; o.pad = blk == H2G_MAX ? 0u : blk + 1u;
		for(uint32_t k = 0; k < AL_MAX_PAIRS; k++) { o.pair_i[k] = k < ws->npairs ? ws->pair_i[k] : 0; o.pair_j[k] = k < ws->npairs ? ws->pair_j[k] : 0; }
		for(int m = 0; m < 2; m++) {
			if(!O.paln[m]) continue;
			const uint32_t n = o.nres[m] < O.pair_slots ? o.nres[m] : O.pair_slots;
			for(uint32_t k = 0; k < n; k++) if(!mach_copy_rec(O.paln[m][(size_t)i * O.pair_slots + k], ws->m[m].res[k], O)) o.overflow |= 1;
			if(blk != H2G_MAX) for(uint32_t k = 0; k < o.nres[m]; k++) if(!mach_copy_rec(O.ovf[(size_t)blk + (m ? o.nres[0] : 0u) + k], ws->m[m].res[k], O)) o.overflow |= 1;
		}
		if(O.pout) O.pout[i] = o;
		M.L.a0 = o.npairs > 0; M.L.a1 = o.overflow != 0;
	}
}

// one op for this lane; `op` is wave-uniform in the kernel
H2G_HD void mach_exec(const AlnCtx& C, Mach& M, uint32_t op) {
	switch(op) {
	case OP_PSEARCH:   mach_op_psearch(C, M); break;
	case OP_GCOORDS:   mach_op_gcoords(C, M); break;
	case OP_EXTEND:    mach_op_extend(C, M); break;
	case OP_LSEARCH:   mach_op_lsearch(C, M); break;
	case OP_LCOORDS:   mach_op_lcoords(C, M); break;
	case OP_GSEARCH:   mach_op_gsearch(C, M); break;
	case OP_COMBINE:   mach_op_combine(C, M); break;
	case OP_ADJUST:    mach_op_adjust(C, M); break;
	case OP_ADJMEMBER: mach_op_adjmember(C, M); break;
	case OP_SW:        mach_op_sw(C, M); break;
	default: break;
	}
	M.L.op = OP_NONE;
}

// One read / pair to completion on ONE lane (tests/emul; the kernels interleave 64 of these per wavefront)
H2G_HD void mach_run_single(const AlnCtx& C, Mach& M, uint32_t read, bool paired_input, const MachOut& O) {
	M.out = &O; M.paired_input = paired_input;
#if defined(H2G_MEMPROF) && !defined(__HIP_DEVICE_COMPILE__)   // tools/memprof: memory-access profile by phase (development only)
	g_mp_ws = M.ws; g_mp_mach = &M; g_mp_mach_sz = sizeof M; g_mp_phase = 0; g_mp_on = 1; mp_trip();
#define MP_PHASE(p) { g_mp_phase = (p); mp_trip(); }
#else
#define MP_PHASE(p)
#endif
	mach_begin(M, read, paired_input);
	while(M.L.pc != PC_FINISHED || M.L.op != OP_NONE) {
		MP_PHASE(1)
		mach_step(C, M);
#if defined(H2G_MACH_STATS) && !defined(__HIP_DEVICE_COMPILE__)
		extern unsigned long long g_mach_stats[64];
		g_mach_stats[M.L.op]++;
#endif
#if defined(H2G_MACH_TRACE) && !defined(__HIP_DEVICE_COMPILE__)
		if((int)read == H2G_MACH_TRACE) fprintf(stderr, "T %u %u %u %u %u %u %u %u\n", M.L.pc, M.L.op, M.L.a0, M.L.a1, M.L.a2, M.L.a3, M.L.a4, M.L.a5);
#endif
#if defined(H2G_MACH_PCTRACE) && !defined(__HIP_DEVICE_COMPILE__)
		mach_pctrace(0x8000u | M.L.op);          // end of a control phase: the primitive requested (0 = finished)
#endif
		if(M.L.op != OP_NONE) { MP_PHASE(2 + (int)M.L.op) mach_exec(C, M, M.L.op); }
	}
	M.L.pc = PC_IDLE;
#if defined(H2G_MEMPROF) && !defined(__HIP_DEVICE_COMPILE__)
	g_mp_on = 0;
#endif
}

}  // namespace h2g
