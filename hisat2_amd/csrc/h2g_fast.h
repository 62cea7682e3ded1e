// h2g_fast.h — HI_Aligner::go for the DOMINANT traces with the per-read state on chip.
//
// The general machine (h2g_machine.h) keeps a read's state in a ~115 KB workspace in HBM and pays for it: 47 KB of HBM
// traffic per read, control bound by scattered workspace lines.  On the benchmark's reads 85 % of the pairs only ever walk
//   partialSearch* -> getGenomeCoords -> extend -> [localGFMSearch -> getGenomeCoords_local -> extend -> combineWith] -> report
// with one or two genome hits, at most two reported alignments per mate and a recursion of depth <= 3.  This file restates
// exactly that part of go() (same reference lines as h2g_machine.h, cited per state) over a state of ~60 registers and
// ~75 words of LDS per pair (+ 50 rarely touched words in private memory), and gives up ("bail") the moment a read leaves
// it: anything it cannot hold, anything rare (random sub-sampling of repeats, mate rescue, global re-search, soft-clip and
// list overflows, hash matches in the searched / redundant tests).  A bailed read is re-run FROM SCRATCH by the general
// machine, so the only obligation of this file is: a read it completes has exactly the machine's results.  tests/ hold it to
// that on the host (the same source, one lane at a time) over every read of the fuzz sets; on the device the kernel of
// h2g_k_go_fast.hip runs one pair per lane, primitives by wave-level majority vote.
//
// Built for: linear index, --no-spliced-alignment, no --secondary, --bowtie2-dp 0, default pair policy (--fr, -I 0), reads of
// 32..128 bases without N.  Everything else never enters (go_run) or bails at the first state.
#pragma once
#include "h2g_align.h"

namespace h2g {

#define FG_FE     3                     // edits kept per hit (more: bail)
#define FG_HW     (6 + FG_FE)           // words of a stored hit
#define FG_NLONG  4                     // partial hits longer than minK + 2 waiting for getAnchorHits
#define FG_NCO    2                     // coordinates per SA resolution
#define FG_NRES   2                     // reported alignments per mate
#define FG_NSRCH  3                     // hybridSearch_recur roots per mate (hashes)
#define FG_NPAIR  4                     // concordant pairs
#define FG_FRS    5                     // scalar words of a saved frame
#define FG_FRW    (FG_FRS + FG_HW + 3 * FG_NCO)
#define FG_NFRAME 3
// word store of one lane: [0, FW_HOT) lives in LDS, [FW_HOT, FW_TOTAL) in private memory (touched by reads with a mismatch only)
#define FW_LONG   0
#define FW_G0     (FW_LONG + 3 * FG_NLONG)
#define FW_FR0    (FW_G0 + FG_HW)
#define FW_RES    (FW_FR0 + FG_FRW)
#define FW_SRCH   (FW_RES + 2 * FG_NRES * 3)
#define FW_HOT    (FW_SRCH + 2 * FG_NSRCH)
#define FW_G1     FW_HOT
#define FW_T1     (FW_G1 + FG_HW)
#define FW_FR1    (FW_T1 + FG_HW)
#define FW_FR2    (FW_FR1 + FG_FRW)       // the deepest frame: scalars + hit, no coordinate list (it may only report)
#define FW_TOTAL  (FW_FR2 + FG_FRS + FG_HW)
#define FW_COLD   (FW_TOTAL - FW_HOT)

enum : uint32_t { FOP_NONE = 0, FOP_PSEARCH, FOP_GCOORDS, FOP_EXTEND, FOP_LSEARCH, FOP_LCOORDS, FOP_COMBINE, FOP_COUNT };
enum : uint32_t {
	FPC_DONE = 0, FPC_BAIL,
	FPC_GO_INIT, FPC_NB_PICK, FPC_NB_AFTER_PS, FPC_ALIGN, FPC_AFTER_ALIGN, FPC_PAIR_READS, FPC_AFTER_LOOP, FPC_FINISH,
	FPC_GAH_LOOP, FPC_GAH_FULL_AFTER, FPC_GAH_END, FPC_HS_EXT_LOOP, FPC_HS_EXT_AFTER, FPC_HS_LOOP, FPC_HS_AFTER_REC1,
	FPC_RC_ENTRY, FPC_RC_ENTRY_LX, FPC_RC_ENTRY_L2, FPC_RC_ENTRY_L3, FPC_RC_ENTRY_RX, FPC_RC_ENTRY_R2, FPC_RC_ENTRY_R3,
	FPC_L_WHILE, FPC_L_LS_LOOP, FPC_L_LS_AFTER, FPC_L_LS_DONE, FPC_L_LC_AFTER, FPC_L_FOR_RI, FPC_L_RI_B, FPC_L_RI_B2, FPC_L_RI_C, FPC_L_R1,
	FPC_L_AFTER_FOR, FPC_L_AFTER_WHILE, FPC_L_TRIM, FPC_L_R4, FPC_L_EXT, FPC_L_EXT_A, FPC_L_R5,
	FPC_R_WHILE, FPC_R_LS_LOOP, FPC_R_LS_AFTER, FPC_R_LS_DONE, FPC_R_LC_AFTER, FPC_R_FOR_RI, FPC_R_RI_B, FPC_R_RI_C, FPC_R_R1,
	FPC_R_AFTER_FOR, FPC_R_AFTER_WHILE, FPC_R_TRIM, FPC_R_R4, FPC_R_EXT, FPC_R_EXT_A, FPC_R_R5
};

// why a read left the fast path (statistics only)
enum : uint32_t {
	FB_NONE = 0, FB_INPUT, FB_LONGPOOL, FB_SUBSAMPLE, FB_COORDS, FB_NGHITS, FB_EDITS, FB_DEPTH, FB_LOCALHITS, FB_GSEARCH, FB_NRES,
	FB_SEARCHED, FB_REDUNDANT, FB_MATE, FB_NPAIRS, FB_PARTIAL, FB_STRADDLE, FB_OTHER, FB_COUNT
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
	FG_PRIV uint32_t* cold;                     // private memory
	H2G_HD uint32_t ld(uint32_t i) const { return i < FW_HOT ? hot[i * hot_stride] : cold[i - FW_HOT]; }
	H2G_HD void st(uint32_t i, uint32_t v) const { if(i < FW_HOT) hot[i * hot_stride] = v; else cold[i - FW_HOT] = v; }
};

struct FastOut {                         // where a completed read leaves its results (MachOut of h2g_machine.h)
	ReadOut*    rout; h2g_alnres* aln; uint32_t aln_slots;
	PairOut*    pout; h2g_alnres* paln[2]; uint32_t pair_slots;
};

struct FState {                          // registers of one lane
	uint32_t pc, op, bail;
	uint32_t a0, a1, a2, a3, a4, a5;
	uint32_t read, paired, nm;
	uint32_t rl[2], ro[2];                         // length / offset of the mates in their read sets
	uint32_t rnd;
	// ReadBWTHit x 4 (index = rdi * 2 + fwi): hi_aligner.h:216
	uint32_t rb_cur[4], rb_nps[4], rb_nus[4], rb_np[4], rb_sumsq[4];
	uint32_t rb_done, rb_nonempty, found;          // bit sets over the four (read, strand)s
	int32_t  sel_r, sel_f, nb_rdi, nb_fwi;
	uint32_t sv_rdi, sv_fw;
	// per mate: the sink's unpaired lists (summaries in the word store) + HI_Aligner's searched list (hashes)
	uint32_t nres[2], nsearched[2];
	int32_t  bestUnp[2], best2Unp[2], minsc[2];
	// concordant pairs
	uint32_t npairs, pairs, insp_i, insp_j;      // pairs: 4 bits per pair (i | j << 2)
	int32_t  bestPair, best2Pair;
	uint32_t nrank, nside, nsteps, nframes_max;
	// getAnchorHits / hybridSearch
	uint32_t nghits, ghit_done, gh_hi, gh_hj, gh_nco, gh_rdoff, hs_hi, hs_hj, hs_found;
	uint32_t localindexatts, max_localindexatts;
	// hybridSearch_recur
	int32_t  sp, rc_minsc, ret;
	uint32_t rc_ret_pc, pr_ret_pc;
	// the CURRENT frame (saved into the word store across a nested call)
	uint32_t f_hitoff, f_hitlen, f_extoff, f_extlen, f_lidx, f_state, f_count, f_ncoords;
	int32_t  f_ri, f_maxsc, f_prev;
	uint32_t f_success, f_first, f_uselocal, f_unique;
	uint32_t f_top, f_bot, f_nelt, f_noext, f_maxHitLen;    // locals of the local-search loop: dead across calls
};
#define F_SMIN INT32_MIN

struct FCtx {
	const DGfm* g; const DRef* ref; const DLocalSet* ls; const AlnParams* P;
	DReads rd[2];
	const uint32_t* pk[2]; uint32_t pk_stride;     // this lane's packed reads (2-bit words only: reads with an N never enter)
	const char* name[2]; uint32_t namelen[2];
	int64_t* sc; uint32_t sc_stride;               // combineWith temp_scores of this lane
	FastOut O;
};

// ---------------------------------------------------------------------------------------- stored hits
H2G_HD uint32_t fg_frame_base(int sp) { return sp == 0 ? (uint32_t)FW_FR0 : (sp == 1 ? (uint32_t)FW_FR1 : (uint32_t)FW_FR2); }
H2G_HD uint32_t fg_frame_hit(int sp) { return fg_frame_base(sp) + FG_FRS; }
H2G_HD uint32_t fg_frame_co(int sp) { return fg_frame_base(sp) + FG_FRS + FG_HW; }

// hit words: tidx, toff, joinedOff, score, rdoff | len << 8 | trim5 << 16 | trim3 << 24, fw | nedits << 1 | hitcount << 8, edits
H2G_HD void fg_hit_init(const FWords& W, uint32_t hb, bool fw, uint32_t rdoff, uint32_t len, uint32_t tidx, uint32_t toff, uint32_t joff) {
	W.st(hb, tidx); W.st(hb + 1, toff); W.st(hb + 2, joff); W.st(hb + 3, 0);
	W.st(hb + 4, rdoff | (len << 8)); W.st(hb + 5, (fw ? 1u : 0u) | (1u << 8));
}
H2G_HD void fg_hit_copy(const FWords& W, uint32_t dst, uint32_t src) {
	if(dst == src) return;
	const uint32_t w5 = W.ld(src + 5), ne = (w5 >> 1) & 7u;
	for(uint32_t k = 0; k < 6; k++) W.st(dst + k, W.ld(src + k));
	for(uint32_t k = 0; k < ne; k++) W.st(dst + 6 + k, W.ld(src + 6 + k));
}
H2G_HD void fg_hit_load(const FWords& W, uint32_t hb, h2g_ghit* h) {
	h->tidx = W.ld(hb); h->toff = W.ld(hb + 1); h->joinedOff = W.ld(hb + 2); h->score = (int64_t)(int32_t)W.ld(hb + 3);
	const uint32_t w4 = W.ld(hb + 4), w5 = W.ld(hb + 5);
	h->rdoff = w4 & 0xffu; h->len = (w4 >> 8) & 0xffu; h->trim5 = (w4 >> 16) & 0xffu; h->trim3 = w4 >> 24;
	h->fw = w5 & 1u; h->nedits = (w5 >> 1) & 7u; h->read = w5 >> 8;
	h->overflow = 0; h->splicescore = 0;
	for(uint32_t k = 0; k < h->nedits; k++) {
		const uint32_t e = W.ld(hb + 6 + k);
		h2g_edit& d = h->edits[k];
		d.pos = e & 0xffu; d.chr = (uint8_t)(e >> 8); d.qchr = (uint8_t)(e >> 16); d.type = (uint8_t)(e >> 24); d.pad = 0; d.snp = H2G_MAX;
	}
}
// false: the hit does not fit the stored form (the caller bails)
H2G_HD bool fg_hit_store(const FWords& W, uint32_t hb, const h2g_ghit* h) {
	if(h->overflow || h->nedits > FG_FE || h->score < -(1 << 30) || h->score > (1 << 30) || h->read > 0xffffu) return false;
	if(h->rdoff > 255 || h->len > 255 || h->trim5 > 255 || h->trim3 > 255) return false;
	W.st(hb, h->tidx); W.st(hb + 1, h->toff); W.st(hb + 2, h->joinedOff); W.st(hb + 3, (uint32_t)(int32_t)h->score);
	W.st(hb + 4, h->rdoff | (h->len << 8) | (h->trim5 << 16) | (h->trim3 << 24));
	W.st(hb + 5, (h->fw ? 1u : 0u) | (h->nedits << 1) | (h->read << 8));
	for(uint32_t k = 0; k < h->nedits; k++) {
		const h2g_edit& e = h->edits[k];
		if(e.pos > 255 || e.snp != H2G_MAX || e.pad != 0) return false;
		W.st(hb + 6 + k, e.pos | ((uint32_t)e.chr << 8) | ((uint32_t)e.qchr << 16) | ((uint32_t)e.type << 24));
	}
	return true;
}
// a hash of exactly what GenomeHit::operator== compares (hit_equal, hi_aligner.h:1156): equal hits => equal hashes
H2G_HD uint32_t fg_hit_hash(const h2g_ghit* h) {
	uint32_t x = 0x9e3779b9u;
#define FG_MIX(V) do { x ^= (uint32_t)(V); x *= 0x85ebca6bu; x ^= x >> 13; } while(0)
	FG_MIX(h->fw); FG_MIX(h->rdoff); FG_MIX(h->len); FG_MIX(h->tidx); FG_MIX(h->toff); FG_MIX(h->trim5); FG_MIX(h->trim3); FG_MIX(h->nedits);
	for(uint32_t i = 0; i < h->nedits; i++) {
		const h2g_edit& e = h->edits[i];
		if(e.type == H2G_EDIT_READ_GAP || e.type == H2G_EDIT_REF_GAP) FG_MIX(e.type);
		else FG_MIX(e.pos | ((uint32_t)e.chr << 8) | ((uint32_t)e.qchr << 16) | ((uint32_t)e.type << 24));
	}
#undef FG_MIX
	return x;
}

// small per-strand arrays by explicit selects (registers, no private-memory indexing)
#define FG_GET4(A, I) ((I) == 0 ? (A)[0] : (I) == 1 ? (A)[1] : (I) == 2 ? (A)[2] : (A)[3])
#define FG_SET4(A, I, V) do { const uint32_t v_ = (V); if((I) == 0) (A)[0] = v_; else if((I) == 1) (A)[1] = v_; else if((I) == 2) (A)[2] = v_; else (A)[3] = v_; } while(0)
#define FG_GET2(A, I) ((I) == 0 ? (A)[0] : (A)[1])

H2G_HD SeqView fg_view(const FCtx& C, const FState& S, uint32_t set, bool fw) {
	const DReads& r = C.rd[set];
	const uint32_t ro = FG_GET2(S.ro, set);
	SeqView s;
	s.fwc = r.codes + ro; s.q = r.quals ? r.quals + ro : nullptr; s.len = FG_GET2(S.rl, set); s.fw = fw;
	s.pk = C.pk[set]; s.pk_stride = C.pk_stride; s.pk_nomask = true;
	return s;
}
H2G_HD SeqView fg_sv(const FCtx& C, const FState& S) { return fg_view(C, S, S.paired ? S.sv_rdi : 0u, S.sv_fw != 0); }

// the long partial hits of strand x leave the pool (the strand is done)
H2G_HD void fg_pool_free(const FWords& W, uint32_t x) {
	for(uint32_t k = 0; k < FG_NLONG; k++) { const uint32_t m = W.ld(FW_LONG + 3 * k + 2); if(m && ((m >> 20) & 3u) == x) W.st(FW_LONG + 3 * k + 2, 0); }
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

// genRandSeed (pat.h:55-91, gen_rand_seed of h2g_align.h) of a read without N from its packed words: the 2-bit codes of 16 bases
// XOR into the seed exactly as they lie in a packed word
H2G_HD uint32_t fg_rand_seed(const FCtx& C, const FState& S, uint32_t set) {
	uint32_t rseed = (0u + 101u) * 59u * 61u * 67u * 71u * 73u * 79u * 83u;
	const uint32_t len = FG_GET2(S.rl, set);
	for(uint32_t w = 0; w < H2G_PK_WORDS; w++) rseed ^= C.pk[set][w * C.pk_stride];
	const DReads& r = C.rd[set];
	if(r.quals) {
		const char* q = r.quals + FG_GET2(S.ro, set);
		for(uint32_t i = 0; i < len; i++) rseed ^= ((uint32_t)q[i] << ((i & 3) << 3));
	} else {
		for(uint32_t b = 0; b < 4; b++) if(((len + 3 - b) >> 2) & 1u) rseed ^= (uint32_t)'I' << (b << 3);
	}
	const char* name = C.name[set];
	for(uint32_t i = 0; i < C.namelen[set]; i++) {
		const int p = name[i];
		if(p == '/') break;
		rseed ^= ((uint32_t)p << ((i & 3) << 3));
	}
	return rseed;
}

// The worker-loop prelude (mach_begin; hisat2.cpp:3380-3530).  `ok0/ok1`: the mates were packed without an N and have 32..128 bases.
H2G_HD void fast_begin(const FCtx& C, FState& S, uint32_t read, bool paired, bool packed_ok) {
	S.read = read; S.op = FOP_NONE; S.bail = FB_NONE; S.paired = paired ? 1u : 0u; S.nm = paired ? 2u : 1u;
	for(int k = 0; k < 2; k++) { const DReads& r = C.rd[paired ? k : 0]; S.ro[k] = r.offs[read]; S.rl[k] = r.offs[read + 1] - S.ro[k]; }
	if(!packed_ok || S.rl[0] < 32 || S.rl[0] > 128 || (paired && (S.rl[1] < 32 || S.rl[1] > 128))) { S.pc = FPC_BAIL; S.bail = FB_INPUT; return; }
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
#define F_MINSC_LIVE(MV) do { const int32_t b_ = FG_GET2(S.bestUnp, S.sv_rdi); if(b_ > (MV)) (MV) = b_; } while(0)

H2G_HD void fg_frame_save(const FWords& W, const FState& S) {
	const uint32_t b = fg_frame_base(S.sp);
	W.st(b, S.f_hitoff | (S.f_hitlen << 8) | (S.f_extoff << 16) | (S.f_extlen << 24));
	W.st(b + 1, (uint32_t)S.f_maxsc); W.st(b + 2, (uint32_t)S.f_prev); W.st(b + 3, S.f_lidx);
	W.st(b + 4, S.f_state | (S.f_count << 8) | (S.f_ncoords << 10) | ((uint32_t)(S.f_ri + 1) << 12) | (S.f_success << 15) | (S.f_first << 16) |
	            (S.f_uselocal << 17) | (S.f_unique << 18));
}
H2G_HD void fg_frame_restore(const FWords& W, FState& S) {
	const uint32_t b = fg_frame_base(S.sp);
	const uint32_t w0 = W.ld(b), w4 = W.ld(b + 4);
	S.f_hitoff = w0 & 0xffu; S.f_hitlen = (w0 >> 8) & 0xffu; S.f_extoff = (w0 >> 16) & 0xffu; S.f_extlen = w0 >> 24;
	S.f_maxsc = (int32_t)W.ld(b + 1); S.f_prev = (int32_t)W.ld(b + 2); S.f_lidx = W.ld(b + 3);
	S.f_state = w4 & 0xffu; S.f_count = (w4 >> 8) & 3u; S.f_ncoords = (w4 >> 10) & 3u; S.f_ri = (int32_t)((w4 >> 12) & 7u) - 1;
	S.f_success = (w4 >> 15) & 1u; S.f_first = (w4 >> 16) & 1u; S.f_uselocal = (w4 >> 17) & 1u; S.f_unique = (w4 >> 18) & 1u;
}

// reportHit + AlnSinkWrap::report (al_report of h2g_align.h) for a full-length hit; the record goes straight to its output slot
H2G_HD void fg_write_rec(h2g_alnres& d, const h2g_ghit* hit, uint32_t rdlen) {
	d.fw = hit->fw; d.tidx = hit->tidx; d.toff = hit->toff; d.len = hit->len; d.trim5 = hit->trim5; d.trim3 = hit->trim3;
	d.nedits = hit->nedits; d.splicescore = hit->splicescore; d.score = hit->score;
	const uint32_t trim5p = hit->fw ? hit->trim5 : hit->trim3;
	for(uint32_t k = 0; k < hit->nedits; k++) {
		h2g_edit e;
		if(hit->fw) { e = hit->edits[k]; e.pos += hit->trim5; }
		else e = inverted_edit(hit, k, rdlen, hit->trim5);
		e.pos -= trim5p;
		d.edits[k] = e;
	}
}
H2G_HD uint32_t fg_ref_extent(const h2g_ghit* h) {
	uint32_t ext = h->len;
	for(uint32_t k = 0; k < h->nedits; k++) {
		if(h->edits[k].type == H2G_EDIT_READ_GAP) ext++;
		else if(h->edits[k].type == H2G_EDIT_REF_GAP) ext--;
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
		for(int r = 0; r < 2; r++) {
			S.nres[r] = 0; S.nsearched[r] = 0; S.bestUnp[r] = F_SMIN; S.best2Unp[r] = F_SMIN; S.minsc[r] = INT32_MAX;
			if((uint32_t)r < S.nm) {
				const int64_t m = min_score_for(P, S.rl[r]);
				if(m < -(1 << 24)) F_BAIL(FB_INPUT);
				S.minsc[r] = (int32_t)m;
			}
		}
		for(int k = 0; k < 4; k++) { S.rb_cur[k] = 0; S.rb_nps[k] = 0; S.rb_nus[k] = 0; S.rb_np[k] = 0; S.rb_sumsq[k] = 0; }
		S.rb_done = 0; S.rb_nonempty = 0;
		S.found = S.paired ? 15u : 3u;                           // found[0][0], [0][1], [1][0], [1][1]
		for(uint32_t k = 0; k < FG_NLONG; k++) W.st(FW_LONG + 3 * k + 2, 0);
		F_GOTO(FPC_NB_PICK);
	}
	case FPC_NB_PICK: {                                   // one iteration of nextBWT's loop (:4644-4760)
		int rdi = -1, fwi = -1;
		int64_t maxScore = INT64_MIN;
		for(uint32_t r = 0; r < S.nm; r++) for(int k = 0; k < 2; k++) {
			const uint32_t x = r * 2 + (uint32_t)k;
			if((S.rb_done >> x) & 1u) continue;
			const uint32_t act = FG_GET4(S.rb_nps, x) - FG_GET4(S.rb_nus, x);
			int64_t cs = (int64_t)FG_GET4(S.rb_sumsq, x) - (int64_t)act * minK * minK - ((int64_t)1 << (act << 1));   // ReadBWTHit::searchScore :320
			if(FG_GET4(S.rb_cur, x) == 0) cs = INT64_MAX;
			if(cs > maxScore) { maxScore = cs; rdi = (int)r; fwi = k; }
		}
		if(rdi < 0) F_GOTO(FPC_AFTER_LOOP);
		const uint32_t x = (uint32_t)rdi * 2 + (uint32_t)fwi, xr = (uint32_t)rdi * 2 + (uint32_t)(1 - fwi);
		{
			const uint32_t numSearched = FG_GET4(S.rb_nps, x) - FG_GET4(S.rb_nus, x);
			const int32_t bestScore = FG_GET2(S.bestUnp, rdi), msc = FG_GET2(S.minsc, rdi);
			if(bestScore >= msc) {
				const uint32_t maxmm = (uint32_t)((-(int64_t)bestScore + sc.mmpMax - 1) / sc.mmpMax);
				if(numSearched > maxmm + 1) {
					S.rb_done |= 1u << x; fg_pool_free(W, x);
					if(S.paired) {
						const int32_t ob = FG_GET2(S.bestUnp, 1 - rdi), om = FG_GET2(S.minsc, 1 - rdi);
						if(ob >= om && S.npairs > 0) F_GOTO(FPC_AFTER_LOOP); else F_GOTO(FPC_NB_PICK);
					} else F_GOTO(FPC_AFTER_LOOP);
				}
			}
			if(((S.rb_done >> xr) & 1u) && bestScore < msc) {
				const uint32_t rcs = FG_GET4(S.rb_nps, xr) - FG_GET4(S.rb_nus, xr);
				if(numSearched > rcs + (P.anchorStop ? 1u : 0u)) { S.rb_done |= 1u << x; fg_pool_free(W, x); F_GOTO(FPC_AFTER_LOOP); }
			}
		}
		S.nb_rdi = rdi; S.nb_fwi = fwi;
		S.sv_rdi = (uint32_t)rdi; S.sv_fw = fwi == 0;
		S.a0 = FG_GET4(S.rb_cur, x);
		F_OP(FOP_PSEARCH, FPC_NB_AFTER_PS);
	}
	case FPC_NB_AFTER_PS: {
		// a0 top, a1 bot, a2 len | hit_type << 8 | done << 16 | anchorStop << 17 | numUniqueSearch << 18, a3 cur, a4 nrank | nside << 16, a5 bwoff
		const int rdi = S.nb_rdi, fwi = S.nb_fwi;
		const uint32_t x = (uint32_t)rdi * 2 + (uint32_t)fwi;
		const uint32_t top = S.a0, bot = S.a1, len = S.a2 & 0xffu, type = (S.a2 >> 8) & 0xffu, done = (S.a2 >> 16) & 1u, anchor = (S.a2 >> 17) & 1u, nus = S.a2 >> 18;
		S.nrank += S.a4 & 0xffffu; S.nside += S.a4 >> 16;
		FG_SET4(S.rb_nps, x, FG_GET4(S.rb_nps, x) + 1); FG_SET4(S.rb_nus, x, FG_GET4(S.rb_nus, x) + nus); FG_SET4(S.rb_cur, x, S.a3);
		const uint32_t np = FG_GET4(S.rb_np, x);
		if(np >= 16) F_BAIL(FB_PARTIAL);                    // AL_MAX_PARTIAL of the default workspace
		FG_SET4(S.rb_sumsq, x, FG_GET4(S.rb_sumsq, x) + len * len);
		if(bot > top && bot != H2G_MAX) {
			S.rb_nonempty |= 1u << x;
			if(len > minK + 2) {                            // getAnchorHits looks at these only (:5033)
				uint32_t k = 0;
				for(; k < FG_NLONG; k++) if(W.ld(FW_LONG + 3 * k + 2) == 0) break;
				if(k >= FG_NLONG) F_BAIL(FB_LONGPOOL);
				W.st(FW_LONG + 3 * k, top); W.st(FW_LONG + 3 * k + 1, bot);
				W.st(FW_LONG + 3 * k + 2, 0x80000000u | S.a5 | (len << 8) | (type << 16) | (x << 20) | (np << 24));   // bwoff, len, type, strand, index
			}
		}
		FG_SET4(S.rb_np, x, np + 1);
		if(done) { S.rb_done |= 1u << x; S.sel_r = rdi; S.sel_f = fwi; F_GOTO(FPC_ALIGN); }
		{ const uint32_t cur = FG_GET4(S.rb_cur, x); if(cur + 1 < FG_GET2(S.rl, rdi)) FG_SET4(S.rb_cur, x, cur + 1); }   // !pseudogeneStop (never set without spliced alignment)
		if(anchor) { S.rb_done |= 1u << x; S.sel_r = rdi; S.sel_f = fwi; F_GOTO(FPC_ALIGN); }
		F_GOTO(FPC_NB_PICK);
	}
	// ======================================================================== align() :5484-5573
	case FPC_ALIGN: {
		S.sv_rdi = (uint32_t)S.sel_r; S.sv_fw = S.sel_f == 0;
		const uint32_t x = (uint32_t)S.sel_r * 2 + (uint32_t)S.sel_f;
		if(!((S.rb_nonempty >> x) & 1u)) { S.hs_found = 0; F_GOTO(FPC_AFTER_ALIGN); }
		int32_t bestScore = FG_GET2(S.bestUnp, S.sel_r);
		const int32_t msc = FG_GET2(S.minsc, S.sel_r);
		if(bestScore < msc) bestScore = msc;
		const uint32_t maxmm = (uint32_t)((-(int64_t)bestScore + sc.mmpMax - 1) / sc.mmpMax);
		const uint32_t nact = FG_GET4(S.rb_nps, x) - FG_GET4(S.rb_nus, x);
		if(nact > maxmm + 1) { S.hs_found = 1; F_GOTO(FPC_AFTER_ALIGN); }
		S.nghits = 0; S.gh_hi = 0;
		F_GOTO(FPC_GAH_LOOP);
	}
	case FPC_AFTER_ALIGN: {
		// the strand is done for good (align() runs once per strand): its long partial hits leave the pool
		{
			const uint32_t x = (uint32_t)S.sel_r * 2 + (uint32_t)S.sel_f;
			fg_pool_free(W, x);
			if(S.hs_found) S.found |= 1u << x; else S.found &= ~(1u << x);
		}
		if(S.found == 0) F_GOTO(FPC_AFTER_LOOP);
		if(S.paired) { S.pr_ret_pc = FPC_NB_PICK; F_GOTO(FPC_PAIR_READS); }
		F_GOTO(FPC_NB_PICK);
	}
	case FPC_PAIR_READS: {                                // pairReads hi_aligner.h:5948-6055 (al_pair_reads) over the summaries
		const uint32_t start_i = S.insp_i, start_j = S.insp_j;
		S.insp_i = S.nres[0]; S.insp_j = S.nres[1];
		for(uint32_t i = 0; i < S.nres[0]; i++) {
			for(uint32_t j = (i >= start_i ? 0 : start_j); j < S.nres[1]; j++) {
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
					if(S.bestUnp[0] >= S.minsc[0] && S.bestUnp[1] >= S.minsc[1]) {
						const int64_t tmp = (int64_t)((double)((int64_t)S.bestUnp[0] + S.bestUnp[1]) - (double)(S.rl[0] + S.rl[1]) * 0.03 * (double)sc.mmpMax);
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
		if(S.paired && S.npairs == 0 && (S.bestUnp[0] >= S.minsc[0] || S.bestUnp[1] >= S.minsc[1])) F_BAIL(FB_MATE);
		F_GOTO(FPC_FINISH);
	}
	// ======================================================================== getAnchorHits :5007-5193
	case FPC_GAH_LOOP: {
		const uint32_t x = (uint32_t)S.sel_r * 2 + (uint32_t)S.sel_f;
		const uint32_t offsetSize = FG_GET4(S.rb_np, x);
		if(S.gh_hi >= offsetSize) F_GOTO(FPC_GAH_END);
		// candidates in index order: pool entries of this strand that were not resolved yet (bit 30 = has coordinates)
		uint32_t hj = FG_NLONG, mj = 0, tj_top = 0, tj_bot = 0;
		for(uint32_t idx = 0; idx < 16; idx++) {                  // by the hit's index in the strand's list
			uint32_t k = 0, m = 0;
			for(; k < FG_NLONG; k++) { m = W.ld(FW_LONG + 3 * k + 2); if(m && ((m >> 20) & 3u) == x && ((m >> 24) & 31u) == idx) break; }
			if(k >= FG_NLONG) continue;
			if(m & 0x40000000u) continue;                        // ncoords > 0
			const uint32_t top = W.ld(FW_LONG + 3 * k), bot = W.ld(FW_LONG + 3 * k + 1);
			if(hj == FG_NLONG) { hj = k; mj = m; tj_top = top; tj_bot = bot; continue; }
			const uint32_t tj = (mj >> 16) & 15u, tk = (m >> 16) & 15u, lj = (mj >> 8) & 0xffu, lk = (m >> 8) & 0xffu;
			const uint32_t sj = tj_bot - tj_top, sk = bot - top;
			const bool better = tj == tk ? (sj > sk || (sj == sk && lj < lk)) : (tk > tj);
			if(better) { hj = k; mj = m; tj_top = top; tj_bot = bot; }
		}
		if(hj == FG_NLONG) F_GOTO(FPC_GAH_END);
		const uint32_t remained = maxsz - S.nghits;
		if(remained == 0) F_GOTO(FPC_GAH_END);
		const uint32_t expected = tj_bot - tj_top, len = (mj >> 8) & 0xffu, bwoff = mj & 0xffu;
		if(expected > remained) F_BAIL(FB_SUBSAMPLE);       // the random sub-sample of a repeat's rows (:5096-5136)
		if(expected > FG_NCO) F_BAIL(FB_COORDS);
		S.gh_hj = hj; S.gh_nco = 0;
		S.gh_rdoff = FG_GET2(S.rl, S.sel_r) - bwoff - len;
		S.a0 = tj_top; S.a1 = tj_bot; S.a2 = expected; S.a3 = len; S.a4 = 0; S.a5 = fg_frame_co(0);
		F_OP(FOP_GCOORDS, FPC_GAH_FULL_AFTER);
	}
	case FPC_GAH_FULL_AFTER: {
		const uint32_t nco = S.a0;
		S.nsteps += S.a1;
		if(nco == 0) F_BAIL(FB_COORDS);                     // joinedToTextOff failed: the reference retries the same hit (:5140)
		const uint32_t mslot = FW_LONG + 3 * S.gh_hj + 2;
		const uint32_t m = W.ld(mslot);
		W.st(mslot, m | 0x40000000u);                       // ph.ncoords = nco
		const uint32_t len = (m >> 8) & 0xffu, type = (m >> 16) & 15u;
		const uint32_t gsize = S.nghits;                    // gsize + nco <= maxsz: no shuffle (:5147)
		const uint32_t rl = FG_GET2(S.rl, S.sel_r);
		for(uint32_t k = 0; k < nco; k++) {
			const uint32_t cb = fg_frame_co(0) + 3 * k;
			const uint32_t tidx = W.ld(cb), toff = W.ld(cb + 1), joff = W.ld(cb + 2);
			if(tidx == H2G_MAX) F_BAIL(FB_STRADDLE);
			bool overlapped = false;
			for(uint32_t l = 0; l < gsize; l++) {
				const uint32_t gb = l == 0 ? (uint32_t)FW_G0 : (uint32_t)FW_G1;
				const uint32_t w5 = W.ld(gb + 5);
				if(W.ld(gb) != tidx || ((w5 & 1u) != 0) != (S.sv_fw != 0)) continue;
				const uint32_t g_rdoff = W.ld(gb + 4) & 0xffu;
				const uint32_t hitoff = W.ld(gb + 1) + rl - g_rdoff, hitoff2 = toff + rl - S.gh_rdoff;
				if(hitoff == hitoff2) { overlapped = true; W.st(gb + 5, w5 + (1u << 8)); break; }   // _hitcount++
			}
			if(!overlapped) {
				if(S.nghits >= 2) F_BAIL(FB_NGHITS);
				fg_hit_init(W, S.nghits == 0 ? (uint32_t)FW_G0 : (uint32_t)FW_G1, S.sv_fw != 0, S.gh_rdoff, len, tidx, toff, joff);
				S.nghits++;
			}
			if(type == H2G_CANDIDATE_HIT && S.nghits >= maxsz) break;
		}
		if(type == H2G_CANDIDATE_HIT && S.nghits >= maxsz) F_GOTO(FPC_GAH_END);
		S.gh_hi++;
		F_GOTO(FPC_GAH_LOOP);
	}
	case FPC_GAH_END: {
		const uint32_t numHits = S.nghits;
		if(numHits == 0) { S.hs_found = 0; F_GOTO(FPC_AFTER_ALIGN); }
		const uint64_t add = (uint64_t)((-(int64_t)FG_GET2(S.minsc, S.sel_r)) / sc.mmpMax) * numHits;
		S.max_localindexatts = S.localindexatts + (uint32_t)(add > 10 ? add : 10);
		S.hs_hi = 0;
		F_GOTO(FPC_HS_EXT_LOOP);
	}
	// ======================================================================== hybridSearch spliced_aligner.h:112-322
	case FPC_HS_EXT_LOOP: {
		if(S.hs_hi >= S.nghits) { S.hs_hi = 0; F_GOTO(FPC_HS_LOOP); }
		S.a0 = 0; S.a1 = H2G_MAX; S.a2 = H2G_MAX; S.a3 = S.hs_hi == 0 ? (uint32_t)FW_G0 : (uint32_t)FW_G1;
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
			const uint32_t bj = hj == 0 ? (uint32_t)FW_G0 : (uint32_t)FW_G1, bk = hk == 0 ? (uint32_t)FW_G0 : (uint32_t)FW_G1;
			const uint32_t ar = W.ld(bj + 5) >> 8, al = (W.ld(bj + 4) >> 8) & 0xffu, br = W.ld(bk + 5) >> 8, bl = (W.ld(bk + 4) >> 8) & 0xffu;
			if(br > ar || (br == ar && bl > al)) hj = hk;
		}
		S.hs_hj = hj;
		const uint32_t gb = hj == 0 ? (uint32_t)FW_G0 : (uint32_t)FW_G1;
		// hybridSearch_recur(root) (RC_START): frame 0
		const uint32_t w4 = W.ld(gb + 4);
		fg_hit_copy(W, fg_frame_hit(0), gb);
		S.sp = 0; S.f_hitoff = w4 & 0xffu; S.f_hitlen = (w4 >> 8) & 0xffu;
		S.rc_minsc = FG_GET2(S.minsc, S.sel_r); S.rc_ret_pc = FPC_HS_AFTER_REC1; S.ret = F_SMIN;
		F_GOTO(FPC_RC_ENTRY);
	}
	case FPC_HS_AFTER_REC1: { S.ghit_done |= 1u << S.hs_hj; S.hs_hi++; F_GOTO(FPC_HS_LOOP); }     // (no SwAligner pass: bowtie2_dp == 0)
	// ======================================================================== hybridSearch_recur spliced_aligner.h:331-2052
	case FPC_RC_ENTRY: {
		h2g_ghit hit;
		fg_hit_load(W, fg_frame_hit(S.sp), &hit);
		const uint32_t hitoff = S.f_hitoff, hitlen = S.f_hitlen, rdlen = FG_GET2(S.rl, S.sv_rdi);
		const int32_t minsc = S.rc_minsc;
		S.f_maxsc = F_SMIN;
		if(hit.score < (int64_t)minsc) F_RC_RET(S.f_maxsc);
		if(hitoff == hit.rdoff - hit.trim5 && hitlen == hit.len + hit.trim5 + hit.trim3) {
			const uint32_t hsh = fg_hit_hash(&hit);
			const uint32_t sb = FW_SRCH + S.sv_rdi * FG_NSRCH, ns = FG_GET2(S.nsearched, S.sv_rdi);
			for(uint32_t i = 0; i < ns; i++) if(W.ld(sb + i) == hsh) F_BAIL(FB_SEARCHED);   // isSearched (or a collision): not ours to decide
			if(ns >= FG_NSRCH) F_BAIL(FB_SEARCHED);
			W.st(sb + ns, hsh);
			if(S.sv_rdi == 0) S.nsearched[0] = ns + 1; else S.nsearched[1] = ns + 1;
		}
		if(hitoff == 0 && hitlen == rdlen) {
			// redundant() :6311 over the summaries: same locus, strand and edit count => the general machine compares the edits
			const uint32_t nr = FG_GET2(S.nres, S.sv_rdi);
			for(uint32_t i = 0; i < nr; i++) {
				const uint32_t rb = fg_res_base(S.sv_rdi, i), m = W.ld(rb + 2);
				if(W.ld(rb) == hit.tidx && W.ld(rb + 1) == hit.toff && (m & 1u) == (hit.fw ? 1u : 0u) && ((m >> 1) & 7u) == hit.nedits) F_BAIL(FB_REDUNDANT);
			}
			// reportHit :6064 (al_report)
			if(!(hit.rdoff - hit.trim5 > 0 || hit.len + hit.trim5 + hit.trim3 < rdlen) && hit.score >= (int64_t)minsc) {
				if(nr >= FG_NRES) F_BAIL(FB_NRES);
				const uint32_t ext = fg_ref_extent(&hit);
				if(hit.score < -32000 || hit.score > 32000 || ext > 0xfffu) F_BAIL(FB_OTHER);
				if(S.paired) {
					if(nr >= C.O.pair_slots) F_BAIL(FB_NRES);
					fg_write_rec(C.O.paln[S.sv_rdi][(size_t)S.read * C.O.pair_slots + nr], &hit, rdlen);
				} else {
					if(nr >= C.O.aln_slots) F_BAIL(FB_NRES);
					fg_write_rec(C.O.aln[(size_t)S.read * C.O.aln_slots + nr], &hit, rdlen);
				}
				const uint32_t rb = fg_res_base(S.sv_rdi, nr);
				W.st(rb, hit.tidx); W.st(rb + 1, hit.toff);
				W.st(rb + 2, (hit.fw ? 1u : 0u) | (hit.nedits << 1) | (ext << 4) | ((uint32_t)((int32_t)hit.score + 32768) << 16));
				const int32_t s = (int32_t)hit.score;
				if(S.sv_rdi == 0) {
					S.nres[0] = nr + 1;
					if(S.bestUnp[0] == F_SMIN || s > S.bestUnp[0]) { S.best2Unp[0] = S.bestUnp[0]; S.bestUnp[0] = s; } else if(S.best2Unp[0] == F_SMIN || s > S.best2Unp[0]) S.best2Unp[0] = s;
				} else {
					S.nres[1] = nr + 1;
					if(S.bestUnp[1] == F_SMIN || s > S.bestUnp[1]) { S.best2Unp[1] = S.bestUnp[1]; S.bestUnp[1] = s; } else if(S.best2Unp[1] == F_SMIN || s > S.best2Unp[1]) S.best2Unp[1] = s;
				}
			}
			// (the frame's maximum follows the hit whether or not it was reported: spliced_aligner.h:676)
			if(hit.score < -(1 << 30)) F_BAIL(FB_OTHER);
			if(S.f_maxsc == F_SMIN || (int32_t)hit.score > S.f_maxsc) S.f_maxsc = (int32_t)hit.score;
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
		const uint32_t rdlen = FG_GET2(S.rl, S.sv_rdi);
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
		if((w4 & 0xffu) + ((w4 >> 8) & 0xffu) == FG_GET2(S.rl, S.sv_rdi)) S.f_uselocal = 0;
		F_GOTO(FPC_RC_ENTRY_R3);
	}
	case FPC_RC_ENTRY_L3:
	case FPC_RC_ENTRY_R3: {
		const uint32_t hb = fg_frame_hit(S.sp);
		S.f_lidx = local_index_of(*C.ls, W.ld(hb), W.ld(hb + 1));
		S.f_success = 0; S.f_first = 1; S.f_count = 0; S.f_prev = (int32_t)W.ld(hb + 3); S.f_ncoords = 0; S.f_ri = 0;
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
		if(!(S.f_extoff < FG_GET2(S.rl, S.sv_rdi))) F_GOTO(FPC_L_LS_DONE);
		S.f_extlen = 0; S.f_unique = 1;
		S.localindexatts++;
		if(C.ls->desc[S.f_lidx].len == 0) { S.a0 = 0; S.a1 = 0; S.a2 = S.f_top; S.a3 = S.f_bot; S.a4 = 1; F_GOTO(FPC_L_LS_AFTER); }
		S.a0 = S.f_lidx; S.a1 = S.f_extoff; S.a2 = 0xffffu; S.a3 = 1; S.a4 = S.f_top; S.a5 = S.f_bot;
		F_OP(FOP_LSEARCH, FPC_L_LS_AFTER);
	}
	case FPC_L_LS_AFTER: {
		S.f_nelt = S.a0; S.f_extlen = S.a1; S.f_top = S.a2; S.f_bot = S.a3; S.f_unique = S.a4 & 1u;
		if(S.f_extoff + 1 - S.f_extlen >= S.f_hitoff) { S.f_noext = 1; F_GOTO(FPC_L_LS_DONE); }
		if(S.f_nelt <= 5) F_GOTO(FPC_L_LS_DONE);
		S.f_extoff++;
		F_GOTO(FPC_L_LS_LOOP);
	}
	case FPC_L_LS_DONE: {
		S.f_ncoords = 0; S.f_ri = -1;
		if(S.f_nelt > 0 && S.f_nelt <= 5 && S.f_extlen >= P.minAnchorLen && !S.f_noext) {
			if(S.f_bot - S.f_top > FG_NCO) F_BAIL(FB_COORDS);
			S.a0 = S.f_lidx; S.a1 = S.f_top; S.a2 = S.f_bot; S.a3 = S.f_extoff + 1 - S.f_extlen; S.a4 = S.f_extlen; S.a5 = fg_frame_co(S.sp);
			F_OP(FOP_LCOORDS, FPC_L_LC_AFTER);
		}
		F_GOTO(FPC_L_FOR_RI);
	}
	case FPC_L_LC_AFTER:
	case FPC_R_LC_AFTER: {
		S.f_ncoords = S.a0;
		if(S.f_ncoords == 2) {                              // sort_coords
			const uint32_t cb = fg_frame_co(S.sp);
			const uint32_t t0 = W.ld(cb), o0 = W.ld(cb + 1), j0 = W.ld(cb + 2), t1 = W.ld(cb + 3), o1 = W.ld(cb + 4), j1 = W.ld(cb + 5);
			if(t0 > t1 || (t0 == t1 && o0 > o1)) { W.st(cb, t1); W.st(cb + 1, o1); W.st(cb + 2, j1); W.st(cb + 3, t0); W.st(cb + 4, o0); W.st(cb + 5, j0); }
		}
		if(S.pc == FPC_L_LC_AFTER) { S.f_ri = (int32_t)S.f_ncoords - 1; F_GOTO(FPC_L_FOR_RI); }
		F_GOTO(FPC_R_FOR_RI);
	}
	case FPC_L_FOR_RI: {
		if(S.f_ri < 0) F_GOTO(FPC_L_AFTER_FOR);
		const uint32_t cb = fg_frame_co(S.sp) + 3 * (uint32_t)S.f_ri, hb = fg_frame_hit(S.sp);
		const bool fw = (W.ld(hb + 5) & 1u) != 0;
		fg_hit_init(W, FW_T1, fw, S.f_extoff + 1 - S.f_extlen, S.f_extlen, W.ld(cb), W.ld(cb + 1), W.ld(cb + 2));
		h2g_ghit t, fh;
		fg_hit_load(W, FW_T1, &t); fg_hit_load(W, hb, &fh);
		if(!hit_compatible(&t, &fh, P.maxIntronLen, true)) {
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
			F_BAIL(FB_LOCALHITS);                           // _local_genomeHits: kept for later by the general machine
		}
		F_GOTO(FPC_L_FOR_RI);
	}
	case FPC_L_R1: { if(S.ret > S.f_maxsc) S.f_maxsc = S.ret; F_GOTO(FPC_L_FOR_RI); }
	case FPC_L_AFTER_FOR: {
		if(S.f_maxsc != F_SMIN && S.f_maxsc >= S.f_prev - sc.mmpMax) S.f_success = 1;
		F_GOTO(FPC_L_WHILE);                                // (the loop over _local_genomeHits is empty: nlocal == 0)
	}
	case FPC_L_AFTER_WHILE: {
		if(S.f_success) F_RC_RET(S.f_maxsc);
		S.f_ncoords = 0; S.f_ri = -1;
		if(S.f_hitoff > minK && S.localindexatts < S.max_localindexatts) F_BAIL(FB_GSEARCH);   // global search for long introns (:1085)
		F_GOTO(FPC_L_TRIM);
	}
	case FPC_L_TRIM: {
		h2g_ghit hit;
		fg_hit_load(W, fg_frame_hit(S.sp), &hit);
		const int64_t minsc = S.rc_minsc;
		const int64_t floor_ = (S.f_maxsc != F_SMIN && (int64_t)S.f_maxsc > minsc) ? (int64_t)S.f_maxsc : minsc;
		const int64_t tm = (hit.score - floor_) / sc_penalty(sc, 0);
		const uint32_t trimMax = (uint32_t)tm;
		if(hit.rdoff < trimMax) {
			hit.trim5 = hit.rdoff;                            // GenomeHit::trim5 hi_aligner.h:831
			calculate_score(sc, fg_sv(C, S), &hit);
			if((S.f_maxsc == F_SMIN || hit.score > (int64_t)S.f_maxsc) && hit.score >= minsc) {
				if(!fg_hit_store(W, FW_T1, &hit)) F_BAIL(FB_EDITS);
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
		const uint32_t rdlen = FG_GET2(S.rl, S.sv_rdi);
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
		if(!(S.f_maxHitLen < S.f_extoff + 1 && S.f_extoff < FG_GET2(S.rl, S.sv_rdi))) F_GOTO(FPC_R_LS_DONE);
		S.f_extlen = 0; S.f_unique = 0;
		S.localindexatts++;
		if(C.ls->desc[S.f_lidx].len == 0) { S.a0 = 0; S.a1 = 0; S.a2 = S.f_top; S.a3 = S.f_bot; S.a4 = 0; F_GOTO(FPC_R_LS_AFTER); }
		S.a0 = S.f_lidx; S.a1 = S.f_extoff; S.a2 = S.f_maxHitLen; S.a3 = 0; S.a4 = S.f_top; S.a5 = S.f_bot;
		F_OP(FOP_LSEARCH, FPC_R_LS_AFTER);
	}
	case FPC_R_LS_AFTER: {
		const uint32_t rdlen = FG_GET2(S.rl, S.sv_rdi);
		S.f_nelt = S.a0; S.f_extlen = S.a1; S.f_top = S.a2; S.f_bot = S.a3; S.f_unique = S.a4 & 1u;
		if(S.f_extoff < S.f_hitoff + S.f_hitlen) { S.f_noext = 1; F_GOTO(FPC_R_LS_DONE); }
		if(S.f_nelt <= 5) F_GOTO(FPC_R_LS_DONE);
		if(S.f_extoff + 1 < rdlen) S.f_extoff++;
		else { if(S.f_extlen < S.f_maxHitLen) F_GOTO(FPC_R_LS_DONE); else S.f_maxHitLen++; }
		F_GOTO(FPC_R_LS_LOOP);
	}
	case FPC_R_LS_DONE: {
		S.f_ncoords = 0; S.f_ri = 0;
		if(S.f_nelt > 0 && S.f_nelt <= 5 && S.f_extlen >= P.minAnchorLen && !S.f_noext) {
			if(S.f_bot - S.f_top > FG_NCO) F_BAIL(FB_COORDS);
			S.a0 = S.f_lidx; S.a1 = S.f_top; S.a2 = S.f_bot; S.a3 = S.f_extoff + 1 - S.f_extlen; S.a4 = S.f_extlen; S.a5 = fg_frame_co(S.sp);
			F_OP(FOP_LCOORDS, FPC_R_LC_AFTER);
		}
		F_GOTO(FPC_R_FOR_RI);
	}
	case FPC_R_FOR_RI: {
		if(S.f_ri >= (int32_t)S.f_ncoords) F_GOTO(FPC_R_AFTER_FOR);
		const uint32_t cb = fg_frame_co(S.sp) + 3 * (uint32_t)S.f_ri, hb = fg_frame_hit(S.sp);
		const bool fw = (W.ld(hb + 5) & 1u) != 0;
		fg_hit_init(W, FW_T1, fw, S.f_extoff + 1 - S.f_extlen, S.f_extlen, W.ld(cb), W.ld(cb + 1), W.ld(cb + 2));
		h2g_ghit t, fh;
		fg_hit_load(W, FW_T1, &t); fg_hit_load(W, hb, &fh);
		if(!hit_compatible(&fh, &t, P.maxIntronLen, true)) {
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
			F_BAIL(FB_LOCALHITS);
		}
		F_GOTO(FPC_R_FOR_RI);
	}
	case FPC_R_R1: { if(S.ret > S.f_maxsc) S.f_maxsc = S.ret; F_GOTO(FPC_R_FOR_RI); }
	case FPC_R_AFTER_FOR: {
		if(S.f_maxsc != F_SMIN && S.f_maxsc >= S.f_prev - sc.mmpMax) S.f_success = 1;
		F_GOTO(FPC_R_WHILE);                                // (the loop over _local_genomeHits is empty: nlocal == 0)
	}
	case FPC_R_AFTER_WHILE: {
		if(S.f_success) F_RC_RET(S.f_maxsc);
		S.f_ncoords = 0; S.f_ri = 0;
		if(S.f_hitoff + S.f_hitlen + minK + 1 < FG_GET2(S.rl, S.sv_rdi) && S.localindexatts < S.max_localindexatts) F_BAIL(FB_GSEARCH);
		F_GOTO(FPC_R_TRIM);
	}
	case FPC_R_TRIM: {
		h2g_ghit hit;
		fg_hit_load(W, fg_frame_hit(S.sp), &hit);
		const int64_t minsc = S.rc_minsc;
		const uint32_t trimLen = FG_GET2(S.rl, S.sv_rdi) - S.f_hitoff - hit.len - hit.trim5;
		const int64_t floor_ = (S.f_maxsc != F_SMIN && (int64_t)S.f_maxsc > minsc) ? (int64_t)S.f_maxsc : minsc;
		const uint32_t trimMax = (uint32_t)((hit.score - floor_) / sc_penalty(sc, 0));
		if(trimLen < trimMax) {
			hit.trim3 = trimLen;                              // GenomeHit::trim3 hi_aligner.h:855
			calculate_score(sc, fg_sv(C, S), &hit);
			if((S.f_maxsc == F_SMIN || hit.score > (int64_t)S.f_maxsc) && hit.score >= minsc) {
				if(!fg_hit_store(W, FW_T1, &hit)) F_BAIL(FB_EDITS);
				F_RC_CALL(FW_T1, hit.rdoff - hit.trim5, hit.len + hit.trim5 + hit.trim3, FPC_R_R4);
			}
		}
		F_GOTO(FPC_R_EXT);
	}
	case FPC_R_R4: { if(S.ret > S.f_maxsc) S.f_maxsc = S.ret; F_GOTO(FPC_R_EXT); }
	case FPC_R_EXT: {
		const uint32_t hb = fg_frame_hit(S.sp);
		fg_hit_copy(W, FW_T1, hb);
		const uint32_t rdlen = FG_GET2(S.rl, S.sv_rdi);
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
		const uint32_t re = S.a1, hitoff = S.f_hitoff, hitlen = S.f_hitlen, rdlen = FG_GET2(S.rl, S.sv_rdi);
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
			const uint32_t sz = S.nres[0];
			o.nres = sz; o.overflow = 0; o.nrank = S.nrank; o.nsteps = S.nsteps; o.depth = S.nframes_max; o.nside = S.nside;
			for(uint32_t k = 0; k < H2G_SELECT_CAP; k++) o.select[k] = 0;
			// the records are in their slots in report order; selectByScore (al_select) over at most two of them
			int64_t key[FG_NRES]; int64_t scv[FG_NRES];
			h2g_alnres* recs = C.O.aln + (size_t)S.read * C.O.aln_slots;
			for(uint32_t k = 0; k < sz; k++) { scv[k] = recs[k].score; key[k] = fg_hisat2_key(recs[k].score, recs[k].trim5 + recs[k].trim3); }
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
			for(uint32_t k = 0; k < nsel; k++) o.select[k] = (uint8_t)ord[k];
			int64_t b = INT64_MIN, sb = INT64_MIN, bh = 0, sbh = 0;        // AlnSetSumm::init aligner_result.cpp:1209
			for(uint32_t k = 0; k < sz; k++) {
				const int64_t h = key[k], s = scv[k];
				if(b == INT64_MIN || s > b || (s == b && h > bh)) { sb = b; sbh = bh; b = s; bh = h; }
				else if(sb == INT64_MIN || s > sb || (s == sb && h > sbh)) { sb = s; sbh = h; }
			}
			o.best = b == INT64_MIN ? INT32_MIN : (int32_t)b; o.secbest = sb == INT64_MIN ? INT32_MIN : (int32_t)sb;
			o.best_h2 = (uint32_t)(uint64_t)bh; o.secbest_h2 = (uint32_t)(uint64_t)sbh;
			// output slot k holds res[select[k]]
			if(nsel >= 1 && ord[0] == 1) {
				h2g_alnres r0 = recs[0], r1 = recs[1];
				recs[0] = r1; if(nsel == 2) recs[1] = r0;
			}
			C.O.rout[S.read] = o;
			S.rnd = rnd.last;
			S.a0 = nsel > 0;
		} else {
			PairOut o;
			o.nres[0] = S.nres[0]; o.nres[1] = S.nres[1]; o.npairs = S.npairs; o.overflow = 0;
			o.nrank = S.nrank; o.nsteps = S.nsteps; o.depth = S.nframes_max; o.nside = S.nside; o.rnd_state = S.rnd; o.pad = 0;
			for(uint32_t k = 0; k < AL_MAX_PAIRS; k++) {
				o.pair_i[k] = k < S.npairs ? (uint8_t)((S.pairs >> (4 * k)) & 3u) : 0;
				o.pair_j[k] = k < S.npairs ? (uint8_t)((S.pairs >> (4 * k + 2)) & 3u) : 0;
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
	partial_search_item(*C.g, fg_sv(C, S), S.a0, P.pseudogeneStop != 0, P.anchorStop != 0, P.khits, &fh);
	S.a5 = S.a0;
	S.a0 = fh.top; S.a1 = fh.bot;
	S.a2 = (fh.len & 0xffu) | (fh.hit_type << 8) | ((fh.done ? 1u : 0u) << 16) | ((fh.anchorStop ? 1u : 0u) << 17) | (fh.numUniqueSearch << 18);
	S.a3 = fh.cur; S.a4 = (fh.nrank & 0xffffu) | (fh.nside << 16);
}
H2G_HD void fast_op_gcoords(const FCtx& C, FState& S, const FWords& W) {
	h2g_coord co[FG_NCO];
	h2g_sa_result res;
	genome_coords_item(*C.g, S.a0, S.a1, S.a2, S.a3, S.a4 != 0, co, FG_NCO, &res);
	for(uint32_t k = 0; k < res.ncoords && k < FG_NCO; k++) { W.st(S.a5 + 3 * k, co[k].tidx); W.st(S.a5 + 3 * k + 1, co[k].toff); W.st(S.a5 + 3 * k + 2, co[k].joinedOff); }
	S.a0 = res.ncoords; S.a1 = res.nsteps;
}
H2G_HD void fast_op_extend(const FCtx& C, FState& S, const FWords& W) {
	h2g_ghit h;
	fg_hit_load(W, S.a3, &h);
	uint32_t le = H2G_MAX, re = H2G_MAX;
	extend_item(*C.ref, C.P->sc, fg_sv(C, S), &h, S.a0, S.a1, S.a2, &le, &re);
	if(!fg_hit_store(W, S.a3, &h)) { S.pc = FPC_BAIL; S.bail = FB_EDITS; }
	S.a0 = le; S.a1 = re;
}
H2G_HD void fast_op_lsearch(const FCtx& C, FState& S) {
	const AlnParams& P = *C.P;
	uint32_t extlen = 0, top = S.a4, bot = S.a5, nr[2] = {0, 0};
	bool uniqueStop = S.a3 != 0;
	LIdx lx; lx.ls = C.ls; lx.d = &C.ls->desc[S.a0];
	const uint32_t nelt = gfm_search(lx, fg_sv(C, S), S.a1, &extlen, &top, &bot, &uniqueStop, P.minK_local, S.a2, P.kseeds, true, nr);
	S.nrank += nr[0]; S.nside += nr[1];
	S.a0 = nelt; S.a1 = extlen; S.a2 = top; S.a3 = bot; S.a4 = uniqueStop ? 1u : 0u;
}
H2G_HD void fast_op_lcoords(const FCtx& C, FState& S, const FWords& W) {
	LIdx lx; lx.ls = C.ls; lx.d = &C.ls->desc[S.a0];
	h2g_coord co[FG_NCO];
	uint32_t n = 0, steps = 0;
	genome_coords_local(lx, S.a1, S.a2, S.a3, S.a4, co, FG_NCO, &n, &steps);
	S.nsteps += steps;
	for(uint32_t k = 0; k < n; k++) { W.st(S.a5 + 3 * k, co[k].tidx); W.st(S.a5 + 3 * k + 1, co[k].toff); W.st(S.a5 + 3 * k + 2, co[k].joinedOff); }
	S.a0 = n;
}
H2G_HD void fast_op_combine(const FCtx& C, FState& S, const FWords& W) {
	const AlnParams& P = *C.P;
	h2g_ghit a, b;
	fg_hit_load(W, S.a3, &a); fg_hit_load(W, S.a4, &b);
	const bool ok = hit_combine(*C.ref, P.sc, fg_sv(C, S), &a, &b, (int64_t)S.rc_minsc, P.minIntronLen, true,
	                            ScVec{C.sc, C.sc_stride}, ScVec{C.sc + (size_t)H2G_COMBINE_MAXLEN * C.sc_stride, C.sc_stride}, nullptr);
	if(!fg_hit_store(W, S.a3, &a)) { S.pc = FPC_BAIL; S.bail = FB_EDITS; }
	S.a0 = ok ? 1u : 0u;
}
H2G_HD void fast_exec(const FCtx& C, FState& S, const FWords& W, uint32_t op) {
	switch(op) {
	case FOP_PSEARCH: fast_op_psearch(C, S); break;
	case FOP_GCOORDS: fast_op_gcoords(C, S, W); break;
	case FOP_EXTEND:  fast_op_extend(C, S, W); break;
	case FOP_LSEARCH: fast_op_lsearch(C, S); break;
	case FOP_LCOORDS: fast_op_lcoords(C, S, W); break;
	case FOP_COMBINE: fast_op_combine(C, S, W); break;
	default: break;
	}
	S.op = FOP_NONE;
}

// One read / pair on ONE lane until it completes or bails (tests/emul).  true = completed.
H2G_HD bool fast_run_single(const FCtx& C, FState& S, const FWords& W, uint32_t read, bool paired, bool packed_ok) {
	fast_begin(C, S, read, paired, packed_ok);
	while(S.pc != FPC_DONE && S.pc != FPC_BAIL) {
		fast_step(C, S, W);
		if(S.op != FOP_NONE) fast_exec(C, S, W, S.op);
	}
	return S.pc == FPC_DONE;
}

}  // namespace h2g
