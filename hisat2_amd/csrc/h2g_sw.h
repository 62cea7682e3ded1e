// h2g_sw.h — the end-to-end Smith-Waterman of SwAligner as HISAT2 uses it, 8-bit and 16-bit cells.
//
// Reference: frameSeedExtensionRect dp_framer.cpp:81-130; SwAligner::initRef aligner_sw.cpp:137-253;
// alignNucleotidesEnd2EndSseU8 aligner_swsse_ee_u8.cpp:791-1172; gatherCellsNucleotidesEnd2EndSseU8 :1202-1234;
// SwAligner::align / nextAlignment aligner_sw.cpp:477-870; backtraceNucleotidesEnd2EndSseU8 :1309-1900.
//
// The reference fills H/E/F with Farrar's striped SSE loop plus a lazy-F fix-up that runs until F stops
// improving against the *stored* F, so the bytes it leaves in SSEMatrix are exactly the saturating-u8
// recurrences
//     E[i][j] = max(E[i][j-1] (-) rdGapExt,  (H[i][j-1] (-) rdGapOpen) (-) gbar[i])          E[i][0] = 0
//     F[i][j] = max(F[i-1][j] (-) rfGapExt,   H[i-1][j] (-) rfGapOpen) (-) gbar[i]           F[0][j] = 0
//     H[i][j] = max(H[i-1][j-1] (-) pen(i,j), E[i][j], F[i][j])    H[-1][*] = 0xff, H[i>=0][-1] = 0
// ((-) = unsigned saturating subtract; gbar[i] = 0xff inside the gap barrier, else 0), independent of the fill
// order.  That is what lets the GPU fill by anti-diagonals — 64 lanes, one cell each, matrices in LDS — and
// still hand the reference's backtrace byte-identical matrices.  The backtrace itself (deterministic `#if 1`
// tie-breaks, branch stack, reported-through masks) is sequential and runs on lane 0.
//
// 16-bit cells (row a24).  SwAligner::align takes the 8-bit fill when minsc >= -254 and alignNucleotidesEnd2EndSseI16 (aligner_swsse_ee_i16.cpp,
// with its own gather :1216-1249 and backtrace :1324-1915) otherwise (aligner_sw.cpp:496); nothing else selects between them in an end-to-end
// run (readSse16_ is never set).  The i16 cells are signed with saturating adds: 0x7fff = score 0, 0x8000 = "minus infinity", the gap barrier
// is 0x8000 added twice (forces 0x8000).  Read as unsigned after + 0x8000 that is the recurrence above with 0xffff for 0xff — the same code
// at another cell width (SwMats::wide), gather and backtrace decoding with - 0xffff.  One difference outside the matrices: after a candidate the
// 16-bit branch of nextAlignment re-seeds the PRNG with `reseed`, the 8-bit branch with `reseed + 1` (aligner_sw.cpp:906 / :840).
// Pinned by tests/golden/probe_sw16.txt.gz (the reference at --score-min -450 / -900, scores down to -407).
//
// All functions are `__host__ __device__`; tests/emul instantiates them on the host with one "lane".
#pragma once
#include "h2g_core.h"

namespace h2g {

#define H2G_SW_MAX_ROWS 256                 // read length cap of the SW path (longer reads: H2G_ERR_ARG)
#define H2G_SW_NCH ((H2G_SW_MAX_ROWS + 63) / 64)   // row chunks of 64 per lane in the wave-systolic fill
#define H2G_SW_MAXGAP 10                    // readGaps = refGaps = maxhalf = 10 (spliced_aligner.h:222)
#define H2G_SW_MAX_COLS (H2G_SW_MAX_ROWS + 4 * H2G_SW_MAXGAP)
#define H2G_SW_CELLS (H2G_SW_MAX_ROWS + 2 * H2G_SW_MAXGAP + 8)
#define H2G_SW_STACK H2G_SW_CELLS             // branch frames: at most one per cell of the current path (the reference's list is unbounded; overflow is flagged)

struct SwParams {   // Scoring (scoring.h) + the constants of the call site
	DScoring sc;
	int32_t gapbar = 4;          // gGapBarrier hisat2.cpp:419
	uint32_t nceil_pct = 15;     // nCeil = L,0,0.15 (SeedAlignmentPolicy::parseString), evaluated per read length
};

struct SwFrame { uint16_t nedsz, celsz, row, col, gaps; int16_t ns; int32_t score; uint8_t ct; };

// SSEMatrix::masks_ (reportedThru bit 0, H mask 1/2-6, E mask 7/8-9, F mask 10/11-12) is only ever touched along the
// backtraced paths (a few hundred cells of 14 k), so it is kept as a small open-addressing table instead of a matrix.
// Entry = generation << 32 | (row << 8 | col) << 16 | mask; a new problem bumps the generation instead of clearing.
#define H2G_SW_MASK_SLOTS 1024
// A walk that touches more cells than the table holds (many failing candidates of a poor placement: every one of them marks its path) is
// run again over `direct`, a plain matrix of masks (nrow x ncol uint16, cleared first) — sw_finish; without one it ends flagged (`full`).
struct SwMaskTab {
	uint64_t* e;      // [H2G_SW_MASK_SLOTS], zero-initialised once
	uint32_t  gen;    // > 0
	uint32_t  full;   // set when an insert found no slot
	uint16_t* direct = nullptr; uint32_t dcols = 0;
	H2G_HD uint32_t slot(uint32_t key) const { return (key * 40503u >> 4) & (H2G_SW_MASK_SLOTS - 1); }
	// entry = generation (30 bits) | key (18 bits: row << 10 | col; col < 1024 covers 256 rows + 4 * maxgap columns) | mask (16 bits)
	H2G_HD uint32_t get(uint32_t row, uint32_t col) const {
		if(direct) return direct[(size_t)row * dcols + col];
		const uint32_t key = (row << 10) | col;
		uint32_t h = slot(key);
		for(uint32_t n = 0; n < H2G_SW_MASK_SLOTS; n++, h = (h + 1) & (H2G_SW_MASK_SLOTS - 1)) {
			const uint64_t v = e[h];
			if((uint32_t)(v >> 34) != gen) return 0;
			if(((uint32_t)(v >> 16) & 0x3ffffu) == key) return (uint32_t)v & 0xffffu;
		}
		return 0;
	}
	H2G_HD void set(uint32_t row, uint32_t col, uint32_t mask) {
		if(direct) { direct[(size_t)row * dcols + col] = (uint16_t)mask; return; }
		const uint32_t key = (row << 10) | col;
		uint32_t h = slot(key);
		for(uint32_t n = 0; n < H2G_SW_MASK_SLOTS; n++, h = (h + 1) & (H2G_SW_MASK_SLOTS - 1)) {
			const uint64_t v = e[h];
			if((uint32_t)(v >> 34) != gen || ((uint32_t)(v >> 16) & 0x3ffffu) == key) { e[h] = ((uint64_t)gen << 34) | ((uint64_t)key << 16) | (mask & 0xffffu); return; }
		}
		full = 1;
	}
};

// Per-problem working set.  Two layouts of H/E/F:
//   layout 0: row-major [i * ncol + j]                      (single-lane fill: emulator, in-go() use)
//   layout 1: anti-diagonal-major [((i >> 6) * nd + i + j) << 6 | (i & 63)], nd = nrow + ncol - 1
//             — what the wavefront fill writes: the 64 cells of one step are 64 contiguous bytes
struct SwMats {
	uint8_t*  H; uint8_t* E; uint8_t* F;   // cells: uint8_t, or uint16_t when `wide`
	uint8_t*  rf;                          // reference chars 0..4 for the ncol columns
	uint32_t  nrow, ncol, nd, layout;
	uint32_t  wide = 0;                    // 1: 16-bit cells (minsc < -254)
	H2G_HD size_t at(uint32_t i, uint32_t j) const {   // cell index (not a byte offset)
		return layout ? ((((size_t)(i >> 6) * nd + i + j) << 6) | (i & 63u)) : ((size_t)i * ncol + j);
	}
	H2G_HD size_t bytes() const { return (layout ? ((size_t)((nrow + 63) >> 6) * nd) << 6 : (size_t)nrow * ncol) << wide; }
	H2G_HD uint32_t top() const { return wide ? 0xffffu : 0xffu; }   // the cell value of score 0
	H2G_HD uint32_t ld(const uint8_t* M, size_t k) const { return wide ? (uint32_t)reinterpret_cast<const uint16_t*>(M)[k] : (uint32_t)M[k]; }
	H2G_HD void st(uint8_t* M, size_t k, uint32_t v) const { if(wide) reinterpret_cast<uint16_t*>(M)[k] = (uint16_t)v; else M[k] = (uint8_t)v; }
};
H2G_HD bool sw_wide_for(int64_t minsc) { return minsc < -254; }   // aligner_sw.cpp:496

H2G_HD uint32_t subs8(uint32_t a, uint32_t b) { return a > b ? a - b : 0u; }   // unsigned saturating subtract; cells never exceed SwMats::top()
H2G_HD uint32_t max8(uint32_t a, uint32_t b) { return a > b ? a : b; }

// DPRect of frameSeedExtensionRect with maxns = 0, trimToRef = false (dp_framer.cpp:81-130)
struct SwRect { int64_t refl, refr, refl_pretrim, refr_pretrim, triml, trimr, corel, corer; };
H2G_HD SwRect sw_frame(uint32_t refoff, uint32_t rdlen, uint32_t reflen) {
	SwRect r;
	const int64_t maxgap = H2G_SW_MAXGAP;
	int64_t refl = (int64_t)refoff - 2 * maxgap, refr = (int64_t)refoff + ((int64_t)rdlen - 1) + 2 * maxgap;
	r.triml = r.trimr = 0;
	if(refr >= (int64_t)reflen) r.trimr = refr - ((int64_t)reflen - 1);
	if(refl < 0) r.triml = -refl;
	r.refl_pretrim = refl; r.refr_pretrim = refr;
	r.refl = refl + r.triml; r.refr = refr - r.trimr;
	r.corel = maxgap; r.corer = maxgap + 2 * maxgap;
	return r;
}

// -Scoring::score(readc, 1 << refc, q) (scoring.h:259-269) — the query-profile entry (:76-147)
H2G_HD uint32_t sw_pen(const DScoring& sc, int readc, int refc, int q) {
	if(readc > 3 || refc > 3) return (uint32_t)sc.nPen;
	return readc == refc ? 0u : (uint32_t)mm_penalty(sc, q);
}

// One DP cell.  Reads only cells of the two previous anti-diagonals.
H2G_HD void sw_cell(const SwMats& m, const SwParams& P, const SeqView& seq, uint32_t i, uint32_t j) {
	const uint32_t nrow = m.nrow, top = m.top();
	const uint32_t gb = (i < (uint32_t)P.gapbar || (nrow - i - 1) < (uint32_t)P.gapbar) ? top : 0u;
	const uint32_t rdgapo = (uint32_t)(P.sc.rdGapConst + P.sc.rdGapLinear), rdgape = (uint32_t)P.sc.rdGapLinear;
	const uint32_t rfgapo = (uint32_t)(P.sc.rfGapConst + P.sc.rfGapLinear), rfgape = (uint32_t)P.sc.rfGapLinear;
	uint32_t e = 0, f = 0, diag;
	if(j > 0) { const size_t l = m.at(i, j - 1); e = max8(subs8(m.ld(m.E, l), rdgape), subs8(subs8(m.ld(m.H, l), rdgapo), gb)); }
	if(i > 0) { const size_t u = m.at(i - 1, j); f = subs8(max8(subs8(m.ld(m.F, u), rfgape), subs8(m.ld(m.H, u), rfgapo)), gb); }
	diag = i == 0 ? top : (j == 0 ? 0u : m.ld(m.H, m.at(i - 1, j - 1)));
	const uint32_t pen = sw_pen(P.sc, seq.at(i), m.rf[j], seq.qual(i) - 33);
	const size_t at = m.at(i, j);
	m.st(m.E, at, e);
	m.st(m.F, at, f);
	m.st(m.H, at, max8(max8(subs8(diag, pen), e), f));
}

#if defined(__HIP_DEVICE_COMPILE__)
#define H2G_SW_SYNC() __syncthreads()
#else
#define H2G_SW_SYNC() ((void)0)
#endif

// Anti-diagonal fill by `nlanes` cooperating lanes through memory (1 lane: emulator and in-go() use).
// COOP = true adds a barrier per anti-diagonal (lanes of one workgroup sharing the matrices).
template <bool COOP>
H2G_HD void sw_fill(const SwMats& m, const SwParams& P, const SeqView& seq, uint32_t lane, uint32_t nlanes) {
	const uint32_t nrow = m.nrow, ncol = m.ncol;
	for(uint32_t d = 0; d < nrow + ncol - 1; d++) {
		const uint32_t ilo = d >= ncol ? d - ncol + 1 : 0, ihi = d < nrow ? d : nrow - 1;
		for(uint32_t i = ilo + lane; i <= ihi; i += nlanes) sw_cell(m, P, seq, i, d - i);
		if(COOP) H2G_SW_SYNC();
	}
}

#if defined(__HIPCC__)
// Wavefront fill (layout 1): lane l owns rows l, 64 + l, 128 + l.  At step d it computes cell (i, d - i) of each of its
// rows from its own previous cell (H/E to the left) and the previous cells of the lane above (H/F up, H diagonal), which
// arrive by one __shfl_up per chunk — the packed word also conveys the reference character down the diagonal.
// No LDS, no barriers; the three result cells of a step are 64 contiguous cells per matrix (coalesced stores).
//   packed word (CELL = uint8_t):  h_cur | h_old << 8  | f_cur << 16 | refc << 24      (32 bits)
//   packed word (CELL = uint16_t): h_cur | h_old << 16 | f_cur << 32 | refc << 48      (64 bits: two 32-bit shuffles)
template <typename CELL> struct SwPack;
template <> struct SwPack<uint8_t>  { typedef uint32_t W; static constexpr int B = 8;  static constexpr uint32_t TOP = 0xffu; };
template <> struct SwPack<uint16_t> { typedef uint64_t W; static constexpr int B = 16; static constexpr uint32_t TOP = 0xffffu; };
__device__ inline uint32_t sw_shfl_up1(uint32_t v) { return __shfl_up(v, 1); }
__device__ inline uint64_t sw_shfl_up1(uint64_t v) { return (uint64_t)__shfl_up((uint32_t)v, 1) | ((uint64_t)__shfl_up((uint32_t)(v >> 32), 1) << 32); }
__device__ inline uint32_t sw_shfl_63(uint32_t v) { return __shfl(v, 63); }
__device__ inline uint64_t sw_shfl_63(uint64_t v) { return (uint64_t)__shfl((uint32_t)v, 63) | ((uint64_t)__shfl((uint32_t)(v >> 32), 63) << 32); }
template <typename CELL>
__device__ inline void sw_fill_wave_t(const SwMats& m, const SwParams& P, const SeqView& seq, uint32_t lane) {
	typedef typename SwPack<CELL>::W W;
	constexpr int B = SwPack<CELL>::B;
	constexpr uint32_t TOP = SwPack<CELL>::TOP;
	CELL* const mH = reinterpret_cast<CELL*>(m.H); CELL* const mE = reinterpret_cast<CELL*>(m.E); CELL* const mF = reinterpret_cast<CELL*>(m.F);
	const uint32_t nrow = m.nrow, ncol = m.ncol, nd = m.nd;
	const uint32_t nch = (nrow + 63) >> 6;
	const uint32_t rdgapo = (uint32_t)(P.sc.rdGapConst + P.sc.rdGapLinear), rdgape = (uint32_t)P.sc.rdGapLinear;
	const uint32_t rfgapo = (uint32_t)(P.sc.rfGapConst + P.sc.rfGapLinear), rfgape = (uint32_t)P.sc.rfGapLinear;
	W pk[H2G_SW_NCH]; uint32_t e_cur[H2G_SW_NCH];
	int readc[H2G_SW_NCH]; uint32_t mmpen[H2G_SW_NCH], gb[H2G_SW_NCH];
#pragma unroll
	for(int c = 0; c < H2G_SW_NCH; c++) {
		pk[c] = 0; e_cur[c] = 0;
		const uint32_t i = (uint32_t)c * 64 + lane;
		const bool in = i < nrow;
		readc[c] = in ? seq.at(i) : 4;
		mmpen[c] = in ? (uint32_t)mm_penalty(P.sc, seq.qual(i) - 33) : 0u;
		gb[c] = (in && (i < (uint32_t)P.gapbar || (nrow - i - 1) < (uint32_t)P.gapbar)) ? TOP : 0u;
	}
	for(uint32_t d = 0; d < nd; d++) {
		W up[H2G_SW_NCH];
		const uint32_t fresh = d < ncol ? (uint32_t)m.rf[d] : 4u;     // row 0 meets column d at step d
#pragma unroll
		for(int c = 0; c < H2G_SW_NCH; c++) {
			if((uint32_t)c >= nch) break;
			W u = sw_shfl_up1(pk[c]);
			if(c > 0) { const W w = sw_shfl_63(pk[c - 1]); if(lane == 0) u = w; }
			else if(lane == 0) u = (W)fresh << (3 * B);
			up[c] = u;
		}
#pragma unroll
		for(int c = 0; c < H2G_SW_NCH; c++) {
			if((uint32_t)c >= nch) break;
			const uint32_t i = (uint32_t)c * 64 + lane;
			const int32_t j = (int32_t)d - (int32_t)i;
			if(i < nrow && j >= 0 && j < (int32_t)ncol) {
				const uint32_t h_left = (uint32_t)pk[c] & TOP, e_left = e_cur[c];
				const uint32_t up_h = (uint32_t)up[c] & TOP, up_hold = (uint32_t)(up[c] >> B) & TOP, up_f = (uint32_t)(up[c] >> (2 * B)) & TOP, refc = (uint32_t)(up[c] >> (3 * B));
				const uint32_t e = j == 0 ? 0u : max8(subs8(e_left, rdgape), subs8(subs8(h_left, rdgapo), gb[c]));
				const uint32_t f = i == 0 ? 0u : subs8(max8(subs8(up_f, rfgape), subs8(up_h, rfgapo)), gb[c]);
				const uint32_t diag = i == 0 ? TOP : (j == 0 ? 0u : up_hold);
				const uint32_t pen = (readc[c] > 3 || refc > 3) ? (uint32_t)P.sc.nPen : ((uint32_t)readc[c] == refc ? 0u : mmpen[c]);
				const uint32_t h = max8(max8(subs8(diag, pen), e), f);
				const size_t at = ((((size_t)c * nd + d) << 6) | lane);
				mH[at] = (CELL)h; mE[at] = (CELL)e; mF[at] = (CELL)f;
				pk[c] = (W)h | ((W)h_left << B) | ((W)f << (2 * B)) | ((W)refc << (3 * B));
				e_cur[c] = e;
			}
		}
	}
}
__device__ inline void sw_fill_wave(const SwMats& m, const SwParams& P, const SeqView& seq, uint32_t lane) {
	if(m.wide) sw_fill_wave_t<uint16_t>(m, P, seq, lane); else sw_fill_wave_t<uint8_t>(m, P, seq, lane);
}
#endif

struct SwOut {   // mirrors h2g_sw_result
	int32_t  found_align, found;
	int32_t  best, score;
	int64_t  off;                // refcoord().off()
	uint32_t nedits, gaps, overflow, rnd;
	int64_t  refl, refr;
	h2g_edit edits[H2G_GHIT_EDITS];   // (== H2G_MAX_EDITS wherever the struct crosses the C ABI: h2g_kernels.hip)
};

H2G_HD uint32_t sw_lcg_next(uint32_t* last) {   // RandomSource::nextU32 random_source.h:52-61
	*last = 1664525u * *last + 1013904223u;
	const uint32_t ret = *last >> 16;
	*last = 1664525u * *last + 1013904223u;
	return ret ^ *last;
}

H2G_HD char sw_mask2dna(int refm) {   // alphabet.cpp:71-89 for the masks that occur here (1, 2, 4, 8, 16)
	return refm == 1 ? 'A' : refm == 2 ? 'C' : refm == 4 ? 'G' : refm == 8 ? 'T' : 'N';
}

// gather (:1202-1234) + candidate order (aligner_sw_nuc.h:149) + nextAlignment loop (aligner_sw.cpp:709-870) +
// backtrace (:1309-1900).  Sequential; call from one lane after sw_fill.  `rnd` = RandomSource::last.
H2G_HD void sw_gather_backtrace(const SwMats& m, const SwParams& P, const SeqView& seq, const SwRect& rect, int64_t minsc,
                                int nceil, uint32_t* rnd, SwMaskTab& mt, SwFrame* stack, uint16_t* cells /* [2 * H2G_SW_CELLS] */, SwOut* o)
{
	const uint32_t nrow = m.nrow, ncol = m.ncol;
	const int64_t rdgapo = P.sc.rdGapConst + P.sc.rdGapLinear, rdgape = P.sc.rdGapLinear;
	const int64_t rfgapo = P.sc.rfGapConst + P.sc.rfGapLinear, rfgape = P.sc.rfGapLinear;
	const int64_t top = (int64_t)m.top();
	uint32_t lrmax = 0;
	for(uint32_t j = 0; j < ncol; j++) { const uint32_t v = m.ld(m.H, m.at(nrow - 1, j)); if(v > lrmax) lrmax = v; }
	o->best = (int32_t)((int64_t)lrmax - top);
	o->found_align = 0; o->found = 0; o->score = 0; o->off = 0; o->nedits = 0; o->gaps = 0; o->overflow = 0;
	if((int64_t)o->best < minsc || lrmax == 0) return;         // flag -1 / -2 (:1140-1165)
	// Candidates = last-row cells with score >= minsc, visited best score first, then rightmost column first.
	// Instead of sorting a list, repeatedly take the next (score, col) in that order: O(ncol) per candidate.
	uint32_t prev_v = m.top() + 1, prev_col = 0;
	bool any = false;
	while(true) {
		// next candidate strictly after (prev_v, prev_col) in (score desc, col desc) order
		uint32_t best_v = 0, best_col = 0;
		bool have = false;
		for(uint32_t j = 0; j < ncol; j++) {
			const uint32_t v = m.ld(m.H, m.at(nrow - 1, j));
			if((int64_t)v - top < minsc) continue;
			any = true;
			const bool after = v < prev_v || (v == prev_v && j < prev_col);
			if(!after) continue;
			if(!have || v > best_v || (v == best_v && j > best_col)) { have = true; best_v = v; best_col = j; }
		}
		o->found_align = any;
		if(!have) break;
		prev_v = best_v; prev_col = best_col;
		uint32_t row = nrow - 1, col = best_col;
		if(mt.get(row, col) & 1) continue;                 // BT_CAND_FATE_FILT_START
		const uint32_t reseed = sw_lcg_next(rnd) + 1;          // aligner_sw.cpp:766-767
		// ---- backtrace
		uint32_t nstack = 0, ncells = 0, ned = 0, gaps = 0;
		int64_t score = 0; int ns = 0;
		int ct = 0;                                            // SSEMatrix::H / E / F = 0 / 1 / 2
		bool ok = false;
		while(true) {
			const int readc = seq.at(row);
			const int refm = 1 << m.rf[col];
			bool empty = false, canMoveThru = true, branch = false;
			int cur = -1;
			uint16_t mk = (uint16_t)mt.get(row, col);
			if(mk & 1) canMoveThru = false;
			else if(row > 0) {
				const bool gapsAllowed = !(row < (uint32_t)P.gapbar || (nrow - row - 1) < (uint32_t)P.gapbar);
				if(ct == 1) {                                  // E: gap open from H-left or extension from E-left
					const int64_t sc_cur = (int64_t)m.ld(m.E, m.at(row, col)) - top;
					int mask = 0;
					if((int64_t)m.ld(m.H, m.at(row, col - 1)) - top - rdgapo == sc_cur) mask |= 1;
					if((int64_t)m.ld(m.E, m.at(row, col - 1)) - top - rdgape == sc_cur) mask |= 2;
					const int origMask = mask;
					if(mk & (1 << 7)) mask = (mk >> 8) & 3;
					int nm = -1;
					if(mask == 3) { cur = 3; nm = 2; branch = true; }
					else if(mask == 2) { cur = 4; nm = 0; }
					else if(mask == 1) { cur = 3; nm = 0; }
					else { empty = true; canMoveThru = (origMask == 0); }
					if(nm >= 0) mk = (uint16_t)((mk & ~(7 << 7)) | (1 << 7) | (nm << 8));
				} else if(ct == 2) {                           // F: gap open from H-up or extension from F-up
					const int64_t sc_cur = (int64_t)m.ld(m.F, m.at(row, col)) - top;
					int mask = 0;
					if((int64_t)m.ld(m.H, m.at(row - 1, col)) - top - rfgapo == sc_cur) mask |= 1;
					if((int64_t)m.ld(m.F, m.at(row - 1, col)) - top - rfgape == sc_cur) mask |= 2;
					const int origMask = mask;
					if(mk & (1 << 10)) mask = (mk >> 11) & 3;
					int nm = -1;
					if(mask == 3) { cur = 1; nm = 2; branch = true; }
					else if(mask == 2) { cur = 2; nm = 0; }
					else if(mask == 1) { cur = 1; nm = 0; }
					else { empty = true; canMoveThru = (origMask == 0); }
					if(nm >= 0) mk = (uint16_t)((mk & ~(7 << 10)) | (1 << 10) | (nm << 11));
				} else {
					const int64_t sc_cur = (int64_t)m.ld(m.H, m.at(row, col)) - top;
					const bool hasl = col > 0;
					int64_t sc_diag;                           // Scoring::score scoring.h:259
					if(readc > 3 || refm > 15) sc_diag = -P.sc.nPen;
					else sc_diag = (refm & (1 << readc)) ? 0 : -mm_penalty(P.sc, seq.qual(row) - 33);
					int mask = 0;
					if(gapsAllowed) {
						if(sc_cur == (int64_t)m.ld(m.H, m.at(row - 1, col)) - top - rfgapo) mask |= 1;
						if(hasl && sc_cur == (int64_t)m.ld(m.H, m.at(row, col - 1)) - top - rdgapo) mask |= 2;
						if(sc_cur == (int64_t)m.ld(m.F, m.at(row - 1, col)) - top - rfgape) mask |= 4;
						if(hasl && sc_cur == (int64_t)m.ld(m.E, m.at(row, col - 1)) - top - rdgape) mask |= 8;
					}
					if(hasl && sc_cur == (int64_t)m.ld(m.H, m.at(row - 1, col - 1)) - top + sc_diag) mask |= 16;
					const int origMask = mask;
					if(mk & (1 << 1)) mask = (mk >> 2) & 31;
					const int opts = __builtin_popcount((unsigned)mask);
					int select = -1, nm = -1;
					if(opts == 1) { select = __builtin_ctz((unsigned)mask); nm = 0; }
					else if(opts > 1) {                        // the `#if 1` order: diag, H up, F up, H left, E left
						if(mask & 16) select = 4; else if(mask & 1) select = 0; else if(mask & 4) select = 2;
						else if(mask & 2) select = 1; else select = 3;
						nm = mask & ~(1 << select);
						branch = true;
					}
					if(nm >= 0) mk = (uint16_t)((mk & ~(31 << 1)) | (1 << 1) | (nm << 2));   // hMaskSet clears 5 bits at offset 1
					if(select == 4) cur = 0; else if(select == 0) cur = 1; else if(select == 1) cur = 3;
					else if(select == 2) cur = 2; else if(select == 3) cur = 4;
					else { empty = true; canMoveThru = (origMask == 0); }
				}
			}
			mt.set(row, col, (uint32_t)mk | 1u);                     // setReportedThrough
			if(mt.full) { o->overflow = 1; return; }               // the table is full: without the marks the walk would not terminate (sw_finish runs it again over a mask matrix)
			if(!canMoveThru) {
				if(nstack == 0) break;                         // give up on this candidate
				const SwFrame& fr = stack[--nstack];
				ncells = fr.celsz; ned = fr.nedsz; row = fr.row; col = fr.col; gaps = fr.gaps; ns = fr.ns; score = fr.score; ct = fr.ct;
				continue;
			}
			if(empty || row == 0) {
				if(ncells < H2G_SW_CELLS) { cells[2 * ncells] = (uint16_t)row; cells[2 * ncells + 1] = (uint16_t)col; }
				ncells++;
				ok = true;
				break;
			}
			if(branch) {
				if(nstack < H2G_SW_STACK) {
					SwFrame& fr = stack[nstack];
					fr.nedsz = (uint16_t)ned; fr.celsz = (uint16_t)ncells; fr.row = (uint16_t)row; fr.col = (uint16_t)col; fr.gaps = (uint16_t)gaps;
					fr.ns = (int16_t)ns; fr.score = (int32_t)score; fr.ct = (uint8_t)ct;
					nstack++;
				} else o->overflow = 1;
			}
			if(ncells < H2G_SW_CELLS) { cells[2 * ncells] = (uint16_t)row; cells[2 * ncells + 1] = (uint16_t)col; }
			else o->overflow = 1;
			ncells++;
			h2g_edit ed;
			ed.pad = 0; ed.snp = H2G_MAX;
			bool has_edit = true;
			if(cur == 0) {                                     // SW_BT_OALL_DIAG
				const int mt = (refm >= 16 || readc > 3) ? -1 : ((refm >> readc) & 1);
				ct = 0;
				if(mt != 1) {
					ed.pos = row; ed.chr = (uint8_t)sw_mask2dna(refm); ed.qchr = (uint8_t)"ACGTN"[readc]; ed.type = H2G_EDIT_MM;
					score -= (readc > 3 || refm > 15) ? P.sc.nPen : mm_penalty(P.sc, seq.qual(row) - 33);
				} else has_edit = false;
				if(mt == -1) ns++;
				row--; col--;
			} else if(cur == 1 || cur == 2) {                  // REF_OPEN / RFGAP_EXTEND: up
				ed.pos = row; ed.chr = '-'; ed.qchr = (uint8_t)"ACGTN"[readc]; ed.type = H2G_EDIT_REF_GAP;
				row--; ct = cur == 1 ? 0 : 2; score -= cur == 1 ? rfgapo : rfgape; gaps++;
			} else {                                           // READ_OPEN / RDGAP_EXTEND: left
				ed.pos = row + 1; ed.chr = (uint8_t)sw_mask2dna(refm); ed.qchr = '-'; ed.type = H2G_EDIT_READ_GAP;
				col--; ct = cur == 3 ? 0 : 1; score -= cur == 3 ? rdgapo : rdgape; gaps++;
			}
			if(has_edit) {
				if(ned < H2G_GHIT_EDITS) o->edits[ned] = ed;        // (more edits than a record holds: flagged below, if this walk is the one reported)
				ned++;
			}
		}
		if(ok) {                                               // must overlap a core diagonal (:1776-1804)
			bool overlapped = false;
			for(uint32_t k = 0; k < ncells && k < H2G_SW_CELLS && !overlapped; k++) {
				const int64_t diagi = (int64_t)cells[2 * k + 1] - (int64_t)cells[2 * k] + rect.triml;
				if(diagi >= 0 && diagi >= rect.corel && diagi <= rect.corer) overlapped = true;
			}
			if(!overlapped) ok = false;
		}
		if(ok) {                                               // the first aligned column (:1805-1830)
			const int readc = seq.at(row), refm = 1 << m.rf[col];
			const int mt = (refm >= 16 || readc > 3) ? -1 : ((refm >> readc) & 1);
			if(mt != 1) {
				h2g_edit ed;
				ed.pad = 0; ed.snp = H2G_MAX; ed.pos = row; ed.chr = (uint8_t)sw_mask2dna(refm); ed.qchr = (uint8_t)"ACGTN"[readc]; ed.type = H2G_EDIT_MM;
				if(ned < H2G_GHIT_EDITS) o->edits[ned] = ed;
				ned++;
				score -= (readc > 3 || refm > 15) ? P.sc.nPen : mm_penalty(P.sc, seq.qual(row) - 33);
			}
			if(mt == -1) ns++;
			if(ns > nceil) ok = false;
		}
		*rnd = m.wide ? reseed : reseed + 1;                   // aligner_sw.cpp:840 (8-bit branch) / :906 (16-bit branch: rnd.init(reseed))
		if(ok) {
			if(ned > H2G_GHIT_EDITS) o->overflow = 1;
			const uint32_t n = ned < H2G_GHIT_EDITS ? ned : H2G_GHIT_EDITS;
			for(uint32_t a = 0; a < n / 2; a++) { h2g_edit t = o->edits[a]; o->edits[a] = o->edits[n - 1 - a]; o->edits[n - 1 - a] = t; }   // res.reverse()
			o->found = 1; o->score = (int32_t)score; o->nedits = n; o->off = (int64_t)col + rect.refl; o->gaps = gaps;
			break;
		}
	}
	if(mt.full) o->overflow = 1;
}

// Per-lane backtrace state that persists across problems: the mask table (+ its generation), branch stack, cell list, result
struct SwLaneState {
	uint32_t gen, pad[3];
	uint64_t mask[H2G_SW_MASK_SLOTS];
	SwFrame  stack[H2G_SW_STACK];
	uint16_t cells[2 * H2G_SW_CELLS];
	SwOut    out;
};
H2G_HD size_t sw_cell_bytes(uint32_t nrow, uint32_t ncol, bool wide) { return ((((size_t)nrow * ncol) << (wide ? 1 : 0)) + 15) & ~(size_t)15; }
// in-go() scratch of one lane for reads up to `maxlen`: SwLaneState + row-major H/E/F (16-bit cells when `wide`) + the reference window +
// the mask matrix of the rare second walk (SwMaskTab::direct)
H2G_HD size_t sw_scratch_bytes(uint32_t maxlen, bool wide) {
	const uint32_t ncol = maxlen + 4 * H2G_SW_MAXGAP;
	return ((sizeof(SwLaneState) + 15) & ~(size_t)15) + 3 * sw_cell_bytes(maxlen, ncol, wide) + ((ncol + 15) & ~15u) + sw_cell_bytes(maxlen, ncol, true);
}

// gather + backtrace of one filled problem on the calling lane, using its persistent SwLaneState (zero-initialised once).
// `direct`: nrow * ncol uint16 of scratch (contents arbitrary) for the walk that outgrows the mask table, or nullptr (such a walk ends flagged).
H2G_HD SwOut* sw_finish(const SwMats& m, const SwParams& P, const SeqView& sv, const SwRect& rect, int64_t minsc, uint32_t* rnd, SwLaneState* ls,
                        uint16_t* direct, int nceil_given = -1 /* >= 0: instead of nCeil(read length): tests */) {
	SwMaskTab mt;
	if(++ls->gen >= (1u << 30)) { for(uint32_t k = 0; k < H2G_SW_MASK_SLOTS; k++) ls->mask[k] = 0; ls->gen = 1; }   // generation field is 30 bits
	mt.e = ls->mask; mt.gen = ls->gen; mt.full = 0;
	SwOut* o = &ls->out;
	o->refl = rect.refl; o->refr = rect.refr;
	const uint32_t rnd0 = *rnd;
	const int nceil = nceil_given >= 0 ? nceil_given : (int)((double)P.nceil_pct * 0.01 * (double)m.nrow);
	sw_gather_backtrace(m, P, sv, rect, minsc, nceil, rnd, mt, ls->stack, ls->cells, o);
	if(mt.full && direct) {
		const size_t n = (size_t)m.nrow * m.ncol;
		for(size_t k = 0; k < n; k++) direct[k] = 0;
		mt.direct = direct; mt.dcols = m.ncol; mt.full = 0;
		*rnd = rnd0;
		sw_gather_backtrace(m, P, sv, rect, minsc, nceil, rnd, mt, ls->stack, ls->cells, o);
	}
	o->rnd = *rnd;
	return o;
}

// The whole call site (spliced_aligner.h:209-262 without genomeHit bookkeeping) run by ONE lane over scratch memory
// (sw_scratch_bytes(read length, sw_wide_for(minsc))).
H2G_HD void sw_align_single(const DRef& ref, const SwParams& P, const SeqView& sv, uint32_t tidx, uint32_t refoff, int64_t minsc,
                            uint32_t* rnd, uint8_t* scratch, SwOut** out)
{
	const uint32_t nrow = sv.len;
	const SwRect rect = sw_frame(refoff, nrow, ref.refLens[tidx]);
	const uint32_t ncol = (uint32_t)(rect.refr - rect.refl + 1);
	const bool wide = sw_wide_for(minsc);
	const size_t cs = sw_cell_bytes(nrow, ncol, wide);
	SwLaneState* ls = reinterpret_cast<SwLaneState*>(scratch);
	uint8_t* p = scratch + ((sizeof(SwLaneState) + 15) & ~(size_t)15);
	SwMats m;
	m.nrow = nrow; m.ncol = ncol; m.nd = nrow + ncol - 1; m.layout = 0; m.wide = wide;
	m.H = p; m.E = p + cs; m.F = p + 2 * cs; m.rf = p + 3 * cs;
	RefCursor rc;
	rc.init(&ref, tidx);
	for(uint32_t j = 0; j < ncol; j++) m.rf[j] = (uint8_t)rc.get(rect.refl + (int64_t)j);
	sw_fill<false>(m, P, sv, 0, 1);
	*out = sw_finish(m, P, sv, rect, minsc, rnd, ls, reinterpret_cast<uint16_t*>(p + 3 * cs + ((ncol + 15) & ~15u)));
}

}  // namespace h2g
