// h2g_core.h — per-item device functions of the HISAT2 hot path (gfx950), shared by every kernel.
//
// Every function is `__host__ __device__` so that the identical source can be instantiated on the host
// by tests/emul (logic tests in the GPU-less build container).  The product library only ever launches
// the __global__ wrappers in h2g_kernels.hip; it contains no host execution path.
//
// Reference semantics (HISAT2 2.2.3) are cited per function as file:line.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>
#include "../../include/h2g.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define H2G_HD __host__ __device__ __forceinline__
// out-of-line device functions: the graph paths are large and called from many places of the go() state machine; forcing
// them inline makes the kernels explode (compile time, registers) and would tax the linear path that never runs them
#define H2G_HDN __host__ __device__ __noinline__ inline
#else
#define H2G_HD inline
#define H2G_HDN inline
#endif

namespace h2g {

// ------------------------------------------------------------------------------------------ device views
struct DGfm {   // one 32-bit GFM resident in HBM (sides exactly as on disk: 64 B linear / 128 B graph)
	const uint8_t*  sides;
	const uint32_t* ftab;
	const uint32_t* eftab;
	const uint32_t* offs;
	const uint32_t* rstarts;
	const uint32_t* plen;
	const uint32_t* zoffs;   // all '$' rows (graph indexes may hold several; zoff == zoffs[0])
	uint32_t fchr[5];
	uint32_t len, gbwtLen, ftabLim, sideGbwtLen, sideGbwtSz, lineRate, offRate, offMask, ftabChars;
	uint32_t nFrag, nPat, nZ, zoff, minK, linear;
	// geometry of a GRAPH side of the global index (policy interface of h2g_graph.h): 208 symbols in 52 B, F bits at 52,
	// M bits at 78, u32 {F_loc, M_occ, occ[4]} at 104
	static constexpr uint32_t SYMS = 208, NCW = 7, F_OFF = 52, M_OFF = 78, HDR = 104, WSZ = 4;
	H2G_HD uint32_t offs_at(uint32_t i) const { return offs[i]; }
};

struct DRef {   // BitPairReference: records sorted by (text, offset); buf 2 bit/base
	const uint8_t*  buf;
	const uint32_t* rec_start;
	const uint32_t* rec_len;
	const uint32_t* rec_bufoff;
	const uint32_t* refRecOffs;
	const uint32_t* refLens;
	uint32_t nrefs;
};

struct DReads {
	const uint8_t*  codes;   // 0..4 per base, forward strand
	const uint32_t* offs;    // [n+1]
	const char*     quals;   // phred+33 or nullptr (FASTA => 'I')
	uint32_t n;
	// optional per-lane LDS copy of ONE read (k_align): H2G_PK_WORDS 2-bit words then H2G_PK_WORDS/2 N-mask words,
	// word k of this lane at pk[k * pk_stride] (lane-interleaved => conflict-free ds_read_b32)
	const uint32_t* pk = nullptr;
	uint32_t pk_stride = 0, pk_read = 0xffffffffu;
};
#define H2G_PK_WORDS 8           // 128 bases (longer reads are read from HBM base by base)
#define H2G_PK_MAXLEN (H2G_PK_WORDS * 16)

struct DScoring {  // Scoring defaults scoring.h:29-87 / hisat2.cpp:425-441
	int mmpMax = 6, mmpMin = 2, nPen = 1, rdGapConst = 5, rdGapLinear = 3, rfGapConst = 5, rfGapLinear = 3;
	int scMax = 2, scMin = 1, matchBonus = 0;
	// spliced alignment (hisat2.cpp:493-497): --pen-cansplice 0, --pen-noncansplice 12, conflicting splice directions 1000000,
	// --pen-canintronlen / --pen-noncanintronlen G,-8,1 (SimpleFunc LOG: C + L ln(x)); anchor / intron limits of tp.h
	int cp = 0, ncp = 12, csp = 1000000;
	double icpC = -8.0, icpL = 1.0, incpC = -8.0, incpL = 1.0;
	uint32_t icpT = 4, incpT = 4;                  // SimpleFunc type of the intron-length penalties (4 = log)
	uint32_t minAnchorLen = 7, minAnchorLen_noncan = 14, maxIntronLen = 500000;
	// SpliceSiteDB::probscore tables (splice_site.cpp:45-105), resident in HBM; nullptr when spliced alignment is off
	const float* donor_sum = nullptr; const float* acc_sum1 = nullptr; const float* acc_sum2 = nullptr;
};

// ---- splice edits (EDIT_TYPE_SPL): see include/h2g.h for the packing
enum { H2G_SPL_UNKNOWN = 1, H2G_SPL_FW, H2G_SPL_RC, H2G_SPL_SEMI_FW, H2G_SPL_SEMI_RC };   // splice_site.h:37-43
#define H2G_SPL_DONOR_EXONIC 3
#define H2G_SPL_DONOR_INTRONIC 6
#define H2G_SPL_ACC_INTRONIC 14
#define H2G_SPL_ACC_EXONIC 1
#define H2G_SPL_INTRONIC 14          // max(donor_intronic_len, acceptor_intronic_len) splice_site.h:76
#define H2G_SPL_MAXLEN 0xfffffu
H2G_HD uint32_t spl_len(const h2g_edit& e) { return (uint32_t)e.chr | ((uint32_t)e.qchr << 8) | (((uint32_t)e.pad & 15u) << 16); }
H2G_HD uint32_t spl_dir(const h2g_edit& e) { return ((uint32_t)e.pad >> 4) & 7u; }
H2G_HD bool spl_known(const h2g_edit& e) { return (e.pad >> 7) != 0; }
H2G_HD float spl_prob(const h2g_edit& e) { float f; memcpy(&f, &e.snp, 4); return f; }
H2G_HD h2g_edit make_spl_edit(uint32_t pos, uint32_t len, uint32_t dir, bool known, float prob) {
	h2g_edit e;
	e.pos = pos; e.type = H2G_EDIT_SPL;
	e.chr = (uint8_t)(len & 0xff); e.qchr = (uint8_t)((len >> 8) & 0xff); e.pad = (uint8_t)(((len >> 16) & 15u) | ((dir & 7u) << 4) | (known ? 0x80u : 0u));
	memcpy(&e.snp, &prob, 4);
	return e;
}
// SpliceSiteDB::probscore splice_site.cpp:836-846 (old probability model: NEW_PROB_MODEL is not defined)
H2G_HD float spl_probscore(const DScoring& sc, int64_t donor_seq, int64_t acceptor_seq) {
	float p = sc.donor_sum[donor_seq];
	p *= sc.acc_sum1[(int)(acceptor_seq >> 16)];            // acceptor_len2 = 8 bases
	p *= sc.acc_sum2[(int)(acceptor_seq % (1 << 16))];
	return (float)(1.0 / (1.0 + (double)p));
}
// ------------------------------------------------------------------------------------------ splice-site database
// SpliceSiteDB (splice_site.h:470-640) as two sorted arrays: `fw` ordered by (text, left, right, dir) — the reference's _fwIndex —
// and `bw` ordered by (text, right, left, dir) — its _bwIndex (SpliceSitePos::operator< splice_site.h:137).  Sites read from a
// file (--known-splicesite-infile / --novel-splicesite-infile, SpliceSiteDB::read splice_site.cpp:727) are visible to every read;
// a site found by read `readid` is visible to read r only if readid + window <= r (spliced_aligner.h:432).
struct DSpliceSite { uint32_t left, right, readid; uint8_t dir, fromfile, known, pad; };
struct DSpliceDB {
	const DSpliceSite* fw = nullptr; const DSpliceSite* bw = nullptr;
	const uint32_t* fw_first = nullptr; const uint32_t* bw_first = nullptr;   // [nPat + 1]: first entry of each text
	uint32_t n = 0, window = 0;
};
H2G_HD bool ss_visible(const DSpliceDB& db, const DSpliceSite& s, uint32_t rdid) {
	return s.fromfile || !((uint64_t)s.readid + db.window > (uint64_t)rdid);
}
// getLeftSpliceSites splice_site.cpp:370: sites whose RIGHT end lies in [left + 1 - range, left], in _bwIndex order;
// getRightSpliceSites :385: sites whose LEFT end lies in [right, right + range - 1], in _fwIndex order.
// out[k] = {left, right, dir}; returns the number found (the caller flags counts above cap).
H2G_HD uint32_t ss_range(const DSpliceSite* a, uint32_t lo, uint32_t hi, bool by_right, uint32_t klo, uint32_t khi, const DSpliceDB& db,
                         uint32_t rdid, h2g_coord* out, uint32_t cap) {
	uint32_t x = lo, y = hi;
	while(x < y) { const uint32_t m = x + ((y - x) >> 1); if((by_right ? a[m].right : a[m].left) < klo) x = m + 1; else y = m; }
	uint32_t n = 0;
	for(; x < hi; x++) {
		const DSpliceSite s = a[x];
		if((by_right ? s.right : s.left) > khi) break;
		if(!ss_visible(db, s, rdid)) continue;                // the callers skip these (:432, :554, :699, :1380)
		if(n < cap) { out[n].tidx = s.left; out[n].toff = s.right; out[n].joinedOff = s.dir; }
		n++;
	}
	return n;
}
H2G_HD uint32_t ss_left_sites(const DSpliceDB& db, uint32_t tidx, uint32_t left, uint32_t range, uint32_t rdid, h2g_coord* out, uint32_t cap) {
	if(!db.n) return 0;
	return ss_range(db.bw, db.bw_first[tidx], db.bw_first[tidx + 1], true, left + 1 - range, left, db, rdid, out, cap);
}
H2G_HD uint32_t ss_right_sites(const DSpliceDB& db, uint32_t tidx, uint32_t right, uint32_t range, uint32_t rdid, h2g_coord* out, uint32_t cap) {
	if(!db.n) return 0;
	return ss_range(db.fw, db.fw_first[tidx], db.fw_first[tidx + 1], false, right, right + range - 1, db, rdid, out, cap);
}
// MaxIntronLen / MaxIntronLen_noncan hi_aligner.h:48-79
H2G_HD uint32_t max_intron_len(uint32_t anchor, uint32_t minAnchorLen) {
	if(anchor < minAnchorLen) return 0;
	if(anchor < 2) anchor = 2;
	uint32_t shift = (anchor << 1) - 4;
	shift = shift < 13 ? 13 : shift; shift = shift > 30 ? 30 : shift;
	return 1u << shift;
}
H2G_HD uint32_t max_intron_len_noncan(uint32_t anchor, uint32_t minAnchorLen_noncan) {
	if(anchor < minAnchorLen_noncan) return 0;
	if(anchor < 5) anchor = 5;
	uint32_t shift = (anchor << 1) - 10;
	shift = shift > 30 ? 30 : shift;
	return 1u << shift;
}
// Scoring::canSpl / noncanSpl scoring.h:473-487 with the default minanchor (100): intron-length term SimpleFunc::f<int>
H2G_HD int64_t spl_penalty(const DScoring& sc, bool canonical, int intronlen) {
	const double C = canonical ? sc.icpC : sc.incpC, L = canonical ? sc.icpL : sc.incpL;
	const uint32_t T = canonical ? sc.icpT : sc.incpT;
	int pen = 0;
	if(intronlen > 0) {
		const double x = (double)intronlen;
		const double X = T == 1 ? 0.0 : T == 2 ? x : T == 3 ? sqrt(x) : log(x);   // SimpleFunc::f simple_func.h:88-110
		pen = (int)(C + L * X);
	}
	if(pen < 0) pen = 0;
	return (int64_t)pen + (canonical ? sc.cp : sc.ncp);
}

// Read in search orientation: fw -> patFw, !fw -> patRc (Read::constructRevComps read.h:138)
struct SeqView {
	const uint8_t* fwc;
	const char*    q;
	uint32_t       len;
	bool           fw;
	const uint32_t* pk = nullptr;
	uint32_t       pk_stride = 0;
	bool           pk_nomask = false;   // the packed read holds no N: no mask words behind the 2-bit words (h2g_fast.h)
	H2G_HD int at(uint32_t i) const {
		const uint32_t j = fw ? i : len - 1 - i;
		int c;
		if(pk) {
			c = (int)((pk[(j >> 4) * pk_stride] >> ((j & 15) * 2)) & 3u);
			if(!pk_nomask && ((pk[(H2G_PK_WORDS + (j >> 5)) * pk_stride] >> (j & 31)) & 1u)) c = 4;
		} else c = fwc[j];
		if(fw) return c;
		return c < 4 ? 3 - c : 4;
	}
	H2G_HD int qual(uint32_t i) const {   // qual / qualRev
		if(q == nullptr) return 'I';
		return fw ? q[i] : q[len - 1 - i];
	}
	// 32 bases of this strand's view from position i (packed reads without N only), base j at bits 2j .. 2j + 1; what lies past the
	// read's end is not meaningful (the caller masks).  The word-wise comparison loops of h2g_fast.h read the read this way.
	H2G_HD uint64_t pk64(uint32_t w) const {              // packed words w, w + 1 (16 bases each)
		const uint32_t a = w < H2G_PK_WORDS ? pk[w * pk_stride] : 0u, b = w + 1 < H2G_PK_WORDS ? pk[(w + 1) * pk_stride] : 0u;
		return (uint64_t)a | (uint64_t)b << 32;
	}
	H2G_HD uint64_t fwd_chunk32(uint32_t j) const {       // read positions j .. j + 31, forward strand
		const uint32_t w = j >> 4, sh = (j & 15u) * 2u;
		const uint64_t lo = pk64(w);
		if(sh == 0) return lo;
		const uint32_t c = w + 2 < H2G_PK_WORDS ? pk[(w + 2) * pk_stride] : 0u;
		return (lo >> sh) | ((uint64_t)c << (64u - sh));
	}
	H2G_HD uint64_t chunk32(uint32_t i) const {
		if(fw) return fwd_chunk32(i);
		// view position i + j is the complement of read position len - 1 - i - j: the forward chunk ending at p = len - 1 - i, reversed
		const uint32_t p = len - 1 - i;
		uint64_t g = p >= 31 ? fwd_chunk32(p - 31) : fwd_chunk32(0) << ((31u - p) * 2u);
		g = ((g >> 2) & 0x3333333333333333ull) | ((g & 0x3333333333333333ull) << 2);      // reverse the order of the 32 two-bit groups
		g = ((g >> 4) & 0x0f0f0f0f0f0f0f0full) | ((g & 0x0f0f0f0f0f0f0f0full) << 4);
		g = ((g >> 8) & 0x00ff00ff00ff00ffull) | ((g & 0x00ff00ff00ff00ffull) << 8);
		g = ((g >> 16) & 0x0000ffff0000ffffull) | ((g & 0x0000ffff0000ffffull) << 16);
		g = (g >> 32) | (g << 32);
		return ~g;                                         // 3 - c
	}
};

H2G_HD SeqView seq_view(const DReads& r, uint32_t read, bool fw) {
	SeqView s;
	uint32_t a = r.offs[read];
	s.fwc = r.codes + a;
	s.q = r.quals ? r.quals + a : nullptr;
	s.len = r.offs[read + 1] - a;
	s.fw = fw;
	if(r.pk && r.pk_read == read) { s.pk = r.pk; s.pk_stride = r.pk_stride; }
	return s;
}

// ------------------------------------------------------------------------------------------ Occ-rank (a3-a5)
// One linear side = 64 B = 8 x u64: words 0..5 hold 192 symbols (2 bit, LSB-first, pack_2b_in_8b
// gfm.h:4888), word 6 = occ[A] | occ[C] << 32, word 7 = occ[G] | occ[T] << 32 (gfm.h:2953-2957).
struct Side64 { uint64_t w[8]; };

H2G_HD Side64 load_side64(const uint8_t* p) {
	Side64 s;
#if defined(__HIP_DEVICE_COMPILE__)
	const uint4* q = reinterpret_cast<const uint4*>(p);   // 4 x global_load_dwordx4, one 64 B line
	uint4 a = q[0], b = q[1], c = q[2], d = q[3];
	s.w[0] = a.x | ((uint64_t)a.y << 32); s.w[1] = a.z | ((uint64_t)a.w << 32);
	s.w[2] = b.x | ((uint64_t)b.y << 32); s.w[3] = b.z | ((uint64_t)b.w << 32);
	s.w[4] = c.x | ((uint64_t)c.y << 32); s.w[5] = c.z | ((uint64_t)c.w << 32);
	s.w[6] = d.x | ((uint64_t)d.y << 32); s.w[7] = d.z | ((uint64_t)d.w << 32);
#else
	memcpy(s.w, p, 64);
#endif
	return s;
}

// occurrences of symbol c among the first n (0..32) symbols of one u64 — the countInU64 bit trick
// (gfm.h:566-578: x = dw ^ c_table[c]; x & (x >> 1) & 0x55..) with the tail masked instead of shifted
H2G_HD uint32_t count_word(uint64_t w, int c, int n) {
	const uint64_t ct = (c & 2 ? 0ull : 0xaaaaaaaaaaaaaaaaull) | (c & 1 ? 0ull : 0x5555555555555555ull);
	uint64_t x = w ^ ct;
	uint64_t x3 = x & (x >> 1) & 0x5555555555555555ull;
	uint64_t mask = n >= 32 ? ~0ull : ((1ull << (2 * (n < 0 ? 0 : n))) - 1ull);
	return (uint32_t)__builtin_popcountll(x3 & mask);
}

// countBt2Side (gfm.h:2958-3001) on an already loaded linear side; charOff = row % 192
H2G_HD uint32_t rank_in_side64(const DGfm& g, const Side64& s, uint32_t sideNum, uint32_t charOff, int c) {
	uint32_t cnt = 0;
#pragma unroll
	for(int k = 0; k < 6; k++) cnt += count_word(s.w[k], c, (int)charOff - 32 * k);
	// '$' is stored as 'A' (gfm.h:2967-2979; one zOff on a linear index, gfm.h:5400)
	if(c == 0 && g.nZ) {
		uint32_t zs = g.zoff / 192u, zc = g.zoff - zs * 192u;
		if(zs == sideNum && zc < charOff) cnt--;
	}
	const uint64_t ow = (c & 2) ? s.w[7] : s.w[6];       // selects, not dynamic indexing: keeps the side in VGPRs
	const uint32_t occ = (c & 1) ? (uint32_t)(ow >> 32) : (uint32_t)ow;
	const uint32_t fc = c == 0 ? g.fchr[0] : c == 1 ? g.fchr[1] : c == 2 ? g.fchr[2] : g.fchr[3];
	return occ + cnt + fc;
}

H2G_HD int rowL_in_side64(const Side64& s, uint32_t charOff) {   // rowL gfm.h:3615
	const uint32_t k = charOff >> 5;
	const uint64_t a = (k & 1) ? s.w[1] : s.w[0], b = (k & 1) ? s.w[3] : s.w[2], c = (k & 1) ? s.w[5] : s.w[4];
	const uint64_t w = k >= 4 ? c : (k >= 2 ? b : a);
	return (int)((w >> ((charOff & 31) * 2)) & 3);
}

H2G_HD uint32_t rank64(const DGfm& g, uint32_t row, int c) {   // SideLocus::initFromRow gfm.h:376 + mapLF :3712
	uint32_t sideNum = row / 192u, charOff = row - sideNum * 192u;
	Side64 s = load_side64(g.sides + (size_t)sideNum * 64);
	return rank_in_side64(g, s, sideNum, charOff, c);
}

// ------------------------------------------------------------------------------------------ ftab (a10)
H2G_HD uint32_t ftab_hi(const DGfm& g, uint32_t i) {   // gfm.h:2618-2634
	uint32_t v = g.ftab[i];
	if(v <= g.ftabLim) return v;
	return g.eftab[(v ^ H2G_MAX) * 2 + 1];
}
H2G_HD uint32_t ftab_lo(const DGfm& g, uint32_t i) {   // gfm.h:2696-2712
	uint32_t v = g.ftab[i];
	if(v <= g.ftabLim) return v;
	return g.eftab[(v ^ H2G_MAX) * 2];
}

// ------------------------------------------------------------------------------------------ partialSearch (a11)
// hi_aligner.h:6361-6600 for a linear index: mapLF on (tloc,bloc) gfm.h:3739, single-row mapGLF1/mapLF1
// gfm.h:3957-3972 / :3892-3919.  One work item = one (read, strand, offset).
H2G_HD void partial_search_item(const DGfm& g, const SeqView& seq, uint32_t cur_in, bool pseudogeneStopIn,
                                bool anchorStopIn, uint32_t khits, h2g_fm_hit* o)
{
	const uint32_t len = seq.len, ftabLen = g.ftabChars, minK = g.minK;
	bool pseudogeneStop_ = pseudogeneStopIn, anchorStop_ = anchorStopIn;
	bool pseudogeneStop = false, anchorStop = false;
	h2g_fm_hit h;
	h.top = h.bot = h.node_top = h.node_bot = H2G_MAX;
	h.hit_type = H2G_CANDIDATE_HIT;
	h.numPartialSearch = 1; h.numUniqueSearch = 0; h.done = 0; h.nrank = 0; h.nside = 0;
	h.pseudogeneStop = 0; h.anchorStop = 0;
	uint32_t cur = cur_in, offset = cur_in, dep = cur_in;
	h.bwoff = offset;
	const uint32_t left = len - dep;
	bool finished = false;
	if(left < ftabLen + 1) {                       // :6403
		cur = len; h.len = cur - offset; h.done = 1; finished = true;
	}
	uint32_t top = 0, bot = 0;
	if(!finished) {
		uint32_t fi = 0;
		for(uint32_t i = 0; i < ftabLen; i++) {    // N in the ftab window :6419; k-mer packed left-to-right
			int c = seq.at(len - dep - 1 - i);     // (ftabSeqToInt gfm.h:2569, fw index)
			if(c > 3) {
				cur += (i + 1); h.len = cur - offset; if(cur >= len) h.done = 1; finished = true;
				break;
			}
			fi |= (uint32_t)c << (2 * i);
		}
		if(!finished) {
			top = ftab_hi(g, fi);                  // ftabLoHi gfm.h:2670
			bot = ftab_lo(g, fi + 1);
			dep += ftabLen;
			if(top >= bot) {                       // :6442
				cur = dep; h.len = cur - offset; if(cur >= len) h.done = 1; finished = true;
			}
		}
	}
	if(!finished) {
		uint32_t same_range = 0, similar_range = 0, ntop = 0, nbot = 0;
		while(dep < len) {                         // :6459
			int c = seq.at(len - dep - 1);
			uint32_t ttop = 0, tbot = 0;
			if(c <= 3) {
				uint32_t s0 = top / 192u, c0 = top - s0 * 192u;
				Side64 sd = load_side64(g.sides + (size_t)s0 * 64);
				if(bot - top > 1) {                // HIER_INIT_LOCS :5453 -> mapLF(tloc, bloc, c)
					h.nrank += 2;
					ttop = rank_in_side64(g, sd, s0, c0, c);
					uint32_t spread = bot - top;
					if(c0 + spread < 192u) {       // SideLocus::initFromTopBot gfm.h:347-370: same side
						tbot = rank_in_side64(g, sd, s0, c0 + spread, c);
						h.nside += 1;
					} else {
						tbot = rank64(g, bot, c);
						h.nside += 2;
					}
				} else {                           // mapGLF1 -> mapLF1: rowL must equal c, row must not be '$'
					h.nrank += 1; h.nside += 1;
					if(rowL_in_side64(sd, c0) == c && !(g.nZ && top == g.zoff)) {
						ttop = rank_in_side64(g, sd, s0, c0, c);
						tbot = ttop + 1;
					}
				}
			}
			if(ttop >= tbot) break;
			const uint32_t nt = tbot - ttop, no = nbot - ntop;
			if(pseudogeneStop_) {                  // :6488-6503
				if(nt < no && no <= (khits < 5u ? khits : 5u)) {
					if(dep - offset >= minK + 6 && similar_range >= 5) {
						h.numUniqueSearch++; pseudogeneStop = true; break;
					}
				}
				if(nt != 1) {
					if(nt + 2 >= no) similar_range++;
					else if(nt + 4 < no) similar_range = 0;
				} else pseudogeneStop_ = false;
			}
			if(anchorStop_) {                      // :6505-6519
				if(nt != 1 && no == nt) { if(++same_range >= 5) anchorStop_ = false; }
				else same_range = 0;
				if(dep - offset >= minK + 8 && nt >= 4) anchorStop_ = false;
			}
			top = ttop; bot = tbot; ntop = ttop; nbot = tbot;
			dep++;
			if(anchorStop_ && dep - offset >= minK + 12 && bot - top == 1) {   // :6530
				h.numUniqueSearch++; anchorStop = true; break;
			}
		}
		// :6542-6598 (top < bot always holds here)
		uint32_t hit_type = anchorStop ? H2G_ANCHOR_HIT : (pseudogeneStop ? H2G_PSEUDOGENE_HIT : H2G_CANDIDATE_HIT);
		if(ntop < nbot) { h.top = top; h.bot = bot; h.node_top = ntop; h.node_bot = nbot; }
		h.len = dep - offset;
		h.hit_type = hit_type;
		cur = dep;
		if(cur >= len) {
			if(hit_type == H2G_CANDIDATE_HIT) h.numUniqueSearch++;
			h.done = 1;
		}
	}
	h.cur = cur;
	h.pseudogeneStop = pseudogeneStop;
	h.anchorStop = anchorStop;
	*o = h;
}

// ------------------------------------------------------------------------------------------ SA resolve (a14, a15)
// Linear-index meaning of GWState::init/advance (group_walk.h:509-560, 1035-1336) with GFM::tryOffset
// (gfm.h:2719): walk LF until a sampled row ((row & offMask) == row) or the '$' row; off = sample + steps.
H2G_HD uint32_t sa_walk(const DGfm& g, uint32_t row, uint32_t* steps) {
	uint32_t jumps = 0;
	while(true) {
		if(g.nZ && row == g.zoff) break;
		if((row & g.offMask) == row) {
			uint32_t off = g.offs[row >> g.offRate];
			if(off != H2G_MAX) { *steps += jumps; return off + jumps; }
		}
		uint32_t s0 = row / 192u, c0 = row - s0 * 192u;
		Side64 sd = load_side64(g.sides + (size_t)s0 * 64);
		int c = rowL_in_side64(sd, c0);
		row = rank_in_side64(g, sd, s0, c0, c);
		jumps++;
	}
	*steps += jumps;
	return jumps;
}

// GFM::joinedToTextOff gfm.h:5527-5600 (forward index)
H2G_HD bool joined_to_text(const DGfm& g, uint32_t qlen, uint32_t off, uint32_t* tidx, uint32_t* textoff,
                           bool rejectStraddle, bool* straddled)
{
	uint32_t top = 0, bot = g.nFrag, elt = H2G_MAX;
	while(true) {
		uint32_t oldelt = elt;
		elt = top + ((bot - top) >> 1);
		if(oldelt == elt) { *tidx = H2G_MAX; return false; }
		uint32_t lower = g.rstarts[elt * 3];
		uint32_t upper = (elt == g.nFrag - 1) ? g.len : g.rstarts[(elt + 1) * 3];
		if(lower <= off) {
			if(upper > off) {
				if(off + qlen > upper) {
					*straddled = true;
					if(rejectStraddle) { *tidx = H2G_MAX; return false; }
				}
				*tidx = g.rstarts[elt * 3 + 1];
				*textoff = (off - lower) + g.rstarts[elt * 3 + 2];
				return true;
			}
			top = elt;
		} else bot = elt;
	}
}

// GFM::textOffToJoined gfm.h:5603-5651
H2G_HD bool text_off_to_joined(const DGfm& g, uint32_t tid, uint32_t textoff, uint32_t* off) {
	uint32_t top = 0, bot = g.nFrag, elt = H2G_MAX;
	while(true) {
		const uint32_t oldelt = elt;
		elt = top + ((bot - top) >> 1);
		if(oldelt == elt) return false;                              // a text without fragments (the reference asserts)
		const uint32_t elt_tid = g.rstarts[elt * 3 + 1];
		if(elt_tid == tid) {
			while(true) {
				if(tid != g.rstarts[elt * 3 + 1]) return false;
				if(g.rstarts[elt * 3 + 2] <= textoff) break;
				if(elt == 0) return false;
				elt--;
			}
			while(true) {
				if(elt + 1 == g.nFrag || tid + 1 == g.rstarts[(elt + 1) * 3 + 1] || textoff < g.rstarts[(elt + 1) * 3 + 2]) {
					*off = g.rstarts[elt * 3] + (textoff - g.rstarts[elt * 3 + 2]);
					if(elt + 1 < g.nFrag && tid == g.rstarts[(elt + 1) * 3 + 1] && *off >= g.rstarts[(elt + 1) * 3]) return false;
					break;
				}
				elt++;
			}
			return true;
		} else if(elt_tid < tid) top = elt;
		else bot = elt;
	}
}

// HI_Aligner::getGenomeCoords hi_aligner.h:5774-5855
H2G_HD bool genome_coords_item(const DGfm& g, uint32_t top, uint32_t bot, uint32_t maxelt, uint32_t len,
                               bool rejectStraddle, h2g_coord* coords, uint32_t cap, h2g_sa_result* res)
{
	uint32_t nelt = bot - top;
	if(nelt > maxelt) nelt = maxelt;
	if(nelt > cap) nelt = cap;
	res->ok = 1; res->ncoords = 0; res->straddled = 0; res->nsteps = 0;
	for(uint32_t e = 0; e < nelt; e++) {
		uint32_t joff = sa_walk(g, top + e, &res->nsteps);
		uint32_t tidx = 0, toff = 0;
		bool st2 = false;
		joined_to_text(g, len, joff, &tidx, &toff, rejectStraddle, &st2);
		if(st2) res->straddled = 1;
		if(tidx == H2G_MAX) { res->ok = 0; return false; }
		coords[e].tidx = st2 ? H2G_MAX : tidx;
		coords[e].toff = toff;
		coords[e].joinedOff = joff;
		res->ncoords = e + 1;
	}
	return true;
}

// ------------------------------------------------------------------------------------------ reference (a17)
// Meaning of BitPairReference::getStretch (reference.cpp:486-650): base of text `tidx` at `pos`, 4 for an
// ambiguous / out-of-range position.  A cursor caches the record (or gap) interval last resolved, so the
// sequential scans of the extension touch the record table once and then stream the 2-bit payload.
struct RefCursor {
	const DRef* r;
	uint32_t reci, recf;
	int64_t lo, hi;        // cached interval [lo, hi)
	uint32_t bufbase;      // buf offset of lo when inside a record
	bool inrec;
	uint64_t cw;           // 32 cached bases (one aligned 8 B word of the 2-bit payload)
	uint64_t cblk;
	H2G_HD void init(const DRef* r_, uint32_t tidx) {
		r = r_; reci = r->refRecOffs[tidx]; recf = r->refRecOffs[tidx + 1];
		lo = 0; hi = 0; bufbase = 0; inrec = false; cw = 0; cblk = ~0ull;
	}
	H2G_HD void locate(int64_t pos) {
		// last record whose start <= pos
		uint32_t a = reci, b = recf;
		while(a < b) {
			uint32_t m = a + ((b - a) >> 1);
			if((int64_t)r->rec_start[m] <= pos) a = m + 1; else b = m;
		}
		if(a == reci) {   // before the first stretch
			lo = INT64_MIN / 2; hi = (int64_t)r->rec_start[reci]; inrec = false;
			if(reci == recf) hi = INT64_MAX / 2;
			return;
		}
		uint32_t k = a - 1;
		int64_t s = r->rec_start[k], e = s + (int64_t)r->rec_len[k];
		if(pos < e) { lo = s; hi = e; bufbase = r->rec_bufoff[k]; inrec = true; }
		else { lo = e; hi = (k + 1 < recf) ? (int64_t)r->rec_start[k + 1] : INT64_MAX / 2; inrec = false; }
	}
	H2G_HD int get(int64_t pos) {
		if(pos < lo || pos >= hi) locate(pos);
		if(!inrec) return 4;
		const uint64_t bo = (uint64_t)bufbase + (uint64_t)(pos - lo);
		const uint64_t blk = bo >> 5;
		if(blk != cblk) { memcpy(&cw, r->buf + blk * 8, 8); cblk = blk; }   // buf is 256 B-aligned and padded
		return (int)((cw >> ((bo & 31) << 1)) & 3);
	}
	// true: [pos, pos + n) lies inside one stretch of unambiguous bases (the cursor then stands on it)
	H2G_HD bool covers(int64_t pos, uint32_t n) {
		if(pos < lo || pos >= hi) locate(pos);
		return inrec && pos + (int64_t)n <= hi;
	}
	// 32 bases from pos (inside the stretch the cursor stands on), base j at bits 2j .. 2j + 1; bases past the stretch are not meaningful.
	// (buf is padded by 16 bytes: the second word is always readable)
	H2G_HD uint64_t chunk32(int64_t pos) const {
		const uint64_t bo = (uint64_t)bufbase + (uint64_t)(pos - lo);
		const uint64_t blk = bo >> 5;
		const uint32_t sh = (uint32_t)(bo & 31) * 2u;
		uint64_t w0, w1;
		memcpy(&w0, r->buf + blk * 8, 8);
		if(sh == 0) return w0;
		memcpy(&w1, r->buf + blk * 8 + 8, 8);
		return (w0 >> sh) | (w1 << (64u - sh));
	}
};

// ------------------------------------------------------------------------------------------ scoring (a19, a26)
H2G_HD int mm_penalty(const DScoring& sc, int q) {   // Scoring::initPens COST_MODEL_QUAL scoring.h:117-124
	if(q < 0) q = 0;
	int ii = q < 40 ? q : 40;
	float frac = (float)ii / 40.0f;
	return sc.mmpMin + (int)(frac * (float)(sc.mmpMax - sc.mmpMin));
}
H2G_HD int sc_penalty(const DScoring& sc, int q) {   // Scoring::sc scoring.h:312-318
	if(q <= 33) return sc.scMin;
	q -= 33;
	if(q > 40) q = 40;
	return (int)(((float)q / 40.0f) * (float)(sc.scMax - sc.scMin) + (float)sc.scMin);
}

// GenomeHit::calculateScore hi_aligner.h:3711-3891
H2G_HD int64_t calculate_score(const DScoring& sc, const SeqView& seq, h2g_ghit* h) {
	int64_t score = 0;
	double splicescore = 0;
	uint32_t mm = 0, numsplices = 0;
	const uint32_t rdlen = seq.len;
	bool conflict = false;
	uint32_t whichsense = H2G_SPL_UNKNOWN;
	for(uint32_t i = 0; i < h->nedits; i++) {
		const h2g_edit e = h->edits[i];
		if(e.type == H2G_EDIT_SPL) {
			const uint32_t splLen = spl_len(e), splDir = spl_dir(e);
			const bool can = splDir == H2G_SPL_FW || splDir == H2G_SPL_RC;
			if(!spl_known(e)) {
				int left_anchor = (int)(h->rdoff + e.pos), right_anchor = (int)rdlen - left_anchor;
				uint32_t mm2 = 0;
				for(uint32_t j = i + 1; j < h->nedits; j++) { const uint8_t t = h->edits[j].type; if(t == H2G_EDIT_MM || t == H2G_EDIT_READ_GAP || t == H2G_EDIT_REF_GAP) mm2++; }
				left_anchor -= (int)(mm * 2); right_anchor -= (int)(mm2 * 2);
				int shorter = left_anchor < right_anchor ? left_anchor : right_anchor;
				if(shorter <= 0) shorter = 1;
				const uint32_t thresh = can ? max_intron_len((uint32_t)shorter, sc.minAnchorLen) : max_intron_len_noncan((uint32_t)shorter, sc.minAnchorLen_noncan);
				if(thresh < sc.maxIntronLen) {
					if(splLen > thresh) score += INT32_MIN;
					if(can) {
						const float probscore = spl_prob(e);
						float pt = 0.8f;
						if(splLen >> 16) pt = 0.99f; else if(splLen >> 15) pt = 0.97f; else if(splLen >> 14) pt = 0.94f; else if(splLen >> 13) pt = 0.91f; else if(splLen >> 12) pt = 0.88f;
						if(probscore < pt) score += INT32_MIN;
					}
					if(shorter == left_anchor) {
						if(h->trim5 > 0) score += INT32_MIN;
						for(int j = (int)i - 1; j >= 0; j--) { const uint8_t t = h->edits[j].type; if(t == H2G_EDIT_MM || t == H2G_EDIT_READ_GAP || t == H2G_EDIT_REF_GAP) score += INT32_MIN; }
					} else {
						if(h->trim3 > 0) score += INT32_MIN;
						for(uint32_t j = i + 1; j < h->nedits; j++) { const uint8_t t = h->edits[j].type; if(t == H2G_EDIT_MM || t == H2G_EDIT_READ_GAP || t == H2G_EDIT_REF_GAP) score += INT32_MIN; }
					}
				}
				score -= spl_penalty(sc, can, (int)splLen);
				if(shorter <= 15) { numsplices++; splicescore += (double)splLen; }
			}
			if(!conflict) {
				if(whichsense == H2G_SPL_UNKNOWN) whichsense = splDir;
				else if(splDir != H2G_SPL_UNKNOWN) {
					if((splDir == H2G_SPL_FW || splDir == H2G_SPL_SEMI_FW) && whichsense != H2G_SPL_FW && whichsense != H2G_SPL_SEMI_FW) conflict = true;
					if((splDir == H2G_SPL_RC || splDir == H2G_SPL_SEMI_RC) && whichsense != H2G_SPL_RC && whichsense != H2G_SPL_SEMI_RC) conflict = true;
				}
			}
			continue;
		}
		if(e.type == H2G_EDIT_MM) {
			if(e.snp != H2G_MAX) continue;                 // edits through known variants cost nothing (:3737, :3846, :3858)
			int q = seq.qual(h->rdoff + e.pos) - 33;
			if(e.qchr == 'N') score -= sc.nPen;            // Scoring::score scoring.h:259-269: rdc > 3
			else if(e.chr == 'N') score += sc.matchBonus;  // ref mask 15 contains every base
			else score -= mm_penalty(sc, q);
			mm++;
		} else if(e.type == H2G_EDIT_READ_GAP) {
			if(e.snp != H2G_MAX) continue;
			bool open = !(i > 0 && h->edits[i - 1].type == H2G_EDIT_READ_GAP && h->edits[i - 1].pos == e.pos);
			score -= open ? (sc.rdGapConst + sc.rdGapLinear) : sc.rdGapLinear;
		} else if(e.type == H2G_EDIT_REF_GAP) {
			if(e.snp != H2G_MAX) continue;
			bool open = !(i > 0 && h->edits[i - 1].type == H2G_EDIT_REF_GAP && h->edits[i - 1].pos + 1 == e.pos);
			score -= open ? (sc.rfGapConst + sc.rfGapLinear) : sc.rfGapLinear;
		}
	}
	for(uint32_t i = 0; i < h->trim5; i++) score -= sc_penalty(sc, seq.qual(i));   // :3868-3874 (qual[i] both times)
	for(uint32_t i = 0; i < h->trim3; i++) score -= sc_penalty(sc, seq.qual(i));
	if(conflict) score -= sc.csp;
	if(numsplices > 1) splicescore /= (double)numsplices;
	score += (int64_t)(h->len - mm) * sc.matchBonus;
	h->score = score;
	h->splicescore = (uint32_t)(int64_t)splicescore;       // AlnScore keeps it as a TAlScore (truncated)
	return score;
}

// ------------------------------------------------------------------------------------------ extend (a18)
#ifndef H2G_NEW_EDITS           // edits ONE extension may add (the *_big units raise it with H2G_GHIT_EDITS: h2g_go_big.h)
#define H2G_NEW_EDITS 24
#endif
H2G_HD uint8_t base_char(int c) { return (uint8_t)("ACGTN"[c]); }
H2G_HD bool is_gap(uint8_t t) { return t == H2G_EDIT_READ_GAP || t == H2G_EDIT_REF_GAP; }
// the edits getLeft / getRight / combineWith stop at: gaps and mismatches through a known SNP (hi_aligner.h:937-940, :981-984)
H2G_HD bool is_stop_edit(const h2g_edit& e) { return e.type == H2G_EDIT_SPL || is_gap(e.type) || (e.type == H2G_EDIT_MM && e.snp != H2G_MAX); }

// alignWithALTs (hi_aligner.h:683-783) over alignWithALTs_recur without ALTs (:2763-2853 left,
// :3168-3216 right).  Edits are committed in place instead of through a scratch copy.
H2G_HD uint32_t align_no_alts(const DRef& ref, const SeqView& seq, uint32_t base_rdoff, uint32_t rdoff,
                              uint32_t rdlen, uint32_t tidx, int rfoff, uint32_t rflen, bool left, h2g_ghit* h,
                              uint32_t mm, uint32_t* numNs)
{
	if(numNs) *numNs = 0;
	const uint32_t n_old = h->nedits;
	h2g_edit ne[H2G_NEW_EDITS];
	uint32_t tmp_mm = 0, nNs = 0;
	bool updated = false;
	uint32_t extlen = 0;
	const uint32_t contig_len = ref.refLens[tidx];
	bool run = !(rfoff < -16) && !((int64_t)rfoff >= (int64_t)contig_len);
	if(run) {
		if(rfoff >= 0 && (uint64_t)rfoff + rflen > contig_len) rflen = contig_len - (uint32_t)rfoff;
		else if(rfoff < 0 && rflen > contig_len) rflen = contig_len;
		if(rflen == 0) run = false;
	}
	if(run) {
		RefCursor rc;
		rc.init(&ref, tidx);
		const uint32_t rdoff_add = rdoff - base_rdoff;
		if(left) {
			int i = (int)rdoff;
			for(int rf_i = (int)rflen - 1; rf_i >= 0 && i >= 0; rf_i--, i--) {
				int64_t p = (int64_t)rfoff + rf_i;
				int rf_bp = p < 0 ? 4 : rc.get(p), rd_bp = seq.at((uint32_t)i);
				if(rf_bp != rd_bp || rd_bp == 4) {
					if(tmp_mm >= mm) break;
					if(tmp_mm < H2G_NEW_EDITS) {
						ne[tmp_mm].pos = (uint32_t)i; ne[tmp_mm].chr = base_char(rf_bp); ne[tmp_mm].qchr = base_char(rd_bp);
						ne[tmp_mm].type = H2G_EDIT_MM; ne[tmp_mm].pad = 0; ne[tmp_mm].snp = H2G_MAX;
					} else h->overflow = 1;
					tmp_mm++;
				}
				if(rf_bp == 4) nNs++;
			}
			if(i < (int)rdoff) { updated = true; extlen = rdoff - (uint32_t)i; if(numNs) *numNs = nNs; }
		} else {
			uint32_t i = 0;
			for(uint32_t rf_i = 0; rf_i < rflen && i < rdlen; rf_i++, i++) {
				int64_t p = (int64_t)rfoff + rf_i;
				int rf_bp = p < 0 ? 4 : rc.get(p), rd_bp = seq.at(rdoff + i);
				if(rf_bp != rd_bp || rd_bp == 4) {
					if(tmp_mm >= mm) break;
					if(tmp_mm < H2G_NEW_EDITS) {
						ne[tmp_mm].pos = i + rdoff_add; ne[tmp_mm].chr = base_char(rf_bp); ne[tmp_mm].qchr = base_char(rd_bp);
						ne[tmp_mm].type = H2G_EDIT_MM; ne[tmp_mm].pad = 0; ne[tmp_mm].snp = H2G_MAX;
					} else h->overflow = 1;
					tmp_mm++;
				}
			}
			if(i > 0) { updated = true; extlen = i; }
		}
	}
	if(!updated) tmp_mm = 0;
	if(tmp_mm > H2G_NEW_EDITS) tmp_mm = H2G_NEW_EDITS;
	const uint32_t total = n_old + tmp_mm;
	if(extlen > 0 && total > 0) {   // :751-779
		// front()/back() of the list the reference would hold after `edits = tmp_edits`
		h2g_edit f, b;
		if(left) { f = tmp_mm ? ne[tmp_mm - 1] : h->edits[0]; b = n_old ? h->edits[n_old - 1] : ne[0]; }
		else     { f = n_old ? h->edits[0] : ne[0];            b = tmp_mm ? ne[tmp_mm - 1] : h->edits[n_old - 1]; }
		if(f.pos + extlen == base_rdoff + 1) {
			if(is_gap(f.type) || f.type == H2G_EDIT_SPL) extlen = 0;
			if(f.type == H2G_EDIT_MM && f.chr == 'N') extlen = 0;
		}
		if(extlen > 0 && b.pos == rdoff - base_rdoff + extlen - 1) {
			if(is_gap(b.type)) extlen = 0;   // the back test covers gaps only (:769-773)
		}
	}
	if(extlen > 0 && tmp_mm > 0) {   // commit the new edits
		if(total > H2G_GHIT_EDITS) { h->overflow = 1; return extlen; }
		if(left) {                   // new edits go to the front, in increasing read position
			for(int k = (int)n_old - 1; k >= 0; k--) h->edits[k + tmp_mm] = h->edits[k];
			for(uint32_t k = 0; k < tmp_mm; k++) h->edits[k] = ne[tmp_mm - 1 - k];
		} else {
			for(uint32_t k = 0; k < tmp_mm; k++) h->edits[n_old + k] = ne[k];
		}
		h->nedits = total;
	}
	if(extlen == 0 && numNs) *numNs = updated ? nNs : 0;
	return extlen;
}

// GenomeHit::getRight hi_aligner.h:962-1000 (+ getRightOff :1020) for MM / gap edit lists
H2G_HD void hit_get_right(const h2g_ghit* h, uint32_t* rdoff, uint32_t* len, uint32_t* toff) {
	*rdoff = h->rdoff; *len = h->len; *toff = h->toff;
	for(int i = (int)h->nedits - 1; i >= 0; i--) {
		const h2g_edit e = h->edits[i];
		if(is_stop_edit(e)) {
			*rdoff = h->rdoff + e.pos;
			*len = h->len - e.pos;
			if(e.type == H2G_EDIT_REF_GAP || e.type == H2G_EDIT_MM) { (*rdoff)++; (*len)--; }
			uint32_t roff = h->toff + h->len;
			for(uint32_t k = 0; k < h->nedits; k++) {
				if(h->edits[k].type == H2G_EDIT_READ_GAP) roff++;
				else if(h->edits[k].type == H2G_EDIT_REF_GAP) roff--;
				else if(h->edits[k].type == H2G_EDIT_SPL) roff += spl_len(h->edits[k]);
			}
			*toff = roff - *len;
			return;
		}
	}
}

// GenomeHit::extend hi_aligner.h:2031-2232
H2G_HD bool extend_item(const DRef& ref, const DScoring& sc, const SeqView& seq, h2g_ghit* h, uint32_t mm,
                        uint32_t max_leftext, uint32_t max_rightext, uint32_t* leftext, uint32_t* rightext)
{
	const uint32_t rdlen = seq.len;
	*leftext = 0; *rightext = 0;
	if(max_leftext > 0 && h->rdoff > 0) {
		if(h->toff <= 0) return false;
		int rl = (int)h->toff - (int)h->rdoff;
		uint32_t reflen = h->rdoff + 10;
		rl -= (int)(reflen - h->rdoff);
		if(rl < 0) { reflen += rl; rl = 0; }
		uint32_t numNs = 0;
		const uint32_t n_prev = h->nedits;
		uint32_t best_ext = align_no_alts(ref, seq, h->rdoff - 1, h->rdoff - 1, h->rdoff, h->tidx, rl, reflen, true,
		                                  h, mm, &numNs);
		if(h->len == 0 && mm == 0 && h->nedits > 0) { h->nedits = 0; return false; }
		if(best_ext > 0) {
			*leftext = best_ext;
			const uint32_t added = h->nedits - n_prev;
			int ref_ext = (int)best_ext;
			for(uint32_t i = 0; i < added; i++) {
				if(h->edits[i].type == H2G_EDIT_REF_GAP) ref_ext--;
				else if(h->edits[i].type == H2G_EDIT_READ_GAP) ref_ext++;
				else if(h->edits[i].type == H2G_EDIT_SPL) ref_ext += (int)spl_len(h->edits[i]);
			}
			h->rdoff -= best_ext;
			h->toff -= (uint32_t)ref_ext;
			h->len += best_ext;
			h->joinedOff -= (uint32_t)(ref_ext - (int)numNs);
			for(uint32_t i = 0; i < h->nedits; i++) {
				if(i < added) h->edits[i].pos -= h->rdoff;
				else h->edits[i].pos += best_ext;
			}
		}
	}
	if(max_rightext > 0 && h->rdoff + h->len < rdlen) {
		uint32_t r_rdoff, r_len, r_toff;
		hit_get_right(h, &r_rdoff, &r_len, &r_toff);
		const uint32_t rl = r_toff + r_len;
		const uint32_t rr = rdlen - (r_rdoff + r_len);
		const uint32_t tlen = ref.refLens[h->tidx];
		if(rl < tlen) {
			uint32_t reflen = rr + 10;
			if(rl + reflen > tlen) reflen = tlen - rl;
			uint32_t best_ext = align_no_alts(ref, seq, h->rdoff, h->rdoff + h->len, rdlen - (h->rdoff + h->len),
			                                  h->tidx, (int)rl, reflen, false, h, mm, nullptr);
			if(h->len == 0 && mm == 0 && h->nedits > 0) { h->nedits = 0; return false; }
			if(best_ext > 0) { *rightext = best_ext; h->len += best_ext; }
		}
	}
	calculate_score(sc, seq, h);
	return *leftext > 0 || *rightext > 0;
}

// ------------------------------------------------------------------------------------------ fused seed stage
// One (read, strand): coordinates of the partial hit (getAnchorHits hi_aligner.h:5007: only hits longer
// than minK + 2; here the first H2G_SEED_CAP rows of the range) and their 0-mismatch extension
// (hybridSearch spliced_aligner.h:139-163).
H2G_HD void resolve_extend_item(const DGfm& g, const DRef& ref, const DScoring& sc, const SeqView& seq,
                                h2g_seed_result* out, h2g_ghit* scratch)
{
	const h2g_fm_hit hit = out->hit;
	out->ncoords = 0; out->straddled = 0; out->nsteps = 0; out->pad = 0;
	if(hit.top == H2G_MAX || hit.bot <= hit.top || hit.len <= g.minK + 2) return;
	h2g_coord co[H2G_SEED_CAP];
	h2g_sa_result res;
	genome_coords_item(g, hit.top, hit.bot, hit.bot - hit.top, hit.len, false, co, H2G_SEED_CAP, &res);
	out->ncoords = res.ncoords; out->straddled = res.straddled; out->nsteps = res.nsteps;
	for(uint32_t k = 0; k < res.ncoords; k++) {
		out->ext[k].tidx = co[k].tidx; out->ext[k].toff = co[k].toff; out->ext[k].joinedOff = co[k].joinedOff;
		out->ext[k].rdoff = seq.len - hit.bwoff - hit.len; out->ext[k].len = hit.len; out->ext[k].score = 0;
		if(co[k].tidx == H2G_MAX) continue;
		h2g_ghit* h = scratch;
		h->read = 0; h->fw = seq.fw; h->rdoff = seq.len - hit.bwoff - hit.len; h->len = hit.len; h->trim5 = 0; h->trim3 = 0;
		h->tidx = co[k].tidx; h->toff = co[k].toff; h->joinedOff = co[k].joinedOff; h->score = 0; h->nedits = 0;
		h->overflow = 0;
		uint32_t le, re;
		extend_item(ref, sc, seq, h, 0, H2G_MAX, H2G_MAX, &le, &re);
		out->ext[k].toff = h->toff; out->ext[k].joinedOff = h->joinedOff; out->ext[k].rdoff = h->rdoff;
		out->ext[k].len = h->len; out->ext[k].score = (int32_t)h->score;
	}
}

}  // namespace h2g
