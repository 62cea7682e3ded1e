// h2g_graph.h — graph (GFM) index primitives: the 128 B side rank, rank over M, select over F, graph LF
// (SURVEY §8 rows a2, a7, a9).  Same convention as h2g_core.h: `__host__ __device__` per-item functions,
// wrapped by kernels in h2g_kernels.hip and host-instantiated only by tests/emul.
//
// Graph side layout (gfm.h:160-176; written by GFM::buildToDisk gfm.h:4990-5290), lineRate 7, index_t = u32:
//   bytes [  0, 52)  208 gbwt symbols, 2 bit each, LSB first
//   bytes [ 52, 78)  F bits  (1 = first incoming row of a node), bit j of byte i = position 8 i + j
//   bytes [ 78,104)  M bits  (1 = last outgoing row of a node)
//   bytes [104,128)  u32 F_loc, M_occ, occ[A], occ[C], occ[G], occ[T]
// so a side is exactly one 128 B L2 line of MI355X (MI355X_MICROARCH.md: 128 B lines) — the reason the
// graph rank reaches a higher fraction of HBM peak than the 64 B linear side.
#pragma once
#include <vector>
#include "h2g_core.h"

namespace h2g {

#define H2G_GSIDE_SYMS 208u

// Index policy ("X" below): what the graph functions need from an index.  DGfm is the policy of the global index
// (u32 words, 208 symbols per side); LGfm (further down) that of a local index (u16 words, 232 symbols per side).
//   X::SYMS  symbols per side          X::NCW  u64 words holding them        X::F_OFF / M_OFF  byte offsets of the bit vectors
//   X::HDR   byte offset of {F_loc, M_occ, occ[4]}      X::WSZ  word size of those six fields
//   members: sides, nZ, zoff, zoffs, fchr[5], gbwtLen, offMask, offRate and offs_at(i)
struct Side128 { uint64_t w[16]; };

H2G_HD Side128 load_side128(const uint8_t* p) {
	Side128 s;
#if defined(__HIP_DEVICE_COMPILE__)
	const uint4* q = reinterpret_cast<const uint4*>(p);   // 8 x global_load_dwordx4, one 128 B line
#pragma unroll
	for(int k = 0; k < 8; k++) {
		uint4 a = q[k];
		s.w[2 * k] = a.x | ((uint64_t)a.y << 32);
		s.w[2 * k + 1] = a.z | ((uint64_t)a.w << 32);
	}
#else
	memcpy(s.w, p, 128);
#endif
	return s;
}

// graph LF leaf functions: out of line by default (bounded code size / compile time); the single-end graph unit inlines them
// into its search loops (H2G_INLINE_GLF: -5 % there, +5 % in the paired unit whose register budget is tighter)
#ifdef H2G_INLINE_GLF
#define H2G_GLF H2G_HD
#else
#define H2G_GLF H2G_HDN
#endif

template <class X>
H2G_HD bool is_zoff(const X& g, uint32_t row) {   // GFM::_zOffs (gfm.h:2783); a handful of entries at most
	if(g.nZ == 0) return false;
	if(row == g.zoff) return true;
	for(uint32_t i = 1; i < g.nZ; i++) if(g.zoffs[i] == row) return true;
	return false;
}

// occ[c] of a side held in registers — selects, no dynamic indexing of the register array (that would go to scratch)
template <class X>
H2G_HD uint32_t side_occ(const Side128& s, int c) {
	if(X::WSZ == 4) {                                   // u32 occ[4] at byte 112: words 14, 15
		const uint64_t ow = (c & 2) ? s.w[15] : s.w[14];
		return (c & 1) ? (uint32_t)(ow >> 32) : (uint32_t)ow;
	}
	return (uint32_t)((s.w[15] >> (16 * c)) & 0xffffu);   // u16 occ[4] at byte 120: word 15
}
template <class X>
H2G_HD uint32_t side_hdr_mem(const uint8_t* side, int k) {   // same, straight from memory
	if(X::WSZ == 4) return reinterpret_cast<const uint32_t*>(side + X::HDR)[k];
	return reinterpret_cast<const uint16_t*>(side + X::HDR)[k];
}

// countBt2Side (gfm.h:2958-3001) on a loaded graph side; charOff = row % SYMS
template <class X>
H2G_HD uint32_t rank_in_side128(const X& g, const Side128& s, uint32_t sideNum, uint32_t charOff, int c) {
	uint32_t cnt = 0;
#pragma unroll
	for(int k = 0; k < (int)X::NCW; k++) cnt += count_word(s.w[k], c, (int)charOff - 32 * k);   // count_word masks beyond charOff
	if(c == 0 && g.nZ) {                                  // '$' rows are stored as 'A' (gfm.h:2967-2979)
		for(uint32_t i = 0; i < g.nZ; i++) {
			const uint32_t z = i == 0 ? g.zoff : g.zoffs[i];
			const uint32_t zs = z / X::SYMS, zc = z - zs * X::SYMS;
			if(zs == sideNum && zc < charOff) cnt--;
		}
	}
	const uint32_t occ = side_occ<X>(s, c);
	const uint32_t fc = c == 0 ? g.fchr[0] : c == 1 ? g.fchr[1] : c == 2 ? g.fchr[2] : g.fchr[3];
	return occ + cnt + fc;
}

H2G_HD int rowL_in_side128(const Side128& s, uint32_t charOff) {   // rowL gfm.h:3615
	const uint32_t k = charOff >> 5;
	uint64_t w = s.w[0];
#pragma unroll
	for(int j = 1; j < 8; j++) w = (k == (uint32_t)j) ? s.w[j] : w;
	return (int)((w >> ((charOff & 31) * 2)) & 3);
}

template <class X>
H2G_HD uint32_t rank128(const X& g, uint32_t row, int c) {      // SideLocus::initFromRow + mapLF gfm.h:3712
	const uint32_t sideNum = row / X::SYMS, charOff = row - sideNum * X::SYMS;
	Side128 s = load_side128(g.sides + (size_t)sideNum * 128);
	return rank_in_side128(g, s, sideNum, charOff, c);
}

// ---- bit vectors ------------------------------------------------------------------------------------------
// F and M start at arbitrary byte offsets of the side (52 / 78 global, 58 / 87 local); they are read with aligned
// dword loads and funnel shifts.  SYMS bits = 3 full u64 + 16 (global) or 40 (local) bits.
struct Bits256 { uint64_t w[4]; };
H2G_HD uint64_t ld64_at(const uint8_t* side, uint32_t byte) {   // 8 bytes at any offset inside the 128 B side
	const uint32_t* q = reinterpret_cast<const uint32_t*>(side + (byte & ~3u));
	const uint32_t sh = (byte & 3u) * 8;
	const uint64_t lo = q[0] | ((uint64_t)q[1] << 32);
	if(sh == 0) return lo;
	return (lo >> sh) | ((uint64_t)q[2] << (64 - sh));
}
template <class X>
H2G_HD Bits256 load_bits(const uint8_t* side, uint32_t off) {
	Bits256 b;
	b.w[0] = ld64_at(side, off); b.w[1] = ld64_at(side, off + 8); b.w[2] = ld64_at(side, off + 16);
	// the last word may start within 8 bytes of the end of the side: read what is there, keep SYMS - 192 bits
	const uint32_t last = off + 24;
	uint64_t v = 0;
	for(uint32_t k = 0; k < (X::SYMS - 192 + 7) / 8; k++) v |= (uint64_t)side[last + k] << (8 * k);
	b.w[3] = v & ((1ull << (X::SYMS - 192)) - 1ull);
	return b;
}
H2G_HD uint64_t low_mask(int n) { return n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull)); }

// rank_M (gfm.h:4100) = countMSide (:3146): ones of M strictly before `row`, plus the side's M_occ
template <class X>
H2G_HD uint32_t rank_M(const X& g, uint32_t row) {
	const uint32_t sideNum = row / X::SYMS, off = row - sideNum * X::SYMS;
	const uint8_t* side = g.sides + (size_t)sideNum * 128;
	Bits256 m = load_bits<X>(side, X::M_OFF);
	uint32_t cnt = side_hdr_mem<X>(side, 1);
#pragma unroll
	for(int k = 0; k < 4; k++) cnt += (uint32_t)__builtin_popcountll(m.w[k] & low_mask((int)off - 64 * k));
	return cnt;
}

H2G_HD uint32_t select_in_word(uint64_t w, uint32_t count) {   // position of the count-th (1-based) set bit
	for(uint32_t i = 1; i < count; i++) w &= w - 1;
	return (uint32_t)__builtin_ctzll(w);
}

// select_F (gfm.h:4113-4167): row of the count-th F one at or after `row` (count >= 1), crossing sides as needed
template <class X>
H2G_GLF uint32_t select_F(const X& g, uint32_t row, uint32_t count) {
	uint32_t sideNum = row / X::SYMS, off = row - sideNum * X::SYMS;
	const uint32_t lastSide = (g.gbwtLen - 1) / X::SYMS;
	while(true) {
		Bits256 f = load_bits<X>(g.sides + (size_t)sideNum * 128, X::F_OFF);
#pragma unroll
		for(int k = 0; k < 4; k++) {
			const uint64_t w = f.w[k] & ~low_mask((int)off - 64 * k);
			const uint32_t pc = (uint32_t)__builtin_popcountll(w);
			if(count <= pc) return sideNum * X::SYMS + 64u * k + select_in_word(w, count);
			count -= pc;
		}
		if(sideNum >= lastSide) return g.gbwtLen;   // not reachable on a well-formed index (the reference would run off the array)
		sideNum++;
		off = 0;
	}
}

// F-row of node `node`: backward scan over the (F_loc, M_occ) side headers starting at the side of `locRow`
// (mapGLF gfm.h:3788-3810, mapGLF1 :3978-3998).  Returns the scan's F_loc (already +1 when M_occ > 0) and M_occ.
template <class X>
H2G_GLF uint32_t node_to_Frow(const X& g, uint32_t locRow, uint32_t node, uint32_t* F_loc_out, uint32_t* M_occ_out) {
	uint32_t sideNum = locRow / X::SYMS;
	uint32_t F_loc, M_occ;
	while(true) {
		const uint8_t* sd = g.sides + (size_t)sideNum * 128;
		F_loc = side_hdr_mem<X>(sd, 0);
		M_occ = side_hdr_mem<X>(sd, 1);
		if(M_occ <= node || sideNum == 0) break;
		sideNum--;
	}
	if(M_occ > 0) F_loc++;
	*F_loc_out = F_loc;
	*M_occ_out = M_occ;
	if(node + 1 > M_occ) return select_F(g, F_loc, node + 1 - M_occ);
	return F_loc;
}

// BWTHit::_node_iedge_count (hi_aligner.h:199): nodes of a range that have more than one incoming edge
typedef h2g_iedges IEdges;            // n = true count; entries beyond H2G_IEDGE_CAP are dropped (caller checks n)

// getInEdgeCount (gfm.h:4172-4213)
template <class X>
H2G_GLF void in_edge_count(const X& g, uint32_t top, uint32_t bot, IEdges* ie) {
	ie->n = 0;
	uint32_t curr_node = 0, num0s = 0;
	uint32_t sideNum = H2G_MAX;
	Bits256 f;
	f.w[0] = f.w[1] = f.w[2] = f.w[3] = 0;
	for(uint32_t row = top + 1; row < bot; row++) {
		const uint32_t sn = row / X::SYMS, off = row - sn * X::SYMS;
		if(sn != sideNum) { sideNum = sn; f = load_bits<X>(g.sides + (size_t)sn * 128, X::F_OFF); }
		const uint32_t k = off >> 6;
		const uint64_t w = k == 0 ? f.w[0] : (k == 1 ? f.w[1] : (k == 2 ? f.w[2] : f.w[3]));
		if((w >> (off & 63)) & 1) { curr_node++; num0s = 0; }
		else {
			num0s++;
			if(num0s == 1) { if(ie->n < H2G_IEDGE_CAP) ie->e[ie->n][0] = curr_node; ie->n++; }
			if(ie->n <= H2G_IEDGE_CAP) ie->e[ie->n - 1][1] = num0s;
		}
	}
}

struct GRange { uint32_t top, bot, node_top, node_bot; };

// ---- one LF step of ONE row with the bit vectors taken from sides held in registers (round 4) -------------------------------
// mapGLF1 without a required character (gfm.h:4029-4095), reduced to what a single-row walk needs: the incoming row and the node
// the step arrives at.  Equal to map_glf1_nochar's (top, node_top) — the host test h2gemu_glf_fused_check holds it to that — but
// with one 128 B load per side touched (the row's side; the side of t + 1 for its M bits, F_loc and M_occ; the side select_F lands
// in when that is another one) instead of a dependent load per field: 2-3 latencies per step instead of 5-6.
template <class X>
H2G_HD Bits256 bits_of_side(const Side128& s, uint32_t off) {       // load_bits over a side in registers
	const uint32_t i = off >> 3, sh = (off & 7u) * 8u;
	Bits256 b;
#pragma unroll
	for(uint32_t k = 0; k < 4; k++) {
		const uint64_t lo = s.w[i + k] >> sh;
		const uint64_t hi = sh ? (s.w[i + k + 1] << (64u - sh)) : 0ull;
		b.w[k] = lo | hi;
	}
	b.w[3] &= (1ull << (X::SYMS - 192)) - 1ull;
	return b;
}
template <class X> H2G_HD uint32_t side_hdr_reg(const Side128& s, int k) {   // {F_loc, M_occ} of a side in registers (side_hdr_mem)
	if(X::WSZ == 4) return k == 0 ? (uint32_t)s.w[13] : (uint32_t)(s.w[13] >> 32);           // u32 at byte 104 / 108
	return k == 0 ? (uint32_t)((s.w[14] >> 32) & 0xffffu) : (uint32_t)(s.w[14] >> 48);      // u16 at byte 116 / 118
}
// the c1-th and the c2-th F one at or after row F_loc (select_F twice over one scan; a count of 0 stands for "F_loc itself", as mapGLF1 uses
// it), starting in registers when F_loc lies in the side `cur` holds (side number *sideNum)
template <class X>
H2G_HD void select_F2_reg(const X& g, Side128& cur, uint32_t& sideNum, uint32_t F_loc, uint32_t c1, uint32_t c2, uint32_t* p1, uint32_t* p2) {
	*p1 = c1 == 0 ? F_loc : g.gbwtLen;
	*p2 = c2 == 0 ? F_loc : g.gbwtLen;
	if(c1 == 0 && c2 == 0) return;
	uint32_t fs = F_loc / X::SYMS, off = F_loc - fs * X::SYMS;
	const uint32_t lastSide = (g.gbwtLen - 1) / X::SYMS;
	while(true) {
		if(fs != sideNum) { cur = load_side128(g.sides + (size_t)fs * 128); sideNum = fs; }
		const Bits256 f = bits_of_side<X>(cur, X::F_OFF);
#pragma unroll
		for(int k = 0; k < 4; k++) {
			const uint64_t w = f.w[k] & ~low_mask((int)off - 64 * k);
			const uint32_t pc = (uint32_t)__builtin_popcountll(w);
			if(c1 > 0) { if(c1 <= pc) { *p1 = fs * X::SYMS + 64u * k + select_in_word(w, c1); c1 = 0; } else c1 -= pc; }
			if(c2 > 0) { if(c2 <= pc) { *p2 = fs * X::SYMS + 64u * k + select_in_word(w, c2); c2 = 0; } else c2 -= pc; }
		}
		if((c1 == 0 && c2 == 0) || fs >= lastSide) return;
		fs++;
		off = 0;
	}
}
// mapGLF1 with a required character (gfm.h:3957-4021; map_glf1) from sides held in registers: the searches' step once a range is one row
template <class X>
H2G_HD bool map_glf1_fused(const X& g, uint32_t row, int c, GRange* r) {
	r->top = r->bot = r->node_top = r->node_bot = 0;
	const uint32_t s0 = row / X::SYMS, c0 = row - s0 * X::SYMS;
	const Side128 sd = load_side128(g.sides + (size_t)s0 * 128);
	if(rowL_in_side128(sd, c0) != c || is_zoff(g, row)) return false;
	const uint32_t t = rank_in_side128(g, sd, s0, c0, c);
	const uint32_t r1 = t + 1, s1 = r1 / X::SYMS, o1 = r1 - s1 * X::SYMS;
	Side128 cur = sd;
	if(s1 != s0) cur = load_side128(g.sides + (size_t)s1 * 128);
	uint32_t node;
	{
		const Bits256 m = bits_of_side<X>(cur, X::M_OFF);
		uint32_t cnt = side_hdr_reg<X>(cur, 1);
#pragma unroll
		for(int k = 0; k < 4; k++) cnt += (uint32_t)__builtin_popcountll(m.w[k] & low_mask((int)o1 - 64 * k));
		node = cnt - 1;
	}
	uint32_t sideNum = s1, F_loc = side_hdr_reg<X>(cur, 0), M_occ = side_hdr_reg<X>(cur, 1);
	while(!(M_occ <= node || sideNum == 0)) {
		sideNum--;
		cur = load_side128(g.sides + (size_t)sideNum * 128);
		F_loc = side_hdr_reg<X>(cur, 0); M_occ = side_hdr_reg<X>(cur, 1);
	}
	if(M_occ > 0) F_loc++;
	const uint32_t node_bot = node + 1;
	uint32_t ft, fb;
	select_F2_reg(g, cur, sideNum, F_loc, node + 1 > M_occ ? node + 1 - M_occ : 0u, node_bot + 1 > M_occ ? node_bot + 1 - M_occ : 0u, &ft, &fb);
	r->top = ft; r->bot = fb; r->node_top = node; r->node_bot = node_bot;
	return true;
}
template <class X>
H2G_HD void glf1_top_fused(const X& g, uint32_t row, uint32_t* top_out, uint32_t* node_out) {
	const uint32_t s0 = row / X::SYMS, c0 = row - s0 * X::SYMS;
	const Side128 sd = load_side128(g.sides + (size_t)s0 * 128);
	const int c = rowL_in_side128(sd, c0);
	const uint32_t t = rank_in_side128(g, sd, s0, c0, c);
	// rank_M(t + 1) - 1
	const uint32_t r1 = t + 1, s1 = r1 / X::SYMS, o1 = r1 - s1 * X::SYMS;
	Side128 cur = sd;
	if(s1 != s0) cur = load_side128(g.sides + (size_t)s1 * 128);
	uint32_t node;
	{
		const Bits256 m = bits_of_side<X>(cur, X::M_OFF);
		uint32_t cnt = side_hdr_reg<X>(cur, 1);
#pragma unroll
		for(int k = 0; k < 4; k++) cnt += (uint32_t)__builtin_popcountll(m.w[k] & low_mask((int)o1 - 64 * k));
		node = cnt - 1;
	}
	// node_to_Frow(t + 1, node): backward scan over the side headers
	uint32_t sideNum = s1, F_loc = side_hdr_reg<X>(cur, 0), M_occ = side_hdr_reg<X>(cur, 1);
	while(!(M_occ <= node || sideNum == 0)) {
		sideNum--;
		cur = load_side128(g.sides + (size_t)sideNum * 128);
		F_loc = side_hdr_reg<X>(cur, 0); M_occ = side_hdr_reg<X>(cur, 1);
	}
	if(M_occ > 0) F_loc++;
	uint32_t ft = F_loc;
	if(node + 1 > M_occ) {                                   // select_F(F_loc, node + 1 - M_occ), starting in registers when F_loc lies in `cur`'s side
		uint32_t count = node + 1 - M_occ;
		uint32_t fs = F_loc / X::SYMS, off = F_loc - fs * X::SYMS;
		const uint32_t lastSide = (g.gbwtLen - 1) / X::SYMS;
		ft = g.gbwtLen;
		while(true) {
			if(fs != sideNum) { cur = load_side128(g.sides + (size_t)fs * 128); sideNum = fs; }
			const Bits256 f = bits_of_side<X>(cur, X::F_OFF);
			bool hit = false;
#pragma unroll
			for(int k = 0; k < 4; k++) {
				if(hit) continue;
				const uint64_t w = f.w[k] & ~low_mask((int)off - 64 * k);
				const uint32_t pc = (uint32_t)__builtin_popcountll(w);
				if(count <= pc) { ft = fs * X::SYMS + 64u * k + select_in_word(w, count); hit = true; }
				else count -= pc;
			}
			if(hit || fs >= lastSide) break;
			fs++;
			off = 0;
		}
	}
	*top_out = ft; *node_out = node;
}

// mapGLF (gfm.h:3759-3837): LF of a row range + translation of the outgoing-edge rows back to incoming rows
// through M-rank / F-select.  false = empty range.  `ie` may be null.
template <class X>
H2G_GLF bool map_glf(const X& g, uint32_t top, uint32_t bot, int c, uint32_t k, GRange* r, IEdges* ie) {
	const uint32_t s0 = top / X::SYMS, c0 = top - s0 * X::SYMS;
	Side128 sd = load_side128(g.sides + (size_t)s0 * 128);
	uint32_t t = rank_in_side128(g, sd, s0, c0, c), b;
	const uint32_t spread = bot - top;
	if(c0 + spread < X::SYMS) b = rank_in_side128(g, sd, s0, c0 + spread, c);   // initFromTopBot gfm.h:347
	else b = rank128(g, bot, c);
	if(ie) ie->n = 0;
	r->top = r->bot = r->node_top = r->node_bot = 0;
	if(t + 1 >= g.gbwtLen || t >= b) return false;
	const uint32_t node_top = rank_M(g, t + 1) - 1;
	uint32_t F_loc, M_occ;
	const uint32_t ft = node_to_Frow(g, t + 1, node_top, &F_loc, &M_occ);
	const uint32_t node_bot = rank_M(g, b);
	// :3812-3827 — the bottom takes the header of bot's own side, no backward scan
	const uint8_t* bsd = g.sides + (size_t)(b / X::SYMS) * 128;
	uint32_t bF = side_hdr_mem<X>(bsd, 0);
	const uint32_t bM = side_hdr_mem<X>(bsd, 1);
	if(bM > 0) bF++;
	const uint32_t fb = (node_bot + 1 > bM) ? select_F(g, bF, node_bot + 1 - bM) : bF;
	r->top = ft; r->bot = fb; r->node_top = node_top; r->node_bot = node_bot;
	if(ie && node_bot - node_top <= k && node_bot - node_top < fb - ft) in_edge_count(g, ft, fb, ie);
	return true;
}

// mapGLF1 (gfm.h:3957-4021) with mapLF1 (:3892): one row; false = cannot proceed on c
template <class X>
H2G_GLF bool map_glf1(const X& g, uint32_t row, int c, GRange* r) {
	r->top = r->bot = r->node_top = r->node_bot = 0;
	const uint32_t s0 = row / X::SYMS, c0 = row - s0 * X::SYMS;
	Side128 sd = load_side128(g.sides + (size_t)s0 * 128);
	if(rowL_in_side128(sd, c0) != c || is_zoff(g, row)) return false;
	const uint32_t t = rank_in_side128(g, sd, s0, c0, c);
	const uint32_t node_top = rank_M(g, t + 1) - 1;
	uint32_t F_loc, M_occ;
	const uint32_t ft = node_to_Frow(g, t + 1, node_top, &F_loc, &M_occ);
	const uint32_t node_bot = node_top + 1;
	const uint32_t fb = (node_bot + 1 > M_occ) ? select_F(g, F_loc, node_bot + 1 - M_occ) : F_loc;
	r->top = ft; r->bot = fb; r->node_top = node_top; r->node_bot = node_bot;
	return true;
}

// ------------------------------------------------------------------------------------------ partialSearch (a11)
// hi_aligner.h:6361-6600 on a graph index: mapGLF / mapGLF1 per base, node ranges drive the stop rules,
// the in-edge list of the last step rides along (:6522-6527) and gates reporting (:6551-6553).
// (inline in the fast unit it was measured slower: 73.2 -> 74.9 ms per step on the 256 Mbp SNP graph, lease T of round 6)
H2G_HDN void partial_search_graph_item(const DGfm& g, const SeqView& seq, uint32_t cur_in, bool pseudogeneStopIn,
                                      bool anchorStopIn, uint32_t khits, uint32_t kseeds, h2g_fm_hit* o, IEdges* ie_out)
{
	// (the aligner never arms pseudogeneStop on a graph index, hi_aligner.h:4669; the function itself honours it)
	const uint32_t len = seq.len, ftabLen = g.ftabChars, minK = g.minK;
	bool pseudogeneStop_ = pseudogeneStopIn, pseudogeneStop = false;
	bool anchorStop_ = anchorStopIn, anchorStop = false;
	h2g_fm_hit h;
	h.top = h.bot = h.node_top = h.node_bot = H2G_MAX;
	h.hit_type = H2G_CANDIDATE_HIT;
	h.numPartialSearch = 1; h.numUniqueSearch = 0; h.done = 0; h.nrank = 0; h.nside = 0;
	h.pseudogeneStop = 0; h.anchorStop = 0;
	uint32_t cur = cur_in, offset = cur_in, dep = cur_in;
	h.bwoff = offset;
	IEdges cur_ie, tmp_ie;
	cur_ie.n = 0;
	bool finished = false;
	if(len - dep < ftabLen + 1) { cur = len; h.len = cur - offset; h.done = 1; finished = true; }
	uint32_t top = 0, bot = 0;
	if(!finished) {
		uint32_t fi = 0;
		for(uint32_t i = 0; i < ftabLen; i++) {
			int c = seq.at(len - dep - 1 - i);
			if(c > 3) { cur += (i + 1); h.len = cur - offset; if(cur >= len) h.done = 1; finished = true; break; }
			fi |= (uint32_t)c << (2 * i);
		}
		if(!finished) {
			top = ftab_hi(g, fi);
			bot = ftab_lo(g, fi + 1);
			dep += ftabLen;
			if(top >= bot) { cur = dep; h.len = cur - offset; if(cur >= len) h.done = 1; finished = true; }
		}
	}
	if(!finished) {
		uint32_t same_range = 0, similar_range = 0, ntop = 0, nbot = 0;
		while(dep < len) {
			const int c = seq.at(len - dep - 1);
			GRange r;
			r.top = r.bot = r.node_top = r.node_bot = 0;
			tmp_ie.n = 0;
			if(c <= 3) {
				// nside: the sides of the BWT rows themselves (as on a linear index: one when top and bot share a side).  The sides the M / F bit vectors and the
				// header back-scan add are NOT counted — a lower bound of the unique sides SURVEY §8(d) asks for (round 6; it was 0 through round 5)
				if(bot - top > 1) {
					h.nrank += 2; h.nside += top / DGfm::SYMS == bot / DGfm::SYMS ? 1 : 2;
					map_glf(g, top, bot, c, kseeds, &r, &tmp_ie);
				} else {
					h.nrank += 1; h.nside += 1;
					if(map_glf1_fused(g, top, c, &r) && r.top + 1 < r.bot) {   // :6476-6482 (map_glf1 with one load per side touched)
						tmp_ie.n = 1; tmp_ie.e[0][0] = 0; tmp_ie.e[0][1] = r.bot - r.top - 1;
					}
				}
			}
			if(r.top >= r.bot) break;
			const uint32_t nt = r.node_bot - r.node_top, no = nbot - ntop;
			if(pseudogeneStop_) {                  // :6488-6503
				if(nt < no && no <= (khits < 5u ? khits : 5u)) {
					if(dep - offset >= minK + 6 && similar_range >= 5) { h.numUniqueSearch++; pseudogeneStop = true; break; }
				}
				if(nt != 1) {
					if(nt + 2 >= no) similar_range++;
					else if(nt + 4 < no) similar_range = 0;
				} else pseudogeneStop_ = false;
			}
			if(anchorStop_) {
				if(nt != 1 && no == nt) { if(++same_range >= 5) anchorStop_ = false; }
				else same_range = 0;
				if(dep - offset >= minK + 8 && nt >= 4) anchorStop_ = false;
			}
			top = r.top; bot = r.bot; ntop = r.node_top; nbot = r.node_bot;
			cur_ie.n = tmp_ie.n;
			for(uint32_t e = 0; e < tmp_ie.n && e < H2G_IEDGE_CAP; e++) { cur_ie.e[e][0] = tmp_ie.e[e][0]; cur_ie.e[e][1] = tmp_ie.e[e][1]; }
			dep++;
			if(anchorStop_ && dep - offset >= minK + 12 && bot - top == 1) { h.numUniqueSearch++; anchorStop = true; break; }
		}
		const uint32_t hit_type = anchorStop ? H2G_ANCHOR_HIT : (pseudogeneStop ? H2G_PSEUDOGENE_HIT : H2G_CANDIDATE_HIT);
		bool report = ntop < nbot;
		if(nbot - ntop < bot - top && cur_ie.n == 0) report = false;
		if(report) { h.top = top; h.bot = bot; h.node_top = ntop; h.node_bot = nbot; }
		else cur_ie.n = 0;
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
	if(ie_out) {
		ie_out->n = cur_ie.n;
		for(uint32_t e = 0; e < cur_ie.n && e < H2G_IEDGE_CAP; e++) { ie_out->e[e][0] = cur_ie.e[e][0]; ie_out->e[e][1] = cur_ie.e[e][1]; }
	}
}


// ------------------------------------------------------------------------------------------ graph SA walk (a14)
// getGenomeCoords (hi_aligner.h:5774-5855) on a graph index = GroupWalk2S::init/advanceElement (group_walk.h:1430-1545)
// driving GWState::init (:464-885) and GWState::advance (:1035-1336).  Elements are the NODES of [node_top, node_bot);
// a range is walked left as a group, split by preceding character (mapLFRange masks, gfm.h:3636) and at '$' rows, and
// merged when several rows lead into one node — merged-away duplicates get their own element index as the "offset"
// (group_walk.h:1171, :1246), reproduced because that value reaches joinedToTextOff.  tryOffset (gfm.h:2719) samples by
// node: (node & offMask) == node -> offs[node >> offRate].  Fixed capacities; exceeding one sets GwCtx::overflow.
// a capacity of the group walk was hit (the read is flagged; -DH2G_TRACE builds of the host instantiation say where)
#if defined(H2G_TRACE) && !defined(__HIP_DEVICE_COMPILE__)
#define GW_OVF(X) ((X)->overflow = 1, fprintf(stderr, "  group walk capacity hit at h2g_graph.h:%d\n", __LINE__))
#else
#define GW_OVF(X) ((X)->overflow = 1)
#endif
#ifndef H2G_GW_MAXELT             // the *_big units raise these (h2g_go_big.h)
#define H2G_GW_MAXELT 24          // >= kseeds (20 on graph indexes)
#define H2G_GW_MAXST 40
#define H2G_GW_MAXROWS 64         // rows of one sub-range handed to mapLFRange
#endif
struct GwPair { uint32_t first, second; };
struct GwState {
	uint32_t top, bot, node_top, node_bot, step, mapi, nmap, nie;
	uint32_t map[H2G_GW_MAXELT];
	GwPair   ie[H2G_GW_MAXELT];
};
struct GwCtx {
	uint32_t nelt, nst, nsteps, overflow;
	uint32_t offs[H2G_GW_MAXELT];
	uint32_t fmap[H2G_GW_MAXELT];       // GWHit::fmap[elt].first (the range currently holding the element)
	GwState  st[H2G_GW_MAXST];
};

template <class X>
H2G_HD int rowL128(const X& g, uint32_t row) {
	const uint32_t s0 = row / X::SYMS, c0 = row - s0 * X::SYMS;
	const uint32_t w = reinterpret_cast<const uint32_t*>(g.sides + (size_t)s0 * 128)[c0 >> 4];
	return (int)((w >> ((c0 & 15) * 2)) & 3);
}
template <class X>
H2G_HD uint32_t gw_try_offset(const X& g, uint32_t row, uint32_t node) {
	if(is_zoff(g, row)) return 0;
	if((node & g.offMask) == node) return g.offs_at(node >> g.offRate);
	return H2G_MAX;
}
// mapGLF1(row, l, &node_range) — no required character (gfm.h:4029-4095)
template <class X>
H2G_HDN void map_glf1_nochar(const X& g, uint32_t row, GRange* r) {
	if(is_zoff(g, row)) { r->top = r->bot = H2G_MAX; r->node_top = r->node_bot = 0; return; }
	const uint32_t s0 = row / X::SYMS, c0 = row - s0 * X::SYMS;
	Side128 sd = load_side128(g.sides + (size_t)s0 * 128);
	const int c = rowL_in_side128(sd, c0);
	const uint32_t t = rank_in_side128(g, sd, s0, c0, c);
	const uint32_t node_top = rank_M(g, t + 1) - 1;
	uint32_t F_loc, M_occ;
	const uint32_t ft = node_to_Frow(g, t + 1, node_top, &F_loc, &M_occ);
	const uint32_t node_bot = node_top + 1;
	const uint32_t fb = (node_bot + 1 > M_occ) ? select_F(g, F_loc, node_bot + 1 - M_occ) : F_loc;
	r->top = ft; r->bot = fb; r->node_top = node_top; r->node_bot = node_bot;
}
H2G_HD GwState* gw_new_state(GwCtx* x) {
	if(x->nst >= H2G_GW_MAXST) { GW_OVF(x); return &x->st[H2G_GW_MAXST - 1]; }
	GwState* s = &x->st[x->nst++];
	s->top = s->bot = s->node_top = s->node_bot = s->step = s->mapi = s->nmap = s->nie = 0;
	return s;
}
// GWState::init (group_walk.h:506-885).  The '$'-split creates new states and initialises them; those never split again
// at the same step deeper than the number of '$' rows, so the recursion of the reference is a bounded loop here.
template <class X>
H2G_HDN void gw_init(const X& g, GwCtx* x, uint32_t range0) {
	uint32_t pending[H2G_GW_MAXST];
	uint32_t npend = 0;
	pending[npend++] = range0;
	while(npend > 0) {
		const uint32_t range = pending[--npend];
		GwState* s = &x->st[range];
		uint32_t trimBegin = 0, trimEnd = 0, num_iedges = 0, e = 0;
		bool empty = true;
		for(uint32_t i = s->mapi; i < s->nmap; i++) {
			if(x->offs[s->map[i]] == H2G_MAX) {
				while(e < s->nie) {
					if(i <= s->ie[e].first) break;
					num_iedges += s->ie[e].second;
					e++;
				}
				uint32_t toff = gw_try_offset(g, s->top + i + num_iedges, s->node_top + i);
#if defined(H2G_TRACE) && !defined(__HIP_DEVICE_COMPILE__)
				fprintf(stderr, "        gw range %u step %u row %u node %u -> toff %u (top %u bot %u ntop %u nbot %u nie %u)\n", range, s->step, s->top + i + num_iedges, s->node_top + i, toff, s->top, s->bot, s->node_top, s->node_bot, s->nie);
#endif
				if(toff != H2G_MAX) {
					const uint32_t k = i + s->mapi;            // setOff indexes map_[i + mapi_] (:1013); mapi_ is 0 here
					if(k < s->nmap) x->offs[s->map[k]] = toff + s->step;
				}
			}
			if(x->offs[s->map[i]] != H2G_MAX) { if(empty) trimBegin++; else trimEnd++; }
			else { trimEnd = 0; empty = false; x->fmap[s->map[i]] = range; }
		}
		s->mapi += trimBegin;
		if(trimBegin > 0) {
			s->top += trimBegin;
			uint32_t k = 0;
			for(; k < s->nie; k++) {
				if(s->ie[k].first >= trimBegin) break;
				s->top += s->ie[k].second;
			}
			if(k > 0) { for(uint32_t q = k; q < s->nie; q++) s->ie[q - k] = s->ie[q]; s->nie -= k; }
			for(k = 0; k < s->nie; k++) s->ie[k].first -= trimBegin;
		}
		s->node_top += trimBegin;
		if(trimEnd > 0) {
			s->nmap -= trimEnd;
			s->bot -= trimEnd;
			const uint32_t node_range = s->node_bot - s->node_top;
			while(s->nie > 0) {
				if(s->ie[s->nie - 1].first < (node_range - trimEnd)) break;
				s->bot -= s->ie[s->nie - 1].second;
				s->nie--;
			}
		}
		s->node_bot -= trimEnd;
		if(empty) continue;
		// '$' rows strictly inside (top, bot): split (:741-868)
		uint32_t zin[4], nz = 0;
		for(uint32_t i = 0; i < g.nZ; i++) {
			const uint32_t z = i == 0 ? g.zoff : g.zoffs[i];
			if(z > s->top && z < s->bot) { if(nz < 4) zin[nz++] = z; else GW_OVF(x); }
		}
		if(nz == 0) continue;
		uint32_t g2n[H2G_GW_MAXROWS], ng = 0, n = 0, ee = 0;
		for(uint32_t r = 0; r < s->bot - s->top; r++) {
			if(ng < H2G_GW_MAXROWS) g2n[ng++] = n; else GW_OVF(x);
			if(ee < s->nie && n == s->ie[ee].first) {
				for(uint32_t a = 0; a < s->ie[ee].second; a++) { if(ng < H2G_GW_MAXROWS) g2n[ng++] = n; else GW_OVF(x); r++; }
				ee++;
			}
			n++;
		}
		if(x->overflow) continue;
		for(uint32_t i = 0; i < nz; i++) {
			const uint32_t new_top = zin[i] + 1;
			if(i + 1 < nz && new_top == zin[i + 1]) continue;
			if(new_top - s->top == ng) break;
			const uint32_t new_node_top = g2n[new_top - s->top] + s->node_top;
			const uint32_t new_bot = (i + 1 < nz) ? zin[i + 1] : s->bot;
			uint32_t new_node_bot = s->node_bot;
			if(new_bot - s->top < ng) {
				new_node_bot = s->node_top + g2n[new_bot - s->top];
				if(new_bot - s->top > 0 && g2n[new_bot - s->top] == g2n[new_bot - s->top - 1]) new_node_bot++;
			}
			if(new_top >= new_bot) continue;
			GwState* ns = gw_new_state(x);
			if(x->overflow) break;
			for(uint32_t j = new_top - s->top; j + 1 < new_bot - s->top;) {
				const uint32_t nn = g2n[j];
				uint32_t j2 = j + 1;
				while(j2 < new_bot - s->top) { if(nn != g2n[j2]) break; j2++; }
				if(j + 1 < j2) {
					if(ns->nie < H2G_GW_MAXELT) { ns->ie[ns->nie].first = nn - (new_node_top - s->node_top); ns->ie[ns->nie].second = j2 - j - 1; ns->nie++; }
					else GW_OVF(x);
				}
				j = j2;
			}
			ns->nmap = new_node_bot - new_node_top; ns->mapi = 0;
			if(ns->nmap > H2G_GW_MAXELT) { GW_OVF(x); ns->nmap = H2G_GW_MAXELT; }
			for(uint32_t j = 0; j < ns->nmap; j++) ns->map[j] = s->map[new_node_top + j - s->node_top + s->mapi];
			ns->top = new_top; ns->bot = new_bot; ns->node_top = new_node_top; ns->node_bot = new_node_bot; ns->step = s->step;
			// the reference initialises the new state right here (depth-first); deferring it is equivalent because the
			// new states only touch their own elements' offs/fmap entries, which this state no longer owns
			if(npend < H2G_GW_MAXST) pending[npend++] = x->nst - 1; else GW_OVF(x);
		}
		s->bot = zin[0];
		s->node_bot = g2n[s->bot - s->top - 1] + s->node_top + 1;
		s->nmap = s->node_bot - s->node_top + s->mapi;
		uint32_t width = s->node_bot - s->node_top;
		for(uint32_t k = 0; k < s->nie; k++) {
			if(s->ie[k].first >= s->node_bot - s->node_top) { s->nie = k; break; }
			width += s->ie[k].second;
		}
		if(width != s->bot - s->top && s->nie > 0) {
			s->ie[s->nie - 1].second -= 1;
			if(s->ie[s->nie - 1].second == 0) s->nie--;
		}
	}
}
// narrowing of a freshly mapped element list whose rows merged into fewer nodes (:1143-1185, :1218-1262)
template <class X>
H2G_HDN void gw_merge_dups(const X& g, GwCtx* x, uint32_t curtop, uint64_t mask, uint32_t nmask, int c, uint32_t* map, uint32_t* nmap) {
	uint32_t j1 = 0, j2 = 0;
	for(uint32_t k = 0; k < nmask; k++) if((mask >> k) & 1) { j1 = k; break; }
	for(uint32_t j = 0; j + 1 < *nmap; j++) {
		for(uint32_t k = j1 + 1; k < nmask; k++) if((mask >> k) & 1) { j2 = k; break; }
		GRange r;
		map_glf(g, curtop + j1, curtop + j2 + 1, c, 5, &r, nullptr);
		if(r.node_bot - r.node_top == 1) { x->offs[map[j]] = map[j]; map[j] = H2G_MAX; }
		j1 = j2; j2 = 0;
	}
	uint32_t w = 0;
	for(uint32_t j = 0; j < *nmap; j++) if(map[j] != H2G_MAX) map[w++] = map[j];
	*nmap = w;
}
// GWState::advance (group_walk.h:1035-1336)
template <class X>
H2G_HDN void gw_advance(const X& g, GwCtx* x, uint32_t range) {
	GwState* s = &x->st[range];
	x->nsteps++;
	if(s->bot - s->top > 1) {
		bool first = true;
		uint32_t newtop = 0, newbot = 0, new_node_top = 0, new_node_bot = 0;
		uint32_t gmap[H2G_GW_MAXELT], ngmap = 0;
		IEdges backup;
		backup.n = 0;
		uint32_t curtop = s->top, curbot = s->bot, cur_node_top = s->node_top, cur_node_bot = s->node_bot;
		const uint32_t nie0 = s->nie;
		for(uint32_t e = 0; e < nie0 + 1; e++) {
			if(e >= s->nie) {
				if(e > 0) {
					curtop = curbot + s->ie[e - 1].second;
					curbot = s->bot;
					if(curtop >= curbot) break;
					cur_node_top = cur_node_bot;
					cur_node_bot = s->node_bot;
				}
			} else {
				if(e > 0) {
					curtop = curbot + s->ie[e - 1].second;
					curbot = curtop + (s->ie[e].first - s->ie[e - 1].first);
					cur_node_top = cur_node_bot;
				} else curbot = curtop + s->ie[e].first + 1;
				cur_node_bot = s->node_top + s->ie[e].first + 1;
			}
			uint32_t n = curbot - curtop;
			if(n > H2G_GW_MAXROWS) { GW_OVF(x); return; }
			uint64_t mask[4] = {0, 0, 0, 0};
			for(uint32_t k = 0; k < n; k++) mask[rowL128(g, curtop + k)] |= 1ull << k;          // mapLFRange masks
			for(int c = 0; c < 4; c++) {
				if(mask[c] == 0) continue;
				GRange r;
				IEdges tie;
				map_glf(g, curtop, curbot, c, cur_node_bot - cur_node_top, &r, &tie);
				if(tie.n > H2G_GW_MAXELT) { GW_OVF(x); return; }
				if(first) {
					first = false;
					newtop = r.top; newbot = r.bot; new_node_top = r.node_top; new_node_bot = r.node_bot;
					backup.n = tie.n;
					for(uint32_t k = 0; k < tie.n; k++) { backup.e[k][0] = tie.e[k][0]; backup.e[k][1] = tie.e[k][1]; }
					for(uint32_t j = 0; j < n; j++) if((mask[c] >> j) & 1) {
						if(ngmap < H2G_GW_MAXELT) gmap[ngmap++] = s->map[j + s->mapi + (cur_node_top - s->node_top)]; else GW_OVF(x);
					}
					if(new_node_bot - new_node_top < ngmap) gw_merge_dups(g, x, curtop, mask[c], n, c, gmap, &ngmap);
				} else {
					GwState* ns = gw_new_state(x);
					if(x->overflow) return;
					s = &x->st[range];
					for(uint32_t j = 0; j < n; j++) if((mask[c] >> j) & 1) {
						if(ns->nmap < H2G_GW_MAXELT) ns->map[ns->nmap++] = s->map[j + s->mapi + (cur_node_top - s->node_top)]; else GW_OVF(x);
					}
					if(r.node_bot - r.node_top < ns->nmap) gw_merge_dups(g, x, curtop, mask[c], n, c, ns->map, &ns->nmap);
					ns->top = r.top; ns->bot = r.bot; ns->node_top = r.node_top; ns->node_bot = r.node_bot;
					ns->nie = tie.n;
					for(uint32_t k = 0; k < tie.n; k++) { ns->ie[k].first = tie.e[k][0]; ns->ie[k].second = tie.e[k][1]; }
					ns->step = s->step + 1;
					gw_init(g, x, x->nst - 1);
					if(x->overflow) return;
				}
			}
		}
		s->mapi = 0;
		s->top = newtop; s->bot = newbot; s->node_top = new_node_top; s->node_bot = new_node_bot;
		s->nie = backup.n;
		for(uint32_t k = 0; k < backup.n; k++) { s->ie[k].first = backup.e[k][0]; s->ie[k].second = backup.e[k][1]; }
		if(ngmap > 0) { for(uint32_t k = 0; k < ngmap; k++) s->map[k] = gmap[k]; s->nmap = ngmap; }
	} else {
		GRange r;
		if(is_zoff(g, s->top)) map_glf1_nochar(g, s->top, &r);       // (not reached: GWState::init resolves a '$' row before it is advanced)
		else { glf1_top_fused(g, s->top, &r.top, &r.node_top); r.node_bot = r.node_top + 1; }   // map_glf1_nochar's (top, node range) with one load per side touched
		s->top = r.top; s->bot = r.top + 1; s->node_top = r.node_top; s->node_bot = r.node_bot;
		if(s->mapi > 0) { s->map[0] = s->map[s->mapi]; s->mapi = 0; }
		s->nmap = 1;
	}
	s->step++;
	gw_init(g, x, range);
}

// The group walk of ONE element that is ONE row (no in-edges): GWState::advance's single-row branch (group_walk.h:1290-1336) and
// GWState::init's tryOffset, nothing else — a row stays a row.  Advances (*row, *node, *steps) by at most `budget` LF steps; true = the
// offset was found (*off = tryOffset + steps, what gw_resolve leaves in offs[0]; *steps = its nsteps).
template <class X>
H2G_HD bool gw_walk_single(const X& g, uint32_t* row, uint32_t* node, uint32_t* steps, uint32_t budget, uint32_t* off) {
	while(true) {
		const uint32_t toff = gw_try_offset(g, *row, *node);
		if(toff != H2G_MAX) { *off = toff + *steps; return true; }
		if(budget == 0) return false;
		budget--;
		uint32_t r, n;
		glf1_top_fused(g, *row, &r, &n);
		*row = r; *node = n; (*steps)++;
	}
}

// GroupWalk2S::init + advanceElement for every element (group_walk.h:1430-1545): fills x->offs[0 .. *nelt) with the joined
// offsets (index-local for a local index).  false on capacity overflow.
template <class X>
H2G_HDN bool gw_resolve(const X& g, GwCtx* x, uint32_t top, uint32_t bot, uint32_t node_top, uint32_t node_bot, const IEdges* ie,
                       uint32_t maxelt, uint32_t* nelt_out)
{
	uint32_t nelt = node_bot - node_top;
	if(nelt > maxelt) nelt = maxelt;
	*nelt_out = nelt;
	if(nelt > H2G_GW_MAXELT || (ie && ie->n > H2G_GW_MAXELT)) return false;
	x->nelt = nelt; x->nst = 0; x->nsteps = 0; x->overflow = 0;
	for(uint32_t i = 0; i < nelt; i++) { x->offs[i] = H2G_MAX; x->fmap[i] = H2G_MAX; }
	GwState* s = gw_new_state(x);                             // GroupWalk2S::init :1430-1470
	s->nmap = nelt;
	for(uint32_t i = 0; i < nelt; i++) s->map[i] = i;
	s->top = top; s->bot = bot; s->node_top = node_top; s->node_bot = node_top + nelt;
	s->nie = ie ? ie->n : 0;
	for(uint32_t k = 0; k < s->nie; k++) { s->ie[k].first = ie->e[k][0]; s->ie[k].second = ie->e[k][1]; }
	gw_init(g, x, 0);
	for(uint32_t elt = 0; elt < nelt; elt++) {
		uint32_t guard = 0;
		while(x->offs[elt] == H2G_MAX) {                      // advanceElement :1491-1545
			if(x->overflow || x->fmap[elt] == H2G_MAX || ++guard > 200000u) {
#if defined(H2G_TRACE) && !defined(__HIP_DEVICE_COMPILE__)
				fprintf(stderr, "  gw_resolve gives up: overflow %u fmap %u guard %u (elt %u of %u, top %u bot %u nodes %u %u)\n", x->overflow, x->fmap[elt], guard, elt, nelt, top, bot, node_top, node_bot);
#endif
				return false;
			}
			gw_advance(g, x, x->fmap[elt]);
		}
	}
	return !x->overflow;
}

// one getGenomeCoords call (hi_aligner.h:5774-5855); x = caller-provided scratch.  res->ok = 0 also on capacity overflow
// (res->nsteps = H2G_MAX then)
H2G_HDN void genome_coords_graph_item(const DGfm& g, GwCtx* x, uint32_t top, uint32_t bot, uint32_t node_top, uint32_t node_bot,
                                     const IEdges* ie, uint32_t maxelt, uint32_t rdlen, bool rejectStraddle, h2g_coord* coords,
                                     uint32_t cap, h2g_sa_result* res)
{
	res->ok = 0; res->ncoords = 0; res->straddled = 0; res->nsteps = 0;
	uint32_t nelt = 0;
	if(!gw_resolve(g, x, top, bot, node_top, node_bot, ie, maxelt, &nelt) || nelt > cap) { res->nsteps = H2G_MAX; return; }
	for(uint32_t elt = 0; elt < nelt; elt++) {
		uint32_t tidx = 0, toff = 0;
		bool st2 = false;
		joined_to_text(g, rdlen, x->offs[elt], &tidx, &toff, rejectStraddle, &st2);
		res->straddled |= st2 ? 1u : 0u;
		if(tidx == H2G_MAX) { res->nsteps = x->nsteps; return; }
		coords[res->ncoords].tidx = st2 ? H2G_MAX : tidx;
		coords[res->ncoords].toff = toff;
		coords[res->ncoords].joinedOff = x->offs[elt];
		res->ncoords++;
	}
	res->nsteps = x->nsteps;
	res->ok = 1;
}

// globalGFMSearch (hi_aligner.h:6606-6744) / localGFMSearch (:6751-6892) on a graph index.  X = side policy (DGfm / LGfm),
// FT = ftab access (ftabChars(), lohi()).  Returns nelt (nodes); r / ie as the reference leaves top/bot/node range/iedges.
template <class X, class FT>
H2G_HDN uint32_t gfm_search_graph(const X& g, const FT& ft, const SeqView& seq, uint32_t rdoff, uint32_t* hitlen, GRange* out,
                                 IEdges* ie_out, bool* uniqueStop, uint32_t minUniqueLen, uint32_t maxHitLen, uint32_t maxHits,
                                 bool local, uint32_t kseeds, uint32_t* nrank)
{
	const bool uniqueStop_ = *uniqueStop;
	*uniqueStop = false;
	const uint32_t ftabLen = ft.ftabChars(), len = seq.len;
	const uint32_t offset = len - rdoff - 1;
	uint32_t dep = offset;
	ie_out->n = 0;
	if(local) { out->top = out->bot = out->node_top = out->node_bot = 0; }
	const uint32_t left = len - dep;
	if(left < ftabLen + 1) { *hitlen = left; return 0; }
	uint32_t fi = 0;
	for(uint32_t i = 0; i < ftabLen; i++) {
		const int c = seq.at(len - dep - 1 - i);
		if(c > 3) { *hitlen = i + 1; return 0; }
		fi |= (uint32_t)c << (2 * i);
	}
	uint32_t top, bot;
	ft.lohi(fi, &top, &bot);
	dep += ftabLen;
#if defined(H2G_TRACE) && !defined(__HIP_DEVICE_COMPILE__)
	if(local) fprintf(stderr, "      lsearch ftab key %u -> top %u bot %u\n", fi, top, bot);
#endif
	if(top >= bot) { *hitlen = ftabLen; return 0; }
	uint32_t ntop = 0, nbot = 0;
	IEdges tmp;
	while(dep < len) {
		const int c = seq.at(len - dep - 1);
		GRange r;
		r.top = r.bot = r.node_top = r.node_bot = 0;
		tmp.n = 0;
		if(c <= 3) {
			if(bot - top > 1) { nrank[0] += 2; nrank[1] += top / X::SYMS == bot / X::SYMS ? 1 : 2; map_glf(g, top, bot, c, kseeds, &r, &tmp); }   // (sides of the rows only: partial_search_graph_item)
			else {
				nrank[0] += 1; nrank[1] += 1;
				if(map_glf1_fused(g, top, c, &r) && r.top + 1 < r.bot) { tmp.n = 1; tmp.e[0][0] = 0; tmp.e[0][1] = r.bot - r.top - 1; }
			}
		}
#if defined(H2G_TRACE) && !defined(__HIP_DEVICE_COMPILE__)
		if(local) fprintf(stderr, "      lsearch step c %d: [%u,%u) -> [%u,%u) nodes [%u,%u)\n", c, top, bot, r.top, r.bot, r.node_top, r.node_bot);
#endif
		if(r.top >= r.bot) break;
		top = r.top; bot = r.bot; ntop = r.node_top; nbot = r.node_bot;
		ie_out->n = tmp.n;
		for(uint32_t e = 0; e < tmp.n && e < H2G_IEDGE_CAP; e++) { ie_out->e[e][0] = tmp.e[e][0]; ie_out->e[e][1] = tmp.e[e][1]; }
		dep++;
		if(uniqueStop_ && bot - top == 1 && dep - offset >= minUniqueLen) { *uniqueStop = true; break; }
		if(local && dep - offset >= maxHitLen) break;
	}
	if(ntop < nbot && nbot - ntop <= maxHits) {
		out->top = top; out->bot = bot; out->node_top = ntop; out->node_bot = nbot;
		*hitlen = dep - offset;
		return nbot - ntop;
	}
	return 0;
}

// ------------------------------------------------------------------------------------------ ALT-aware extension (a18, a28)
// ALT (alt.h:41-120) as GFM::GFM loads the list (gfm.h:728-905): one reversed copy per deletion appended
// (pos = last deleted base, reversed = 1 in the low byte of seq), sorted by ALT::operator< (alt.h:88-102).
enum { H2G_ALT_SNP_SGL = 1, H2G_ALT_SNP_INS = 2, H2G_ALT_SNP_DEL = 3, H2G_ALT_SNP_ALT = 4, H2G_ALT_SPLICESITE = 5, H2G_ALT_EXON = 6 };
struct DAlt { uint32_t pos, type, len, pad; uint64_t seq; };
struct DAlts {
	const DAlt* a; uint32_t n; uint32_t maxAltsTried;      // GraphPolicy::maxAltsTried = 16 (hisat2.cpp:521)
	// optional position index: bucket[b] = first ALT with pos >= b << H2G_ALT_BUCKET_SHIFT, so that the lower bound is one load
	// plus a scan of the few ALTs of the bucket instead of ~log2(n) dependent loads (15 at E. coli scale, 24 at GRCh38+SNP scale)
	const uint32_t* bucket = nullptr; uint32_t nbucket = 0;
	// bit 0: ALTDB::hasSpliceSites(), the list holds splice-site ALTs (a --ss index).  bit 1: --haplotype is on AND the haplotype table
	// follows the ALT array in memory (pack_alts below) — kept out of this struct so that the kernel arguments of the units built without
	// H2G_HAPLOTYPE stay what they were
	uint32_t has_splice = 0;
};
// ALTDB::haplotypes() (alt.h:209) as laid out behind the ALTs: u32 {nhap, nids, 0, 0}, left[nhap], right[nhap], maxright[nhap],
// first[nhap + 1], ids[nids].  [left, right] joined, sorted by (left, right); ids[first[h] .. first[h+1]) = positions in the ALT list of
// the SNPs haplotype h carries; maxright[h] = max right over 0..h (gfm.h:907-921)
struct DHaps { const uint32_t *left, *right, *maxright, *first, *ids; uint32_t n; };
H2G_HD DHaps haps_of(const DAlts& A) {
	const uint32_t* h = reinterpret_cast<const uint32_t*>(A.a + A.n);
	DHaps H;
	H.n = h[0]; H.left = h + 4; H.right = H.left + H.n; H.maxright = H.right + H.n; H.first = H.maxright + H.n; H.ids = H.first + H.n + 1;
	return H;
}
// host: the ALT array followed by the haplotype table, one buffer (device upload and the host instantiation share it)
template <class ALT_T>
inline void pack_alts(const std::vector<ALT_T>& alts, const std::vector<uint32_t>& left, const std::vector<uint32_t>& right,
                      const std::vector<uint32_t>& maxright, const std::vector<uint32_t>& first, const std::vector<uint32_t>& ids,
                      std::vector<uint64_t>& out) {
	static_assert(sizeof(ALT_T) == sizeof(DAlt) && sizeof(DAlt) % 8 == 0, "ALT records are 8-byte multiples");
	const size_t nh = left.size();
	std::vector<uint32_t> t = {(uint32_t)nh, (uint32_t)ids.size(), 0u, 0u};
	t.insert(t.end(), left.begin(), left.end()); t.insert(t.end(), right.begin(), right.end()); t.insert(t.end(), maxright.begin(), maxright.end());
	if(first.size() == nh + 1) t.insert(t.end(), first.begin(), first.end()); else t.insert(t.end(), nh + 1, 0u);
	t.insert(t.end(), ids.begin(), ids.end());
	if(t.size() & 1) t.push_back(0u);
	out.assign((alts.size() * sizeof(DAlt) + t.size() * 4) / 8, 0);
	if(!alts.empty()) memcpy(out.data(), alts.data(), alts.size() * sizeof(DAlt));
	memcpy(reinterpret_cast<uint8_t*>(out.data()) + alts.size() * sizeof(DAlt), t.data(), t.size() * 4);
}
#define H2G_ALT_BUCKET_SHIFT 7

H2G_HD uint32_t alt_lobound(const DAlts& A, uint32_t pos) {   // EList::bsearchLoBound with a type-NONE key: first pos >= key
	if(A.bucket) {
		const uint32_t b = pos >> H2G_ALT_BUCKET_SHIFT;
		if(b >= A.nbucket) return A.n;
		uint32_t i = A.bucket[b];
		while(i < A.n && A.a[i].pos < pos) i++;
		return i;
	}
	uint32_t lo = 0, hi = A.n;
	while(lo < hi) { const uint32_t m = (lo + hi) >> 1; if(A.a[m].pos < pos) lo = m + 1; else hi = m; }
	return lo;
}
// host: the bucket table for a sorted ALT list (one extra entry so that every pos below the last bucket end resolves)
inline void alt_buckets(const DAlt* a, uint32_t n, std::vector<uint32_t>& out) {
	out.clear();
	if(n == 0) return;
	const uint32_t nb = (a[n - 1].pos >> H2G_ALT_BUCKET_SHIFT) + 2;
	out.assign(nb, n);
	uint32_t i = 0;
	for(uint32_t b = 0; b < nb; b++) { while(i < n && (a[i].pos >> H2G_ALT_BUCKET_SHIFT) < b) i++; out[b] = i; }
}

H2G_HD int char_base(uint8_t ch) { return ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : ch == 'N' ? 4 : 0; }   // asc2dna (alphabet.cpp:298)
// GenomeHit::replace_edits_with_alts (hi_aligner.h:1229-1319): after the SwAligner pass on a graph index, edits that coincide
// with a known SNP get its id (and then cost nothing in calculateScore).  A run of gap edits is delimited by edit TYPE only,
// as in the reference; the caller re-scores the hit.
H2G_HDN void replace_edits_with_alts(const DAlts& A, h2g_ghit* h) {
	if(A.n == 0 || h->nedits == 0) return;
	int64_t offset = 0;
	uint32_t i = 0;
	while(i < h->nedits) {
		uint32_t next_i = i + 1;
		h2g_edit& ed = h->edits[i];
		const bool gap = ed.type == H2G_EDIT_READ_GAP || ed.type == H2G_EDIT_REF_GAP;
		if(gap) { for(; next_i < h->nedits; next_i++) if(h->edits[next_i].type != ed.type) break; }
		const uint32_t glen = next_i - i;
		if(ed.snp == H2G_MAX) {
			const uint32_t cpos = (uint32_t)((int64_t)h->joinedOff + ed.pos + offset);
			for(uint32_t ai = alt_lobound(A, cpos); ai < A.n; ai++) {
				const DAlt& alt = A.a[ai];
				if(alt.pos > cpos) break;
				if(ed.type == H2G_EDIT_MM) {
					if(alt.type != H2G_ALT_SNP_SGL) continue;
					if(alt.seq < 4 && base_char((int)alt.seq) == ed.qchr) { ed.snp = ai; break; }
				} else if(ed.type == H2G_EDIT_READ_GAP) {
					if(alt.type != H2G_ALT_SNP_DEL) continue;
					if(alt.len == glen) { for(uint32_t k = i; k < next_i; k++) h->edits[k].snp = ai; break; }
				} else {
					if(alt.type != H2G_ALT_SNP_INS) continue;
					if(alt.len == glen) {
						uint64_t seq = 0;
						for(uint32_t k = i; k < next_i; k++) seq = (seq << 2) | (uint64_t)char_base(h->edits[k].qchr);
						if(alt.seq == seq) { for(uint32_t k = i; k < next_i; k++) h->edits[k].snp = ai; break; }
					}
				}
			}
		}
		if(ed.type == H2G_EDIT_READ_GAP) offset += glen;
		else if(ed.type == H2G_EDIT_REF_GAP) offset -= glen;
		i = next_i;
	}
}

// alignWithALTs_recur (hi_aligner.h:2763-3550) for SNP ALTs (single / insertion / deletion), recursion turned into an
// explicit stack, incl. the splice-site ALTs of --ss indexes (exon ALTs are skipped as in the reference); haplotypes are unused
// (use_haplotype = false, hisat2.cpp:522).  The reference window needs no buffers: rfseq[i] is always the base of text
// `tidx` at rfoff + i (4 outside the text), which RefCursor serves directly.
#ifndef H2G_AWA_DEPTH
#define H2G_AWA_DEPTH 12
#endif
struct AwaFrame {
	uint32_t joinedOff, rdoff_add, rdoff, rdlen, rflen, tmp_numNs, orig_nedits, next_rdlen, rd_i, max_rd_i, dep;
	uint32_t prev_alt_type, cur_alt_type;   // the ALT the parent frame went through; the one this frame is trying (splice sites search further, :3523)
	int32_t  rfoff, a_first, a_second, min_rd_i;
	uint32_t state;   // 0 entry, 1 loop, 2 after call
};
#ifndef H2G_AWA_CAND
#define H2G_AWA_CAND 4
#endif
// --haplotype: compiled into the units that define H2G_SPLICE_DB 1 (and the host instantiation); the other units keep the code and
// the workspace layout they were verified with (every unit reports its own layout, h2g_go_args.h).
#ifndef H2G_HAPLOTYPE
#ifdef H2G_SPLICE_DB
#define H2G_HAPLOTYPE H2G_SPLICE_DB
#else
#define H2G_HAPLOTYPE 1
#endif
#endif
#ifndef H2G_HT_CAP
#define H2G_HT_CAP 16
#endif
// ht_llist[dep] (hi_aligner.h:410, :2899): the haplotypes still compatible with the ALTs taken so far, each with the position of the
// next ALT it expects (walking left: counting down; right: counting up, == the haplotype's size once it is exhausted)
struct HtList { uint32_t ht[H2G_HT_CAP]; uint16_t idx[H2G_HT_CAP]; uint32_t n; };
struct AwaWS {
	h2g_edit tmp[H2G_GHIT_EDITS];
	uint32_t ntmp;
	AwaFrame fr[H2G_AWA_DEPTH];
#if H2G_HAPLOTYPE
	HtList ht[H2G_AWA_DEPTH];
#endif
	// candidate_edits (ELList<Edit,128,4>): other edit lists reaching the same best offset (adjustWithALT only)
	h2g_edit cand[H2G_AWA_CAND][H2G_GHIT_EDITS];
	uint32_t cand_n[H2G_AWA_CAND], ncand;
	h2g_ghit scratch;     // adjust_with_alt builds its candidate hit here
};

#if H2G_HAPLOTYPE
// ALT::isSame alt.h:127-149 (reversed = the low byte of seq: set on the mirrored copy of a deletion)
H2G_HD bool alt_is_same(const DAlt& a, const DAlt& o) {
	if(a.type != o.type) return false;
	if(a.type == H2G_ALT_SNP_SGL) return a.pos == o.pos && a.seq == o.seq;
	if(a.type == H2G_ALT_SNP_INS && a.seq != o.seq) return false;
	const bool ra = (a.seq & 0xff) != 0, ro = (o.seq & 0xff) != 0;
	if(ra == ro) return a.pos == o.pos && a.len == o.len;
	if(ra) return a.pos - a.len + 1 == o.pos && a.len == o.len;
	return a.pos == o.pos - o.len + 1 && a.len == o.len;
}
H2G_HD bool ht_has(const HtList& L, uint32_t ht) { for(uint32_t h = 0; h < L.n; h++) if(L.ht[h] == ht) return true; return false; }
H2G_HD void ht_push(HtList& L, uint32_t ht, uint32_t idx, uint32_t* ovf) {
	if(L.n >= H2G_HT_CAP) { *ovf = 1; return; }
	L.ht[L.n] = ht; L.idx[L.n] = (uint16_t)idx; L.n++;
}
// add_haplotypes hi_aligner.h:2648-2757: the haplotypes a walk arriving at cmp (= joinedOff) can still meet within one read length
H2G_HD void add_haplotypes(const DAlts& A, const DHaps& H, uint32_t cmp, HtList& L, uint32_t rdlen, bool left_ext, bool initial, uint32_t* ovf) {
	uint32_t lo = 0, hi = H.n;                                   // bsearchLoBound over (left, right) with key (cmp, cmp)
	while(lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if(H.left[mid] < cmp || (H.left[mid] == cmp && H.right[mid] < cmp)) lo = mid + 1; else hi = mid;
	}
	if(lo >= H.n) return;
	if(left_ext) {
		for(int first = (int)lo; first >= 0; first--) {
			const uint32_t right = H.right[first];
			if(!initial && right >= cmp) continue;
			if((uint32_t)(H.maxright[first] + rdlen - 1) < cmp) break;
			const uint32_t o = H.first[first], nal = H.first[first + 1] - o;
			if(nal == 0 || ht_has(L, (uint32_t)first)) continue;
			if(right < cmp) ht_push(L, (uint32_t)first, nal - 1, ovf);
			else {
				uint32_t second = nal;
				for(int a = (int)nal - 1; a >= 0; a--) {
					const uint32_t alti = H.ids[o + (uint32_t)a];
					second = (uint32_t)a;
					if(alti < A.n && cmp > A.a[alti].pos) break;
				}
				if(second != nal) ht_push(L, (uint32_t)first, second, ovf);
			}
		}
		return;
	}
	if(initial) {
		for(int first = (int)lo; first >= 0; first--) {
			if(H.maxright[first] < cmp) break;
			if(H.right[first] < cmp || H.left[first] > cmp) continue;
			const uint32_t o = H.first[first], nal = H.first[first + 1] - o;
			if(nal == 0 || ht_has(L, (uint32_t)first)) continue;
			uint32_t second = nal;
			for(uint32_t a = 0; a < nal; a++) {
				const uint32_t alti = H.ids[o + a];
				second = a;
				if(alti < A.n && cmp <= A.a[alti].pos) break;
			}
			if(second != nal) ht_push(L, (uint32_t)first, second, ovf);
		}
	}
	for(uint32_t second = lo; second < H.n; second++) {
		if(H.left[second] < cmp) continue;
		if(H.left[second] >= cmp + rdlen) break;
		if(H.first[second + 1] == H.first[second] || ht_has(L, second)) continue;
		ht_push(L, second, 0, ovf);
	}
}
// "Check to see if there is a haplotype that supports this alt" (:2981-2996 leftwards, :3314-3331 rightwards)
H2G_HD bool ht_supports(const DAlts& A, const DHaps& H, const HtList& L, uint32_t alti) {
	for(uint32_t h = 0; h < L.n; h++) {
		const uint32_t o = H.first[L.ht[h]], nal = H.first[L.ht[h] + 1] - o;
		if(L.idx[h] >= nal) continue;
		const uint32_t hi = H.ids[o + L.idx[h]];
		if(hi < A.n && alt_is_same(A.a[alti], A.a[hi])) return true;
	}
	return false;
}
#endif

H2G_HDN uint32_t align_with_alts(const DRef& ref, const DAlts& A, const SeqView& seq, uint32_t joinedOff0, uint32_t base_rdoff,
                                uint32_t rdoff0, uint32_t rdlen0, uint32_t tidx, int rfoff0, uint32_t rflen0, bool left,
                                h2g_ghit* h, uint32_t mm, uint32_t* numNs, AwaWS* W, bool want_cand = false)
{
	if(numNs) *numNs = 0;
	W->ncand = 0;
#if H2G_HAPLOTYPE
	const bool use_hap = (A.has_splice & 2u) != 0;         // --haplotype (GraphPolicy::useHaplotype gp.h:71) and the table is there
	DHaps H; H.left = H.right = H.maxright = H.first = H.ids = nullptr; H.n = 0;
	if(use_hap) H = haps_of(A);
#endif
	const uint32_t nedits0 = h->nedits;
	W->ntmp = nedits0;
	for(uint32_t k = 0; k < nedits0; k++) W->tmp[k] = h->edits[k];
	int best_rdoff = (int)rdoff0;
	uint32_t numALTsTried = 0;
	const uint32_t contig_len = ref.refLens[tidx];
	RefCursor rc;
	rc.init(&ref, tidx);
#define AWA_RF(F, I) ((int64_t)(F).rfoff + (int64_t)(I) < 0 ? 4 : rc.get((int64_t)(F).rfoff + (int64_t)(I)))
#define AWA_PUSH_FRONT(E) do { if(W->ntmp >= H2G_GHIT_EDITS) h->overflow = 1; else { for(int q_ = (int)W->ntmp - 1; q_ >= 0; q_--) W->tmp[q_ + 1] = W->tmp[q_]; W->tmp[0] = (E); W->ntmp++; } } while(0)
#define AWA_PUSH_BACK(E) do { if(W->ntmp >= H2G_GHIT_EDITS) h->overflow = 1; else W->tmp[W->ntmp++] = (E); } while(0)
#define AWA_ERASE_FRONT(N) do { const uint32_t n_ = (N); for(uint32_t q_ = n_; q_ < W->ntmp; q_++) W->tmp[q_ - n_] = W->tmp[q_]; W->ntmp -= n_; } while(0)
#define AWA_COMMIT() do { for(uint32_t q_ = 0; q_ < W->ntmp; q_++) h->edits[q_] = W->tmp[q_]; h->nedits = W->ntmp; } while(0)
#define AWA_RETURN(V) do { ret = (V); sp--; goto next_frame; } while(0)
#define AWA_CAND_PUSH() do { if(want_cand) { if(W->ncand >= H2G_AWA_CAND) h->overflow = 1; else { for(uint32_t q_ = 0; q_ < W->ntmp; q_++) W->cand[W->ncand][q_] = W->tmp[q_]; W->cand_n[W->ncand] = W->ntmp; W->ncand++; } } } while(0)
	int sp = 0;
	uint32_t ret = 0;
	{
		AwaFrame& f = W->fr[0];
		f.joinedOff = joinedOff0; f.rdoff_add = rdoff0 - base_rdoff; f.rdoff = rdoff0; f.rdlen = rdlen0; f.rfoff = rfoff0; f.rflen = rflen0;
		f.tmp_numNs = 0; f.dep = 0; f.state = 0; f.prev_alt_type = 0; f.cur_alt_type = 0;
	}
	while(sp >= 0) {
		{
		AwaFrame& f = W->fr[sp];
		if(f.state == 0) {
			if(numALTsTried > A.maxAltsTried + f.dep) AWA_RETURN(0);
			if(f.rfoff < -16) AWA_RETURN(0);
			if((int64_t)f.rfoff >= (int64_t)contig_len) AWA_RETURN(0);
			if(f.rfoff >= 0 && (uint64_t)f.rfoff + f.rflen > contig_len) f.rflen = contig_len - (uint32_t)f.rfoff;
			else if(f.rfoff < 0 && f.rflen > contig_len) f.rflen = contig_len;
			if(f.rflen == 0) AWA_RETURN(0);
			if(left) {
				uint32_t tmp_mm = 0, mm_tmp_numNs = 0;
				int min_rd_i = (int)f.rdoff, mm_min_rd_i = (int)f.rdoff;
				for(int rf_i = (int)f.rflen - 1; rf_i >= 0 && mm_min_rd_i >= 0; rf_i--, mm_min_rd_i--) {
					const int rf_bp = AWA_RF(f, rf_i), rd_bp = seq.at((uint32_t)mm_min_rd_i);
					if(rf_bp != rd_bp || rd_bp == 4) {
						if(tmp_mm == 0) min_rd_i = mm_min_rd_i;
						if(tmp_mm >= mm) break;
						tmp_mm++;
						h2g_edit e; e.pos = (uint32_t)mm_min_rd_i; e.chr = base_char(rf_bp); e.qchr = base_char(rd_bp); e.type = H2G_EDIT_MM; e.pad = 0; e.snp = H2G_MAX;
						AWA_PUSH_FRONT(e);
					}
					if(rf_bp == 4) { if(tmp_mm == 0) f.tmp_numNs++; mm_tmp_numNs++; }
				}
				if(tmp_mm == 0) min_rd_i = mm_min_rd_i;
				if(mm_min_rd_i < best_rdoff) { best_rdoff = mm_min_rd_i; AWA_COMMIT(); if(numNs) *numNs = mm_tmp_numNs; }
				if(mm_min_rd_i < 0) AWA_RETURN(f.rdlen);
				if(tmp_mm > 0) AWA_ERASE_FRONT(tmp_mm);
				f.min_rd_i = min_rd_i;
				f.a_first = 0; f.a_second = 0;
				if(A.n > 0) {
					uint32_t rd_diff = f.rdoff - (uint32_t)mm_min_rd_i;
					rd_diff = rd_diff > 16 ? rd_diff - 16 : 0;
					const uint32_t cpos = rd_diff >= f.joinedOff ? f.joinedOff : f.joinedOff - rd_diff;
					f.a_first = f.a_second = (int)alt_lobound(A, cpos);
					if(f.a_first >= (int)A.n) f.a_first = f.a_second = f.a_second - 1;
					for(; f.a_first >= 0; f.a_first--) {
						const DAlt alt = A.a[f.a_first];
						if(alt.type == H2G_ALT_SNP_SGL || alt.type == H2G_ALT_SNP_DEL || alt.type == H2G_ALT_SNP_INS) {
							if(alt.type == H2G_ALT_SNP_DEL && !(alt.seq & 0xff)) continue;
							if((uint64_t)alt.pos + f.rdlen < f.joinedOff) break;
						} else if(alt.type == H2G_ALT_SPLICESITE) {
							if(alt.pos < alt.len) continue;
							if((uint64_t)alt.pos + f.rdlen - 1 < f.joinedOff) break;
						} else continue;
					}
				}
#if H2G_HAPLOTYPE
				{   // "Update and find Haplotypes" :2898-2939
					HtList& L = W->ht[sp];
					L.n = 0;
					if(use_hap && H.n > 0) {
						if(sp > 0 && W->ntmp > 0 && W->tmp[0].type != H2G_EDIT_SPL && W->tmp[0].snp < A.n) {   // (a splice-site ALT carries no ALT id: the reference indexes past its list there)
							const HtList& P = W->ht[sp - 1];
							const DAlt taken = A.a[W->tmp[0].snp];
							for(uint32_t p = 0; p < P.n; p++) {
								const uint32_t o = H.first[P.ht[p]], hi_ = H.ids[o + P.idx[p]];
								if(hi_ >= A.n || !alt_is_same(taken, A.a[hi_])) continue;
								if(P.idx[p] == 0) add_haplotypes(A, H, f.joinedOff, L, f.rdlen, true, false, &h->overflow);
								else ht_push(L, P.ht[p], P.idx[p] - 1u, &h->overflow);
							}
						}
						if(L.n == 0) add_haplotypes(A, H, f.joinedOff, L, f.rdlen, true, sp == 0, &h->overflow);
					}
				}
#endif
				f.orig_nedits = W->ntmp;
				f.state = 1;
			} else {
				uint32_t tmp_mm = 0, max_rd_i = 0, mm_max_rd_i = 0, mm_tmp_numNs = 0;
				for(uint32_t rf_i = 0; rf_i < f.rflen && mm_max_rd_i < f.rdlen; rf_i++, mm_max_rd_i++) {
					const int rf_bp = AWA_RF(f, rf_i), rd_bp = seq.at(f.rdoff + mm_max_rd_i);
					if(rf_bp != rd_bp || rd_bp == 4) {
						if(tmp_mm == 0) max_rd_i = mm_max_rd_i;
						if(tmp_mm >= mm) break;
						tmp_mm++;
						h2g_edit e; e.pos = mm_max_rd_i + f.rdoff_add; e.chr = base_char(rf_bp); e.qchr = base_char(rd_bp); e.type = H2G_EDIT_MM; e.pad = 0; e.snp = H2G_MAX;
						AWA_PUSH_BACK(e);
					}
					if(rf_bp == 4) { if(tmp_mm == 0) f.tmp_numNs++; mm_tmp_numNs++; }
				}
				if(tmp_mm == 0) max_rd_i = mm_max_rd_i;
				if((int)(mm_max_rd_i + f.rdoff) > best_rdoff) { best_rdoff = (int)(mm_max_rd_i + f.rdoff); AWA_COMMIT(); if(numNs) *numNs = mm_tmp_numNs; W->ncand = 0; }
				else if((int)(mm_max_rd_i + f.rdoff) == best_rdoff) AWA_CAND_PUSH();
				if(mm_max_rd_i == f.rflen) AWA_RETURN(mm_max_rd_i);
				if(A.n == 0) AWA_RETURN(0);
				const uint32_t rd_diff = max_rd_i > 16 ? max_rd_i - 16 : 0;
				uint32_t a1 = alt_lobound(A, f.joinedOff + rd_diff), a2 = a1;
				if(a1 >= A.n) AWA_RETURN(0);
				for(; a2 < A.n; a2++) {
					const DAlt alt = A.a[a2];
					if(alt.type == H2G_ALT_SPLICESITE) { if(alt.pos > alt.len) continue; }
					if(alt.type == H2G_ALT_SNP_DEL) { if(alt.seq & 0xff) continue; }
					if(alt.pos > f.joinedOff + max_rd_i) break;
				}
				if(mm_max_rd_i == f.rdlen) {                            // the read ends here: only a forward splice site is worth trying (:3236-3245)
					bool further = false;
					for(uint32_t q = a1; q < a2; q++) if(A.a[q].type == H2G_ALT_SPLICESITE && A.a[q].pos < A.a[q].len) { further = true; break; }
					if(!further) AWA_RETURN(mm_max_rd_i);
				}
				if(tmp_mm > 0) W->ntmp -= tmp_mm;
				f.max_rd_i = max_rd_i;
				f.a_first = (int)a1; f.a_second = (int)a2;
#if H2G_HAPLOTYPE
				{   // "Update and find Haplotypes" :3251-3292
					HtList& L = W->ht[sp];
					L.n = 0;
					if(use_hap && H.n > 0) {
						if(sp > 0) {
							const HtList& P = W->ht[sp - 1];
							const uint32_t last = W->ntmp > 0 && W->tmp[W->ntmp - 1].type != H2G_EDIT_SPL ? W->tmp[W->ntmp - 1].snp : H2G_MAX;
							for(uint32_t p = 0; p < P.n; p++) {
								const uint32_t o = H.first[P.ht[p]], nal = H.first[P.ht[p] + 1] - o;
								if(P.idx[p] < nal) {
									const uint32_t hi_ = H.ids[o + P.idx[p]];
									if(last >= A.n || hi_ >= A.n || !alt_is_same(A.a[last], A.a[hi_])) continue;
								}
								if(P.idx[p] + 1u >= nal && f.joinedOff > H.right[P.ht[p]]) add_haplotypes(A, H, f.joinedOff, L, f.rdlen, false, false, &h->overflow);
								else ht_push(L, P.ht[p], P.idx[p] + 1u, &h->overflow);
							}
						}
						if(L.n == 0) add_haplotypes(A, H, f.joinedOff, L, f.rdlen, false, sp == 0 && f.rdoff_add == 0, &h->overflow);
					}
				}
#endif
				f.orig_nedits = W->ntmp;
				f.state = 1;
			}
		} else if(f.state == 2) {                                // back from the recursive call
			if(left) {
				if(ret == f.next_rdlen) AWA_RETURN(f.rdlen);
				if(f.orig_nedits < W->ntmp) AWA_ERASE_FRONT(W->ntmp - f.orig_nedits);
				f.a_second--;
			} else {
				if(ret > 0) {
					bool search_further = false;                          // :3519-3535
					if(f.cur_alt_type == H2G_ALT_SPLICESITE)
						for(int q = f.a_first + 1; q < f.a_second; q++) if(A.a[q].type == H2G_ALT_SPLICESITE && A.a[q].pos < A.a[q].len) { search_further = true; break; }
					if(!search_further && f.rd_i + ret == f.rdlen) AWA_RETURN(f.rd_i + ret);
				}
				if(f.orig_nedits < W->ntmp) W->ntmp = f.orig_nedits;
				f.a_first++;
			}
			f.state = 1;
		}
		// state 1: the loop over candidate ALTs
		if(left) {
			for(; f.a_second > f.a_first; f.a_second--) {
				DAlt alt = A.a[f.a_second];
				if(alt.pos >= f.joinedOff) continue;
				if(alt.type == H2G_ALT_SPLICESITE) {                   // the mirrored copy (left > right) serves the leftward walk (:2946-2951)
					if(alt.pos < alt.len) continue;
					const uint32_t t_ = alt.pos; alt.pos = alt.len; alt.len = t_;
				}
				if(alt.type == H2G_ALT_SNP_DEL) {
					if(!(alt.seq & 0xff)) continue;
					alt.pos = alt.pos - alt.len + 1;
				}
				if(alt.type == H2G_ALT_EXON) continue;
				bool alt_compatible = false;
				int rf_i = (int)f.rflen - 1, rd_i = (int)f.rdoff, diff = 0;
				if(alt.type == H2G_ALT_SNP_SGL) diff = (int)(f.joinedOff - alt.pos - 1);
				else if(alt.type == H2G_ALT_SNP_DEL) {
					if(alt.pos + alt.len >= f.joinedOff) continue;
					diff = (int)(f.joinedOff - (alt.pos + alt.len));
				} else if(alt.type == H2G_ALT_SNP_INS) diff = (int)(f.joinedOff - alt.pos);
				else if(alt.type == H2G_ALT_SPLICESITE) diff = (int)(f.joinedOff - (alt.len + 1));   // alt.len = right
				else continue;
				if(rf_i < diff || rd_i < diff) continue;
				rf_i -= diff; rd_i -= diff;
				int rd_bp = seq.at((uint32_t)rd_i);
				if(rd_i < f.min_rd_i) {
					if(alt.type == H2G_ALT_SNP_INS) { if(rd_i + 1 >= f.min_rd_i) continue; }
					break;
				}
#if H2G_HAPLOTYPE
				if(W->ht[sp].n > 0 && alt.type != H2G_ALT_SPLICESITE && !ht_supports(A, H, W->ht[sp], (uint32_t)f.a_second)) continue;
#endif
				if(alt.type == H2G_ALT_SNP_SGL) {
					if(rd_bp == (int)alt.seq) {
						const int rf_bp = AWA_RF(f, rf_i);
						h2g_edit e; e.pos = (uint32_t)rd_i; e.chr = base_char(rf_bp); e.qchr = base_char(rd_bp); e.type = H2G_EDIT_MM; e.pad = 0; e.snp = (uint32_t)f.a_second;
						AWA_PUSH_FRONT(e);
						rd_i--; rf_i--;
						alt_compatible = true;
					}
				} else if(alt.type == H2G_ALT_SNP_DEL) {
					if(f.rfoff + rf_i > (int)alt.len) {
						// the "long deletion" refetch of the reference (:2986-3008) reads the same text positions: rfoff + rf_i - i
						for(uint32_t i = 0; i < alt.len; i++) {
							const int rf_bp = AWA_RF(f, rf_i - (int)i);
							h2g_edit e; e.pos = (uint32_t)(rd_i + 1); e.chr = base_char(rf_bp); e.qchr = '-'; e.type = H2G_EDIT_READ_GAP; e.pad = 0; e.snp = (uint32_t)f.a_second;
							AWA_PUSH_FRONT(e);
						}
						rf_i -= (int)alt.len;
						alt_compatible = true;
					}
				} else {
					if(rd_i > (int)alt.len) {
						bool same_seq = true;
						for(uint32_t i = 0; i < alt.len; i++) {
							rd_bp = seq.at((uint32_t)(rd_i - (int)i));
							const int snp_bp = (int)((alt.seq >> (i << 1)) & 3);
							if(rd_bp != snp_bp) { same_seq = false; break; }
							h2g_edit e; e.pos = (uint32_t)(rd_i - (int)i); e.chr = '-'; e.qchr = base_char(rd_bp); e.type = H2G_EDIT_REF_GAP; e.pad = 0; e.snp = (uint32_t)f.a_second;
							AWA_PUSH_FRONT(e);
						}
						if(same_seq) { rd_i -= (int)alt.len; alt_compatible = true; }
					}
				}
				uint32_t next_joined = alt.pos;
				int splice_shift = 0;
				if(alt.type == H2G_ALT_SPLICESITE) {                    // :3083-3105
					alt_compatible = false;
					if(!(rd_i == (int)f.rdoff && f.prev_alt_type == H2G_ALT_SPLICESITE)) {
						const uint32_t intronLen = alt.len - alt.pos + 1;
						AWA_PUSH_FRONT(make_spl_edit((uint32_t)(rd_i + 1), intronLen, (alt.seq & 0xff) ? H2G_SPL_FW : H2G_SPL_RC, true, 0.0f));
						alt_compatible = true;
						next_joined = alt.pos; splice_shift = (int)intronLen;   // next_joinedOff = alt.left; the window moves across the intron
					}
				}
				if(alt_compatible) {
					numALTsTried++;
					if(rd_i < 0) { best_rdoff = rd_i; AWA_COMMIT(); AWA_RETURN(f.rdlen); }
					int next_rfoff = f.rfoff - splice_shift, next_rflen = rf_i + 1, next_rdlen = rd_i + 1;
					if(next_rflen < next_rdlen) {
						int add_len = next_rdlen + 10 - next_rflen;
						if(next_rfoff < add_len) add_len = next_rfoff;
						next_rfoff -= add_len; next_rflen += add_len;
					}
					if(sp + 1 >= H2G_AWA_DEPTH) { h->overflow = 1; }
					else {
						AwaFrame& nf = W->fr[sp + 1];
						nf.joinedOff = next_joined; nf.rdoff_add = f.rdoff_add; nf.rdoff = (uint32_t)rd_i; nf.rdlen = (uint32_t)next_rdlen;
						nf.rfoff = next_rfoff; nf.rflen = (uint32_t)next_rflen; nf.tmp_numNs = f.tmp_numNs; nf.dep = f.dep + 1; nf.state = 0;
						nf.prev_alt_type = alt.type; nf.cur_alt_type = 0;
						f.next_rdlen = (uint32_t)next_rdlen;
						f.state = 2;
						sp++;
						goto next_frame;
					}
				}
				if(f.orig_nedits < W->ntmp) AWA_ERASE_FRONT(W->ntmp - f.orig_nedits);
			}
			AWA_RETURN(0);
		} else {
			for(; f.a_first < f.a_second; f.a_first++) {
				const DAlt alt = A.a[f.a_first];
				if(alt.type == H2G_ALT_SPLICESITE) { if(alt.pos > alt.len) continue; }   // forward copies only (left < right)
				if(alt.type == H2G_ALT_EXON) continue;
				if(alt.type == H2G_ALT_SNP_DEL) { if(alt.seq & 0xff) continue; }
				bool alt_compatible = false;
				uint32_t rf_i, rd_i;
				rf_i = rd_i = alt.pos - f.joinedOff;
				if(rd_i >= f.rdlen) continue;
				int rf_bp = AWA_RF(f, rf_i), rd_bp = seq.at(f.rdoff + rd_i);
#if H2G_HAPLOTYPE
				if(W->ht[sp].n > 0 && alt.type != H2G_ALT_SPLICESITE && !ht_supports(A, H, W->ht[sp], (uint32_t)f.a_first)) continue;
#endif
				if(alt.type == H2G_ALT_SNP_SGL) {
					if(rd_bp == (int)alt.seq) {
						h2g_edit e; e.pos = rd_i + f.rdoff_add; e.chr = base_char(rf_bp); e.qchr = base_char(rd_bp); e.type = H2G_EDIT_MM; e.pad = 0; e.snp = (uint32_t)f.a_first;
						AWA_PUSH_BACK(e);
						rd_i++; rf_i++;
						alt_compatible = true;
					}
				} else if(alt.type == H2G_ALT_SNP_DEL) {
					bool try_del = rd_i > 0;
					if(rd_i == 0 && f.dep > 0) { if(W->ntmp > 0 && W->tmp[W->ntmp - 1].type != H2G_EDIT_READ_GAP) try_del = true; }
					if(try_del) {
						// (:3352-3392: the long-deletion refetch reads the same text positions rfoff + rf_i + i)
						for(uint32_t i = 0; i < alt.len; i++) {
							rf_bp = AWA_RF(f, rf_i + i);
							h2g_edit e; e.pos = rd_i + f.rdoff_add; e.chr = base_char(rf_bp); e.qchr = '-'; e.type = H2G_EDIT_READ_GAP; e.pad = 0; e.snp = (uint32_t)f.a_first;
							AWA_PUSH_BACK(e);
						}
						rf_i += alt.len;
						alt_compatible = true;
					}
				} else if(alt.type == H2G_ALT_SNP_INS) {
					if(rd_i + alt.len <= f.rdlen && rf_i > 0) {
						bool same_seq = true;
						for(uint32_t i = 0; i < alt.len; i++) {
							rd_bp = seq.at(f.rdoff + rd_i + i);
							const int snp_bp = (int)((alt.seq >> ((alt.len - i - 1) << 1)) & 3);
							if(rd_bp != snp_bp) { same_seq = false; break; }
							h2g_edit e; e.pos = rd_i + i + f.rdoff_add; e.chr = '-'; e.qchr = base_char(rd_bp); e.type = H2G_EDIT_REF_GAP; e.pad = 0; e.snp = (uint32_t)f.a_first;
							AWA_PUSH_BACK(e);
						}
						if(same_seq) { rd_i += alt.len; alt_compatible = true; }
					}
				} else if(alt.type == H2G_ALT_SPLICESITE) {             // :3425-3449
					bool try_splice = rd_i > 0;
					if(rd_i == 0 && f.dep > 0) { if(W->ntmp > 0 && W->tmp[W->ntmp - 1].type != H2G_EDIT_SPL) try_splice = true; }   // no consecutive introns
					if(try_splice) {
						AWA_PUSH_BACK(make_spl_edit(rd_i + f.rdoff_add, alt.len - alt.pos + 1, (alt.seq & 0xff) ? H2G_SPL_FW : H2G_SPL_RC, true, 0.0f));
						alt_compatible = true;
					}
				}
				if(alt_compatible) {
					numALTsTried++;
					if(rd_i == f.rdlen) {
						if(best_rdoff < (int)(f.rdoff + rd_i)) W->ncand = 0;
						AWA_CAND_PUSH();
						best_rdoff = (int)(f.rdoff + rd_i); AWA_COMMIT(); AWA_RETURN(rd_i);
					}
					uint32_t next_joinedOff;
					uint32_t next_rflen = f.rflen - rf_i;
					const uint32_t next_rdlen = f.rdlen - rd_i;
					if(alt.type == H2G_ALT_SNP_SGL) next_joinedOff = alt.pos + 1;
					else if(alt.type == H2G_ALT_SNP_DEL) { next_joinedOff = alt.pos + alt.len; if(f.rflen <= rf_i) next_rflen = 0; }
					else if(alt.type == H2G_ALT_SPLICESITE) next_joinedOff = alt.len + 1;            // alt.right + 1
					else next_joinedOff = alt.pos;
					const int splice_shift = alt.type == H2G_ALT_SPLICESITE ? (int)(alt.len - alt.pos + 1) : 0;
					if(next_rflen < next_rdlen) next_rflen = next_rdlen + 10;
					if(sp + 1 >= H2G_AWA_DEPTH) { h->overflow = 1; }
					else {
						AwaFrame& nf = W->fr[sp + 1];
						nf.joinedOff = next_joinedOff; nf.rdoff_add = f.rdoff_add + rd_i; nf.rdoff = f.rdoff + rd_i; nf.rdlen = next_rdlen;
						nf.rfoff = f.rfoff + (int)rf_i + splice_shift; nf.rflen = next_rflen; nf.tmp_numNs = f.tmp_numNs; nf.dep = f.dep + 1; nf.state = 0;
						nf.prev_alt_type = alt.type; nf.cur_alt_type = 0;
						f.rd_i = rd_i; f.cur_alt_type = alt.type;
						f.state = 2;
						sp++;
						goto next_frame;
					}
				}
				if(f.orig_nedits < W->ntmp) W->ntmp = f.orig_nedits;
			}
			AWA_RETURN(0);
		}
		}
	next_frame:;
	}
#undef AWA_RF
#undef AWA_PUSH_FRONT
#undef AWA_PUSH_BACK
#undef AWA_ERASE_FRONT
#undef AWA_COMMIT
#undef AWA_RETURN
#undef AWA_CAND_PUSH
	// alignWithALTs :741-783
	uint32_t extlen = left ? rdoff0 - (uint32_t)best_rdoff : (uint32_t)best_rdoff - rdoff0;
	const uint32_t ne = h->nedits;
	if(extlen > 0 && ne > 0) {
		const h2g_edit f = h->edits[0];
		if(f.pos + extlen == base_rdoff + 1) {
			if(is_gap(f.type) || f.type == H2G_EDIT_SPL) extlen = 0;       // :758-763 (the front test covers splices, the back test gaps only)
			if(f.type == H2G_EDIT_MM && f.chr == 'N') extlen = 0;
		}
		const h2g_edit b = h->edits[ne - 1];
		if(extlen > 0 && b.pos == rdoff0 - base_rdoff + extlen - 1) {
			if(is_gap(b.type)) extlen = 0;
		}
		if(extlen == 0 && ne > nedits0) {
			if(left) for(uint32_t k = 0; k < nedits0; k++) h->edits[k] = h->edits[k + (ne - nedits0)];
			h->nedits = nedits0;
		}
	}
	return extlen;
}

// GenomeHit::extend hi_aligner.h:2031-2232 on a graph index (same bookkeeping as extend_item, ALT-aware alignment)
H2G_HDN bool extend_item_alts(const DRef& ref, const DAlts& A, const DScoring& sc, const SeqView& seq, h2g_ghit* h, uint32_t mm,
                             uint32_t max_leftext, uint32_t max_rightext, uint32_t* leftext, uint32_t* rightext, AwaWS* W)
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
		// no ALT within reach of this extension (the read cannot get further than rdoff reference bases to the left without one):
		// alignWithALTs_recur then degenerates to its mismatch scan — the linear-index code path, without the frame stack
		const uint32_t wlo = h->joinedOff > h->rdoff + 16 ? h->joinedOff - h->rdoff - 16 : 0;
		const bool no_alt = alt_lobound(A, wlo) == alt_lobound(A, h->joinedOff + 2);
		const uint32_t best_ext = no_alt ? align_no_alts(ref, seq, h->rdoff - 1, h->rdoff - 1, h->rdoff, h->tidx, rl, reflen, true, h, mm, &numNs)
		                                 : align_with_alts(ref, A, seq, h->joinedOff, h->rdoff - 1, h->rdoff - 1, h->rdoff, h->tidx, rl, reflen, true, h, mm, &numNs, W);
		if(h->len == 0 && mm == 0 && h->nedits > 0) { h->nedits = 0; return false; }
		if(best_ext > 0) {
			*leftext = best_ext;
			const uint32_t added = h->nedits - n_prev;
			int ref_ext = (int)best_ext;
			for(uint32_t i = 0; i < added; i++) {
				if(h->edits[i].type == H2G_EDIT_REF_GAP) ref_ext--;
				else if(h->edits[i].type == H2G_EDIT_READ_GAP) ref_ext++;
				else if(h->edits[i].type == H2G_EDIT_SPL) ref_ext += (int)spl_len(h->edits[i]);     // hi_aligner.h:2121
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
			int ref_ext = (int)h->len;
			for(uint32_t ei = 0; ei < h->nedits; ei++) {
				const h2g_edit e = h->edits[ei];
				if(e.type == H2G_EDIT_REF_GAP) ref_ext--;
				else if(e.type == H2G_EDIT_READ_GAP) ref_ext++;
				else if(e.type == H2G_EDIT_SPL) ref_ext += (int)spl_len(e);                        // :2162
				else if(e.type == H2G_EDIT_MM && e.chr == 'N') ref_ext--;
			}
			const uint32_t jr = h->joinedOff + (uint32_t)ref_ext;
			const bool no_alt = alt_lobound(A, jr > 2 ? jr - 2 : 0) == alt_lobound(A, jr + rr + 16);
			const uint32_t best_ext = no_alt ? align_no_alts(ref, seq, h->rdoff, h->rdoff + h->len, rdlen - (h->rdoff + h->len), h->tidx, (int)rl, reflen, false, h, mm, nullptr)
			                                 : align_with_alts(ref, A, seq, jr, h->rdoff, h->rdoff + h->len,
			                                                   rdlen - (h->rdoff + h->len), h->tidx, (int)rl, reflen, false, h, mm, nullptr, W);
			if(h->len == 0 && mm == 0 && h->nedits > 0) { h->nedits = 0; return false; }
			if(best_ext > 0) { *rightext = best_ext; h->len += best_ext; }
		}
	}
	calculate_score(sc, seq, h);
	return *leftext > 0 || *rightext > 0;
}

// findOffDiffs (hi_aligner.h:2545-2640): offset differences that indel ALTs inside [start, end) can introduce.
// od[k] = {|off|, sign}; returns the number of single-ALT entries (the combinations follow them).
struct OffDiff { uint32_t first; int32_t second; };
#ifndef H2G_OFFDIFF_CAP
#define H2G_OFFDIFF_CAP 32
#endif
H2G_HD bool alt_is_gap_fw(const DAlt& a) { return (a.type == H2G_ALT_SNP_DEL && !(a.seq & 0xff)) || a.type == H2G_ALT_SNP_INS; }
H2G_HDN uint32_t find_off_diffs(const DAlts& A, uint32_t start, uint32_t end, OffDiff* od, uint32_t* nod, uint32_t* overflow) {
	uint32_t n = 0;
	od[n].first = 0; od[n].second = 0; n++;
	*nod = n;
	uint32_t a1 = alt_lobound(A, start), a2 = a1;
	for(; a2 < A.n; a2++) {
		const DAlt alt = A.a[a2];
		if(alt.type == H2G_ALT_SPLICESITE && alt.pos > alt.len) continue;
		if(alt.type == H2G_ALT_SNP_DEL && (alt.seq & 0xff)) continue;
		if(alt.pos >= end) break;
	}
	if(a1 >= a2) return n;
	for(uint32_t s2 = a2; s2 > a1; s2--) {
		const DAlt alt = A.a[s2 - 1];
		if(!alt_is_gap_fw(alt)) continue;
		const int off = alt.type == H2G_ALT_SNP_DEL ? (int)alt.len : -(int)alt.len;
		if(n < H2G_OFFDIFF_CAP) { od[n].first = (uint32_t)(off < 0 ? -off : off); od[n].second = off > 0 ? 1 : -1; n++; } else *overflow = 1;
	}
	if(n > 1) {   // sort + unique
		for(uint32_t i = 1; i < n; i++) {
			const OffDiff x = od[i];
			int j = (int)i - 1;
			while(j >= 0 && (x.first != od[j].first ? x.first < od[j].first : x.second < od[j].second)) { od[j + 1] = od[j]; j--; }
			od[j + 1] = x;
		}
		uint32_t w = 1;
		for(uint32_t i = 1; i < n; i++) if(od[i].first != od[w - 1].first || od[i].second != od[w - 1].second) od[w++] = od[i];
		n = w;
	}
	const uint32_t single = n;
	for(uint32_t s2 = a2; s2 > a1; s2--) {
		const DAlt alt = A.a[s2 - 1];
		if(!alt_is_gap_fw(alt)) continue;
		int off = alt.type == H2G_ALT_SNP_DEL ? (int)alt.len : -(int)alt.len;
		for(uint32_t s3 = s2 - 1; s3 > a1; s3--) {
			const DAlt alt2 = A.a[s3 - 1];
			if(!alt_is_gap_fw(alt2)) continue;
			if(alt2.type == H2G_ALT_SNP_DEL) { if(alt2.pos + alt2.len >= alt.pos) continue; off += (int)alt2.len; }
			else { if(alt2.pos >= alt.pos) continue; off -= (int)alt2.len; }
			bool found = false;
			for(uint32_t i = 0; i < n; i++) if(off == (int)od[i].first * od[i].second) { found = true; break; }
			if(!found) { if(n < H2G_OFFDIFF_CAP) { od[n].first = (uint32_t)(off < 0 ? -off : off); od[n].second = off > 0 ? 1 : -1; n++; } else *overflow = 1; }
		}
	}
	*nod = n;
	return single;
}

// GenomeHit::operator== (hi_aligner.h:1156-1183)
H2G_HD bool ghit_equal(const h2g_ghit* a, const h2g_ghit* b) {
	if(a->fw != b->fw || a->rdoff != b->rdoff || a->len != b->len || a->tidx != b->tidx || a->toff != b->toff || a->trim5 != b->trim5 ||
	   a->trim3 != b->trim3 || a->nedits != b->nedits) return false;
	for(uint32_t i = 0; i < a->nedits; i++) {
		const h2g_edit e = a->edits[i], o = b->edits[i];
		if(e.type == H2G_EDIT_READ_GAP) { if(o.type != H2G_EDIT_READ_GAP) return false; }
		else if(e.type == H2G_EDIT_REF_GAP) { if(o.type != H2G_EDIT_REF_GAP) return false; }
		else if(e.type != o.type || e.pos != o.pos || e.chr != o.chr || e.qchr != o.qchr) return false;
	}
	return true;
}

// GenomeHit::findSSOffs (hi_aligner.h:2482-2540): by how much a coordinate found near splice-site ALTs may be off — an anchor
// that ran through a splice edge of the graph is reported at the far side of the intron.  (0, 0) first, then sorted, unique.
H2G_HDN uint32_t find_ss_offs(const DGfm& g, const DAlts& A, uint32_t start, uint32_t end, OffDiff* so, uint32_t* overflow) {
	uint32_t n = 0;
	so[n].first = 0; so[n].second = 0; n++;
#if H2G_HAPLOTYPE
	if(g.linear || !(A.has_splice & 1u)) return n;
#else
	if(g.linear || !A.has_splice) return n;
#endif
	auto push = [&](uint32_t first, int second) { if(n < H2G_OFFDIFF_CAP) { so[n].first = first; so[n].second = second; n++; } else *overflow = 1; };
	for(uint32_t i = alt_lobound(A, start); i < A.n; i++) {
		const DAlt alt = A.a[i];
		if(alt.pos >= end) break;
		if(alt.type != H2G_ALT_SPLICESITE) continue;
		if(alt.pos < alt.len) {                                  // left < right
			push(alt.len - alt.pos + 1, 1);
			const uint32_t relax = 5;
			const uint32_t from = alt.len > relax ? alt.len - relax : 0;
			for(uint32_t j = alt_lobound(A, from); j < A.n; j++) {
				const DAlt alt2 = A.a[j];
				if(alt2.type != H2G_ALT_SPLICESITE) continue;
				if(alt2.pos < alt2.len) continue;
				if((uint64_t)alt2.pos + alt2.len == (uint64_t)alt.pos + alt.len) continue;
				if(alt2.pos > alt.len + relax) break;
				if(alt2.len < alt.pos) push(alt.pos - alt2.len, -1);
				else push(alt2.len - alt.pos, 1);
			}
		} else push(alt.pos - alt.len + 1, -1);
	}
	if(n > 1) {   // sort (pair order: first, then second) + unique
		for(uint32_t i = 1; i < n; i++) {
			const OffDiff x = so[i];
			int j = (int)i - 1;
			while(j >= 0 && (x.first != so[j].first ? x.first < so[j].first : x.second < so[j].second)) { so[j + 1] = so[j]; j--; }
			so[j + 1] = x;
		}
		uint32_t w = 1;
		for(uint32_t i = 1; i < n; i++) if(so[i].first != so[w - 1].first || so[i].second != so[w - 1].second) so[w++] = so[i];
		n = w;
	}
	return n;
}

// static GenomeHit::adjustWithALT (hi_aligner.h:2239-2390) as getAnchorHits calls it (:5175): for every splice-site offset
// (findSSOffs; the single (0, 0) without splice-site ALTs) and every indel offset (findOffDiffs) the hit is re-seated until the
// ALT-aware comparison covers it.  Appends to hits[*nhits .. cap); returns whether any hit was added.
H2G_HDN bool adjust_with_alt(const DGfm& g, const DRef& ref, const DAlts& A, const SeqView& seq, uint32_t rdoff, uint32_t len,
                            uint32_t tidx, uint32_t toff0, uint32_t joinedOff0, h2g_ghit* hits, uint32_t* nhits, uint32_t cap,
                            AwaWS* W, uint32_t* overflow)
{
	const uint32_t n0 = *nhits;
	if(g.linear) {
		if(*nhits >= cap) { *overflow = 1; return false; }
		h2g_ghit* gh = &hits[*nhits];
		gh->read = 1;   // _hitcount
		gh->fw = seq.fw; gh->rdoff = rdoff; gh->len = len; gh->trim5 = 0; gh->trim3 = 0; gh->tidx = tidx; gh->toff = toff0; gh->joinedOff = joinedOff0;
		gh->score = 0; gh->nedits = 0; gh->overflow = 0; gh->splicescore = 0;
		(*nhits)++;
		return true;
	}
	const uint32_t width = 1u << (g.offRate + 2);
	OffDiff so[H2G_OFFDIFF_CAP];
	const uint32_t nso = find_ss_offs(g, A, joinedOff0 >= width ? joinedOff0 - width : 0, joinedOff0 + width, so, overflow);
	for(uint32_t si = 0; si < nso; si++) {
		uint32_t toff = toff0, joinedOff = joinedOff0;
		if(so[si].first > 0) {
			if(so[si].second > 0) { toff += so[si].first; joinedOff += so[si].first; }
			else { toff -= so[si].first; joinedOff -= so[si].first; }
		}
		if(*nhits >= cap) { *overflow = 1; break; }
		h2g_ghit* gh = &hits[*nhits];
		gh->read = 1;   // _hitcount
		gh->fw = seq.fw; gh->rdoff = rdoff; gh->len = len; gh->trim5 = 0; gh->trim3 = 0; gh->tidx = tidx; gh->toff = toff; gh->joinedOff = joinedOff;
		gh->score = 0; gh->nedits = 0; gh->overflow = 0; gh->splicescore = 0;
		(*nhits)++;
		OffDiff od[H2G_OFFDIFF_CAP];
		uint32_t nod = 0;
		const uint32_t single = find_off_diffs(A, joinedOff >= width ? joinedOff - width : 0, joinedOff + width, od, &nod, overflow);
		const uint32_t max_od = (A.maxAltsTried / 4) > 4 ? (A.maxAltsTried / 4) : 4;
		if(nod - single > max_od) nod = single + max_od;
		bool found2 = false;
		for(uint32_t o = 0; o < nod && !found2; o++) {
			if(od[o].second >= 0) { gh->joinedOff = joinedOff + od[o].first; gh->toff = toff + od[o].first; }
			else { if(toff < od[o].first) continue; gh->joinedOff = joinedOff - od[o].first; gh->toff = toff - od[o].first; }
			gh->nedits = 0;
			const uint32_t alignedLen = align_with_alts(ref, A, seq, gh->joinedOff, gh->rdoff, gh->rdoff, gh->len, gh->tidx, (int)gh->toff, gh->len + 10,
			                                            false, gh, 0, nullptr, W, true);
			if(gh->overflow) *overflow = 1;
			if(alignedLen == gh->len) {
				found2 = true;
				for(uint32_t i = 0; i + 1 < *nhits; i++) if(ghit_equal(&hits[i], gh)) found2 = false;
				if(found2) {
					for(uint32_t e = 0; e < W->ncand; e++) {
						if(*nhits >= cap) { *overflow = 1; break; }
						h2g_ghit* c = &hits[*nhits];
						const h2g_ghit* prev = &hits[*nhits - 1];
						c->read = prev->read; c->fw = prev->fw; c->rdoff = prev->rdoff; c->len = prev->len; c->trim5 = prev->trim5; c->trim3 = prev->trim3;
						c->tidx = prev->tidx; c->toff = prev->toff; c->joinedOff = prev->joinedOff; c->score = prev->score; c->overflow = 0; c->splicescore = 0;
						c->nedits = W->cand_n[e];
						for(uint32_t q = 0; q < c->nedits; q++) c->edits[q] = W->cand[e][q];
						(*nhits)++;
						for(uint32_t i = 0; i + 1 < *nhits; i++) if(ghit_equal(&hits[i], c)) { (*nhits)--; break; }
					}
				}
			} else gh->nedits = 0;
		}
		if(!found2) {   // genomeHits.pop_back(): the entry this offset opened (candidates are only appended when it was kept)
			(*nhits)--;
		}
	}
	return *nhits > n0;
}

// member GenomeHit::adjustWithALT (hi_aligner.h:2395-2476): re-seat an already initialised hit; false = no offset works
H2G_HDN bool adjust_with_alt_member(const DGfm& g, const DRef& ref, const DAlts& A, const SeqView& seq, h2g_ghit* gh, AwaWS* W, uint32_t* overflow) {
	if(g.linear) return true;
	const uint32_t width = 1u << (g.offRate + 2);
	OffDiff od[H2G_OFFDIFF_CAP];
	uint32_t nod = 0;
	const uint32_t single = find_off_diffs(A, gh->joinedOff >= width ? gh->joinedOff - width : 0, gh->joinedOff + width, od, &nod, overflow);
	const uint32_t max_od = (A.maxAltsTried / 4) > 4 ? (A.maxAltsTried / 4) : 4;
	if(nod - single > max_od) nod = single + max_od;
	const uint32_t orig_joinedOff = gh->joinedOff, orig_toff = gh->toff;
	bool found = false;
	for(uint32_t o = 0; o < nod && !found; o++) {
		if(od[o].second >= 0) { gh->joinedOff = orig_joinedOff + od[o].first; gh->toff = orig_toff + od[o].first; }
		else { if(orig_toff < od[o].first) continue; gh->joinedOff = orig_joinedOff - od[o].first; gh->toff = orig_toff - od[o].first; }
		const uint32_t alignedLen = align_with_alts(ref, A, seq, gh->joinedOff, gh->rdoff, gh->rdoff, gh->len, gh->tidx, (int)gh->toff, gh->len + 10,
		                                            false, gh, 0, nullptr, W, true);
		if(gh->overflow) *overflow = 1;
#if defined(H2G_TRACE) && !defined(__HIP_DEVICE_COMPILE__)
		fprintf(stderr, "      adjust(member) off %u/%d joff %u toff %u len %u -> aligned %u nedits %u\n", od[o].first, od[o].second, gh->joinedOff, gh->toff, gh->len, alignedLen, gh->nedits);
#endif
		if(alignedLen == gh->len) found = true;
		else gh->nedits = 0;
	}
	return found;
}

}  // namespace h2g
