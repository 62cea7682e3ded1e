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
#include "h2g_core.h"

namespace h2g {

#define H2G_GSIDE_SYMS 208u

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

H2G_HD bool is_zoff(const DGfm& g, uint32_t row) {   // GFM::_zOffs (gfm.h:2783); a handful of entries at most
	if(g.nZ == 0) return false;
	if(row == g.zoff) return true;
	for(uint32_t i = 1; i < g.nZ; i++) if(g.zoffs[i] == row) return true;
	return false;
}

// countBt2Side (gfm.h:2958-3001) on a loaded graph side; charOff = row % 208
H2G_HD uint32_t rank_in_side128(const DGfm& g, const Side128& s, uint32_t sideNum, uint32_t charOff, int c) {
	uint32_t cnt = 0;
#pragma unroll
	for(int k = 0; k < 7; k++) cnt += count_word(s.w[k], c, (int)charOff - 32 * k);   // word 6: symbols 192..207 only
	if(c == 0 && g.nZ) {                                  // '$' rows are stored as 'A' (gfm.h:2967-2979)
		for(uint32_t i = 0; i < g.nZ; i++) {
			const uint32_t z = i == 0 ? g.zoff : g.zoffs[i];
			const uint32_t zs = z / H2G_GSIDE_SYMS, zc = z - zs * H2G_GSIDE_SYMS;
			if(zs == sideNum && zc < charOff) cnt--;
		}
	}
	const uint64_t ow = (c & 2) ? s.w[15] : s.w[14];
	const uint32_t occ = (c & 1) ? (uint32_t)(ow >> 32) : (uint32_t)ow;
	const uint32_t fc = c == 0 ? g.fchr[0] : c == 1 ? g.fchr[1] : c == 2 ? g.fchr[2] : g.fchr[3];
	return occ + cnt + fc;
}

H2G_HD int rowL_in_side128(const Side128& s, uint32_t charOff) {   // rowL gfm.h:3615
	const uint32_t k = charOff >> 5;
	uint64_t w = s.w[0];
#pragma unroll
	for(int j = 1; j < 7; j++) w = (k == (uint32_t)j) ? s.w[j] : w;
	return (int)((w >> ((charOff & 31) * 2)) & 3);
}

H2G_HD uint32_t rank128(const DGfm& g, uint32_t row, int c) {      // SideLocus::initFromRow + mapLF gfm.h:3712
	const uint32_t sideNum = row / H2G_GSIDE_SYMS, charOff = row - sideNum * H2G_GSIDE_SYMS;
	Side128 s = load_side128(g.sides + (size_t)sideNum * 128);
	return rank_in_side128(g, s, sideNum, charOff, c);
}

// ---- bit vectors ------------------------------------------------------------------------------------------
// Both bit vectors are read with aligned dword loads: F starts at byte 52 (dword 13), M at byte 78 (the
// upper half of dword 19).  208 bits = 3 full u64 + 16 bits.
struct Bits208 { uint64_t w[4]; };

H2G_HD Bits208 load_F(const uint8_t* side) {
	const uint32_t* q = reinterpret_cast<const uint32_t*>(side + 52);
	Bits208 b;
	b.w[0] = q[0] | ((uint64_t)q[1] << 32);
	b.w[1] = q[2] | ((uint64_t)q[3] << 32);
	b.w[2] = q[4] | ((uint64_t)q[5] << 32);
	b.w[3] = q[6] & 0xffffu;
	return b;
}
H2G_HD Bits208 load_M(const uint8_t* side, uint32_t* F_loc, uint32_t* M_occ) {
	const uint32_t* q = reinterpret_cast<const uint32_t*>(side + 76);   // dwords 19..27
	uint32_t a[7];
#pragma unroll
	for(int k = 0; k < 7; k++) a[k] = q[k];
	Bits208 b;
	b.w[0] = (a[0] >> 16) | ((uint64_t)a[1] << 16) | ((uint64_t)a[2] << 48);
	b.w[1] = (a[2] >> 16) | ((uint64_t)a[3] << 16) | ((uint64_t)a[4] << 48);
	b.w[2] = (a[4] >> 16) | ((uint64_t)a[5] << 16) | ((uint64_t)a[6] << 48);
	b.w[3] = a[6] >> 16;
	*F_loc = q[7];
	*M_occ = q[8];
	return b;
}
H2G_HD uint64_t low_mask(int n) { return n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull)); }

// rank_M (gfm.h:4100) = countMSide (:3146): ones of M strictly before `row`, plus the side's M_occ
H2G_HD uint32_t rank_M(const DGfm& g, uint32_t row) {
	const uint32_t sideNum = row / H2G_GSIDE_SYMS, off = row - sideNum * H2G_GSIDE_SYMS;
	uint32_t F_loc, M_occ;
	Bits208 m = load_M(g.sides + (size_t)sideNum * 128, &F_loc, &M_occ);
	uint32_t cnt = M_occ;
#pragma unroll
	for(int k = 0; k < 4; k++) cnt += (uint32_t)__builtin_popcountll(m.w[k] & low_mask((int)off - 64 * k));
	return cnt;
}

H2G_HD uint32_t select_in_word(uint64_t w, uint32_t count) {   // position of the count-th (1-based) set bit
	for(uint32_t i = 1; i < count; i++) w &= w - 1;
	return (uint32_t)__builtin_ctzll(w);
}

// select_F (gfm.h:4113-4167): row of the count-th F one at or after `row` (count >= 1), crossing sides as needed
H2G_HD uint32_t select_F(const DGfm& g, uint32_t row, uint32_t count) {
	uint32_t sideNum = row / H2G_GSIDE_SYMS, off = row - sideNum * H2G_GSIDE_SYMS;
	const uint32_t lastSide = (g.gbwtLen - 1) / H2G_GSIDE_SYMS;
	while(true) {
		Bits208 f = load_F(g.sides + (size_t)sideNum * 128);
#pragma unroll
		for(int k = 0; k < 4; k++) {
			const uint64_t w = f.w[k] & ~low_mask((int)off - 64 * k);
			const uint32_t pc = (uint32_t)__builtin_popcountll(w);
			if(count <= pc) return sideNum * H2G_GSIDE_SYMS + 64u * k + select_in_word(w, count);
			count -= pc;
		}
		if(sideNum >= lastSide) return g.gbwtLen;   // not reachable on a well-formed index (the reference would run off the array)
		sideNum++;
		off = 0;
	}
}

// F-row of node `node`: backward scan over the (F_loc, M_occ) side headers starting at the side of `locRow`
// (mapGLF gfm.h:3788-3810, mapGLF1 :3978-3998).  Returns the scan's F_loc (already +1 when M_occ > 0) and M_occ.
H2G_HD uint32_t node_to_Frow(const DGfm& g, uint32_t locRow, uint32_t node, uint32_t* F_loc_out, uint32_t* M_occ_out) {
	uint32_t sideNum = locRow / H2G_GSIDE_SYMS;
	uint32_t F_loc, M_occ;
	while(true) {
		const uint32_t* q = reinterpret_cast<const uint32_t*>(g.sides + (size_t)sideNum * 128 + 104);
		F_loc = q[0];
		M_occ = q[1];
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
H2G_HD void in_edge_count(const DGfm& g, uint32_t top, uint32_t bot, IEdges* ie) {
	ie->n = 0;
	uint32_t curr_node = 0, num0s = 0;
	uint32_t sideNum = H2G_MAX;
	Bits208 f;
	f.w[0] = f.w[1] = f.w[2] = f.w[3] = 0;
	for(uint32_t row = top + 1; row < bot; row++) {
		const uint32_t sn = row / H2G_GSIDE_SYMS, off = row - sn * H2G_GSIDE_SYMS;
		if(sn != sideNum) { sideNum = sn; f = load_F(g.sides + (size_t)sn * 128); }
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

// mapGLF (gfm.h:3759-3837): LF of a row range + translation of the outgoing-edge rows back to incoming rows
// through M-rank / F-select.  false = empty range.  `ie` may be null.
H2G_HD bool map_glf(const DGfm& g, uint32_t top, uint32_t bot, int c, uint32_t k, GRange* r, IEdges* ie) {
	const uint32_t s0 = top / H2G_GSIDE_SYMS, c0 = top - s0 * H2G_GSIDE_SYMS;
	Side128 sd = load_side128(g.sides + (size_t)s0 * 128);
	uint32_t t = rank_in_side128(g, sd, s0, c0, c), b;
	const uint32_t spread = bot - top;
	if(c0 + spread < H2G_GSIDE_SYMS) b = rank_in_side128(g, sd, s0, c0 + spread, c);   // initFromTopBot gfm.h:347
	else b = rank128(g, bot, c);
	if(ie) ie->n = 0;
	r->top = r->bot = r->node_top = r->node_bot = 0;
	if(t + 1 >= g.gbwtLen || t >= b) return false;
	const uint32_t node_top = rank_M(g, t + 1) - 1;
	uint32_t F_loc, M_occ;
	const uint32_t ft = node_to_Frow(g, t + 1, node_top, &F_loc, &M_occ);
	const uint32_t node_bot = rank_M(g, b);
	// :3812-3827 — the bottom takes the header of bot's own side, no backward scan
	const uint32_t* q = reinterpret_cast<const uint32_t*>(g.sides + (size_t)(b / H2G_GSIDE_SYMS) * 128 + 104);
	uint32_t bF = q[0];
	const uint32_t bM = q[1];
	if(bM > 0) bF++;
	const uint32_t fb = (node_bot + 1 > bM) ? select_F(g, bF, node_bot + 1 - bM) : bF;
	r->top = ft; r->bot = fb; r->node_top = node_top; r->node_bot = node_bot;
	if(ie && node_bot - node_top <= k && node_bot - node_top < fb - ft) in_edge_count(g, ft, fb, ie);
	return true;
}

// mapGLF1 (gfm.h:3957-4021) with mapLF1 (:3892): one row; false = cannot proceed on c
H2G_HD bool map_glf1(const DGfm& g, uint32_t row, int c, GRange* r) {
	r->top = r->bot = r->node_top = r->node_bot = 0;
	const uint32_t s0 = row / H2G_GSIDE_SYMS, c0 = row - s0 * H2G_GSIDE_SYMS;
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
H2G_HD void partial_search_graph_item(const DGfm& g, const SeqView& seq, uint32_t cur_in, bool pseudogeneStopIn,
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
				if(bot - top > 1) {
					h.nrank += 2;
					map_glf(g, top, bot, c, kseeds, &r, &tmp_ie);
				} else {
					h.nrank += 1;
					if(map_glf1(g, top, c, &r) && r.top + 1 < r.bot) {   // :6476-6482
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

}  // namespace h2g
