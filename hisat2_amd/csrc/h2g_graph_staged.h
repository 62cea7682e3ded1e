// h2g_graph_staged.h — glf1_top_fused (h2g_graph.h: one LF step of ONE row of a graph index, mapGLF1 without a required character,
// gfm.h:4029-4095) cut into the stages between its dependent loads, so that a lane can keep several rows' steps in flight: every stage of
// every row is issued before the next stage of any of them is consumed.  A step of one row is a chain of two or three lines — the row's side
// (rank), the side of t + 1 when it is another one (its M bits, F_loc, M_occ), the side select_F lands in when that is another one again.
// Measurement code (k_glf_chain in h2g_kernels.hip, tools/chain_bench.py) and the groundwork of walks with several rows per lane; the shipped
// kernels do not include this file.  Held to glf1_top_fused by the host instantiation (h2gemu_glf_staged_check).
#pragma once
#include "h2g_graph.h"

namespace h2g {

struct GlfStage {
	uint32_t s0, c0;            // the row's side and offset                      (after stage A)
	uint32_t s1, o1;            // side and offset of t + 1                       (after stage B)
	uint32_t sideNum;           // the side `cur` holds
	uint32_t node, count, fs, off, ft;   // the node; the F ones still to skip, from offset `off` of side fs; the result   (after stage C)
	uint32_t pending;           // stage D still has a select to run
	Side128 cur;
};

// A: request the row's side
template <class X> H2G_HD void glf_stage_a(const X& g, uint32_t row, GlfStage& st) {
	st.s0 = row / X::SYMS; st.c0 = row - st.s0 * X::SYMS;
	st.cur = load_side128(g.sides + (size_t)st.s0 * 128);
}
// B: rank of the row's own character; request the side of t + 1 when it is not the one in hand
template <class X> H2G_HD void glf_stage_b(const X& g, GlfStage& st) {
	const int c = rowL_in_side128(st.cur, st.c0);
	const uint32_t t = rank_in_side128(g, st.cur, st.s0, st.c0, c);
	const uint32_t r1 = t + 1;
	st.s1 = r1 / X::SYMS; st.o1 = r1 - st.s1 * X::SYMS;
	if(st.s1 != st.s0) st.cur = load_side128(g.sides + (size_t)st.s1 * 128);
}
// C: rank_M(t + 1) - 1, the backward scan over the side headers (rarely more than the side in hand), and the request for the side the select starts in
template <class X> H2G_HD void glf_stage_c(const X& g, GlfStage& st) {
	{
		const Bits256 m = bits_of_side<X>(st.cur, X::M_OFF);
		uint32_t cnt = side_hdr_reg<X>(st.cur, 1);
#pragma unroll
		for(int k = 0; k < 4; k++) cnt += (uint32_t)__builtin_popcountll(m.w[k] & low_mask((int)st.o1 - 64 * k));
		st.node = cnt - 1;
	}
	uint32_t sideNum = st.s1, F_loc = side_hdr_reg<X>(st.cur, 0), M_occ = side_hdr_reg<X>(st.cur, 1);
	while(!(M_occ <= st.node || sideNum == 0)) {
		sideNum--;
		st.cur = load_side128(g.sides + (size_t)sideNum * 128);
		F_loc = side_hdr_reg<X>(st.cur, 0); M_occ = side_hdr_reg<X>(st.cur, 1);
	}
	if(M_occ > 0) F_loc++;
	st.sideNum = sideNum;
	st.ft = F_loc;
	st.pending = 0;
	if(st.node + 1 > M_occ) {
		st.count = st.node + 1 - M_occ;
		st.fs = F_loc / X::SYMS; st.off = F_loc - st.fs * X::SYMS;
		st.ft = g.gbwtLen;
		st.pending = 1;
		if(st.fs != st.sideNum) { st.cur = load_side128(g.sides + (size_t)st.fs * 128); st.sideNum = st.fs; }
	}
}
// D: select_F from the side in hand (further sides, when the ones run out in this one, are fetched here)
template <class X> H2G_HD void glf_stage_d(const X& g, GlfStage& st, uint32_t* top_out, uint32_t* node_out) {
	if(st.pending) {
		const uint32_t lastSide = (g.gbwtLen - 1) / X::SYMS;
		uint32_t count = st.count, fs = st.fs, off = st.off;
		while(true) {
			if(fs != st.sideNum) { st.cur = load_side128(g.sides + (size_t)fs * 128); st.sideNum = fs; }
			const Bits256 f = bits_of_side<X>(st.cur, X::F_OFF);
			bool hit = false;
#pragma unroll
			for(int k = 0; k < 4; k++) {
				if(hit) continue;
				const uint64_t w = f.w[k] & ~low_mask((int)off - 64 * k);
				const uint32_t pc = (uint32_t)__builtin_popcountll(w);
				if(count <= pc) { st.ft = fs * X::SYMS + 64u * k + select_in_word(w, count); hit = true; }
				else count -= pc;
			}
			if(hit || fs >= lastSide) break;
			fs++;
			off = 0;
		}
	}
	*top_out = st.ft; *node_out = st.node;
}

// ---- map_glf1_fused (the searches' step once a range is one row: mapGLF1 WITH a required character, gfm.h:3957-4021) in the same stages.
// A is glf_stage_a; B fails the step when the row's character is another one (no further line is requested); C and D produce the GRange.
struct GlfSearchStage { GlfStage s; uint32_t c1, c2, F_loc, fail; };
template <class X> H2G_HD void glf_search_stage_b(const X& g, uint32_t row, int c, GlfSearchStage& q) {
	q.fail = (rowL_in_side128(q.s.cur, q.s.c0) != c || is_zoff(g, row)) ? 1u : 0u;
	if(q.fail) return;
	const uint32_t t = rank_in_side128(g, q.s.cur, q.s.s0, q.s.c0, c);
	const uint32_t r1 = t + 1;
	q.s.s1 = r1 / X::SYMS; q.s.o1 = r1 - q.s.s1 * X::SYMS;
	if(q.s.s1 != q.s.s0) q.s.cur = load_side128(g.sides + (size_t)q.s.s1 * 128);
}
template <class X> H2G_HD void glf_search_stage_c(const X& g, GlfSearchStage& q) {
	if(q.fail) return;
	GlfStage& st = q.s;
	{
		const Bits256 m = bits_of_side<X>(st.cur, X::M_OFF);
		uint32_t cnt = side_hdr_reg<X>(st.cur, 1);
#pragma unroll
		for(int k = 0; k < 4; k++) cnt += (uint32_t)__builtin_popcountll(m.w[k] & low_mask((int)st.o1 - 64 * k));
		st.node = cnt - 1;
	}
	uint32_t sideNum = st.s1, F_loc = side_hdr_reg<X>(st.cur, 0), M_occ = side_hdr_reg<X>(st.cur, 1);
	while(!(M_occ <= st.node || sideNum == 0)) {
		sideNum--;
		st.cur = load_side128(g.sides + (size_t)sideNum * 128);
		F_loc = side_hdr_reg<X>(st.cur, 0); M_occ = side_hdr_reg<X>(st.cur, 1);
	}
	if(M_occ > 0) F_loc++;
	st.sideNum = sideNum;
	q.F_loc = F_loc;
	q.c1 = st.node + 1 > M_occ ? st.node + 1 - M_occ : 0u;
	q.c2 = st.node + 2 > M_occ ? st.node + 2 - M_occ : 0u;
	const uint32_t fs = F_loc / X::SYMS;
	if((q.c1 || q.c2) && fs != st.sideNum) { st.cur = load_side128(g.sides + (size_t)fs * 128); st.sideNum = fs; }
}
template <class X> H2G_HD bool glf_search_stage_d(const X& g, GlfSearchStage& q, GRange* r) {
	r->top = r->bot = r->node_top = r->node_bot = 0;
	if(q.fail) return false;
	uint32_t ft, fb;
	select_F2_reg(g, q.s.cur, q.s.sideNum, q.F_loc, q.c1, q.c2, &ft, &fb);
	r->top = ft; r->bot = fb; r->node_top = q.s.node; r->node_bot = q.s.node + 1;
	return true;
}

// C single-row walks (gw_walk_single: a coordinate walk of one node and one row, group_walk.h:1430-1545 reduced to that case) advanced TOGETHER, stage by
// stage, by at most `budget` LF steps each: walk k stops for good once its offset is found (done bit k; off[k] = tryOffset + its step count).  Returns the done
// mask.  A walk that is done, or was done on entry (bit set in `done`), costs nothing further.
template <int C, class X>
H2G_HD uint32_t gw_walk_multi(const X& g, uint32_t* row, uint32_t* node, uint32_t* steps, uint32_t budget, uint32_t* off, uint32_t done) {
	while(true) {
#pragma unroll
		for(int k = 0; k < C; k++) {
			if(done & (1u << k)) continue;
			const uint32_t toff = gw_try_offset(g, row[k], node[k]);
			if(toff != H2G_MAX) { off[k] = toff + steps[k]; done |= 1u << k; }
		}
		if(done == (1u << C) - 1u || budget == 0) return done;
		budget--;
		GlfStage st[C];
#pragma unroll
		for(int k = 0; k < C; k++) if(!(done & (1u << k))) glf_stage_a(g, row[k], st[k]);
#pragma unroll
		for(int k = 0; k < C; k++) if(!(done & (1u << k))) glf_stage_b(g, st[k]);
#pragma unroll
		for(int k = 0; k < C; k++) if(!(done & (1u << k))) glf_stage_c(g, st[k]);
#pragma unroll
		for(int k = 0; k < C; k++) if(!(done & (1u << k))) { glf_stage_d(g, st[k], &row[k], &node[k]); steps[k]++; }
	}
}

}  // namespace h2g
