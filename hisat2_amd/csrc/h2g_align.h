// h2g_align.h — device-resident HI_Aligner::go for one read (linear index, unpaired), as an explicit-stack
// state machine: no recursion, fixed-capacity per-read workspace in HBM, an `overflow` flag instead of
// unbounded lists.  `__host__ __device__` like h2g_core.h so tests/emul can run the identical source on the CPU.
//
// Reference (HISAT2 2.2.3) call tree reproduced here, with the lines each piece follows:
//   HI_Aligner::go                hi_aligner.h:4048      nextBWT :4644   pickNextReadToSearch :4868
//   HI_Aligner::align             hi_aligner.h:5484      getAnchorHits :5007
//   SplicedAligner::hybridSearch  spliced_aligner.h:112  hybridSearch_recur :331 (left :813-1360, right :1496-2050)
//   localGFMSearch :6751  globalGFMSearch :6606  getGenomeCoords_local :5861
//   GenomeHit::compatibleWith :1375  combineWith :1420 (indel scan :1741-1794)  leftAlign :3554
//   reportHit :6064  redundant :6311  isSearched/addSearched :6898  AlnSinkWrap::report aln_sink.h:2565
//   selectByScore aln_sink.h:2680   RandomSource random_source.h:33
#pragma once
#include <math.h>
#include "h2g_core.h"
#include "h2g_sw.h"
#include "h2g_graph.h"
#if defined(H2G_TRACE) && !defined(__HIP_DEVICE_COMPILE__)
#include <stdio.h>
#define AL_TRACE(...) fprintf(stderr, __VA_ARGS__)
#else
#define AL_TRACE(...) do {} while(0)
#endif

namespace h2g {

#ifndef AL_MAX_GHITS           // the linear go() kernels are compiled with -DAL_MAX_GHITS=10 (smaller per-lane workspace)
#define AL_MAX_GHITS    20    // max(khits, kseeds): 10 on linear, 20 on graph indexes (hisat2.cpp:3174-3176, 3903-3906)
#endif
#define AL_MAX_SEARCHED 64
#define AL_MAX_RESULTS  32
#define AL_MAX_DEPTH    32
#define AL_MAX_LOCALHITS 4
#define AL_MAX_COORDS   12

// ---------------------------------------------------------------------------------------- local indexes (a13)
// LocalGFM (hgfm.h:35): 16-bit words.  Linear local side = 64 B = 56 B payload (224 symbols) + u16 occ[4].
struct DLocalDesc {
	uint64_t sides_off;      // byte offset into DLocalSet::sides
	uint32_t ftab_off, eftab_off, offs_off, rstarts_off;   // u16-word offsets into DLocalSet::words
	uint32_t len, gbwtLen, eftabLen, nFrag, nZ, zoff;
	uint32_t tidx, localOffset, joinedOffset;
	uint32_t fchr[5];
	uint32_t ftabLim;        // ftab entries above this point into eftab: len (linear) or gbwtLen (graph), gfm.h:2618
	uint32_t zoffs_off;      // index into DLocalSet::zoffs of this index's nZ '$' rows (a graph local index can have several)
};
struct DLocalSet {
	const DLocalDesc* desc;
	const uint8_t*    sides;
	const uint16_t*   words;
	const uint32_t*   first;   // [nPat+1] first local index of each text (HGFM::_localGFMs[tidx])
	const uint32_t*   zoffs;   // all '$' rows, DLocalDesc::zoffs_off
	uint32_t n, ftabChars, offRate;
};
#define H2G_LOCAL_INTERVAL 56320u   // local_index_interval hier_idx_common.h:24-31

// Policy of a GRAPH local index for the templates of h2g_graph.h: 128 B sides of u16 words — 232 symbols in 58 B,
// F bits at 58, M bits at 87, u16 {F_loc, M_occ, occ[4]} at 116 (GFMParams::init gfm.h:156-179 with index_t = uint16_t)
struct LGfm {
	const uint8_t*  sides;
	const uint16_t* offs;
	const uint32_t* zoffs;
	uint32_t nZ, zoff, gbwtLen, offMask, offRate;
	uint32_t fchr[5];
	static constexpr uint32_t SYMS = 232, NCW = 8, F_OFF = 58, M_OFF = 87, HDR = 116, WSZ = 2;
	H2G_HD uint32_t offs_at(uint32_t i) const { const uint32_t v = offs[i]; return v == 0xffffu ? H2G_MAX : v; }
};
H2G_HD LGfm lgfm_of(const DLocalSet& ls, const DLocalDesc& d) {
	LGfm x;
	x.sides = ls.sides + d.sides_off; x.offs = ls.words + d.offs_off; x.zoffs = ls.zoffs + d.zoffs_off;
	x.nZ = d.nZ; x.zoff = d.zoff; x.gbwtLen = d.gbwtLen; x.offRate = ls.offRate; x.offMask = (0xffffu << ls.offRate) & 0xffffu;
	for(int i = 0; i < 5; i++) x.fchr[i] = d.fchr[i];
	return x;
}

// Uniform view of "an FM index" for the search loops: global (u32, 192 symbols/side) or local (u16, 224/side)
struct GIdx {
	const DGfm* g;
	H2G_HD uint32_t ftabChars() const { return g->ftabChars; }
	H2G_HD bool is_zoff(uint32_t row) const { return g->nZ && row == g->zoff; }
	H2G_HD void lohi(uint32_t fi, uint32_t* top, uint32_t* bot) const { *top = ftab_hi(*g, fi); *bot = ftab_lo(*g, fi + 1); }
	H2G_HD uint32_t rank(uint32_t row, int c) const { return rank64(*g, row, c); }
	H2G_HD uint32_t side_of(uint32_t row) const { return row / 192u; }
	H2G_HD int rowL(uint32_t row) const {
		uint32_t s0 = row / 192u;
		return rowL_in_side64(load_side64(g->sides + (size_t)s0 * 64), row - s0 * 192u);
	}
};
struct LIdx {
	const DLocalSet* ls;
	const DLocalDesc* d;
	H2G_HD uint32_t ftabChars() const { return ls->ftabChars; }
	H2G_HD bool is_zoff(uint32_t row) const { return d->nZ && row == d->zoff; }
	H2G_HD uint32_t side_of(uint32_t row) const { return row / 224u; }
	H2G_HD uint32_t fh(uint32_t i) const {   // ftabHi gfm.h:2618 with 16-bit words
		uint32_t v = ls->words[d->ftab_off + i];
		if(v <= d->ftabLim) return v;
		return ls->words[d->eftab_off + ((v ^ 0xffffu) * 2 + 1)];
	}
	H2G_HD uint32_t fl(uint32_t i) const {
		uint32_t v = ls->words[d->ftab_off + i];
		if(v <= d->ftabLim) return v;
		return ls->words[d->eftab_off + ((v ^ 0xffffu) * 2)];
	}
	H2G_HD void lohi(uint32_t fi, uint32_t* top, uint32_t* bot) const { *top = fh(fi); *bot = fl(fi + 1); }
	H2G_HD uint32_t rank(uint32_t row, int c) const {   // countBt2Side gfm.h:2958 for index_t = uint16_t
		uint32_t sideNum = row / 224u, charOff = row - sideNum * 224u;
		Side64 s = load_side64(ls->sides + d->sides_off + (size_t)sideNum * 64);
		uint32_t cnt = 0;
#pragma unroll
		for(int k = 0; k < 7; k++) cnt += count_word(s.w[k], c, (int)charOff - 32 * k);
		if(c == 0 && d->nZ) {
			uint32_t zs = d->zoff / 224u, zc = d->zoff - zs * 224u;
			if(zs == sideNum && zc < charOff) cnt--;
		}
		uint32_t occ = (uint32_t)((s.w[7] >> (16 * c)) & 0xffffu);
		return occ + cnt + d->fchr[c];
	}
	H2G_HD int rowL(uint32_t row) const {
		uint32_t sideNum = row / 224u, charOff = row - sideNum * 224u;
		const uint8_t* p = ls->sides + d->sides_off + (size_t)sideNum * 64;
		return (p[charOff >> 2] >> ((charOff & 3) * 2)) & 3;
	}
};

// globalGFMSearch hi_aligner.h:6606-6744 / localGFMSearch :6751-6892 on a linear index.
// Returns nelt; hitlen/top/bot as the reference leaves them (hitlen untouched when nothing is reported).
template <typename IDX>
H2G_HD uint32_t gfm_search(const IDX& ix, const SeqView& seq, uint32_t rdoff, uint32_t* hitlen, uint32_t* top_o,
                           uint32_t* bot_o, bool* uniqueStop, uint32_t minUniqueLen, uint32_t maxHitLen, uint32_t maxHits,
                           bool local, uint32_t* nrank /* [0] rank calls, [1] unique sides */)
{
	const bool uniqueStop_ = *uniqueStop;
	*uniqueStop = false;
	const uint32_t ftabLen = ix.ftabChars(), len = seq.len;
	const uint32_t offset = len - rdoff - 1;
	uint32_t dep = offset;
	if(local) { *top_o = 0; *bot_o = 0; }
	const uint32_t left = len - dep;
	if(left < ftabLen + 1) { *hitlen = left; return 0; }
	uint32_t fi = 0;
	for(uint32_t i = 0; i < ftabLen; i++) {
		int c = seq.at(len - dep - 1 - i);
		if(c > 3) { *hitlen = i + 1; return 0; }
		fi |= (uint32_t)c << (2 * i);
	}
	uint32_t top, bot;
	ix.lohi(fi, &top, &bot);
	dep += ftabLen;
	if(top >= bot) { *hitlen = ftabLen; return 0; }
	uint32_t ntop = 0, nbot = 0;
	while(dep < len) {
		int c = seq.at(len - dep - 1);
		uint32_t ttop = 0, tbot = 0;
		if(c <= 3) {
			if(bot - top > 1) { nrank[0] += 2; nrank[1] += ix.side_of(top) == ix.side_of(bot) ? 1 : 2; ttop = ix.rank(top, c); tbot = ix.rank(bot, c); }
			else {
				nrank[0] += 1; nrank[1] += 1;
				if(ix.rowL(top) == c && !ix.is_zoff(top)) { ttop = ix.rank(top, c); tbot = ttop + 1; }
			}
		}
		if(ttop >= tbot) break;
		top = ttop; bot = tbot; ntop = ttop; nbot = tbot;
		dep++;
		if(uniqueStop_ && bot - top == 1 && dep - offset >= minUniqueLen) { *uniqueStop = true; break; }
		if(local && dep - offset >= maxHitLen) break;
	}
	if(ntop < nbot && nbot - ntop <= maxHits) {
		*top_o = top; *bot_o = bot; *hitlen = dep - offset;
		return nbot - ntop;
	}
	return 0;
}

template <typename IDX>
H2G_HD uint32_t sa_walk_idx(const IDX& ix, uint32_t row, uint32_t offMask, uint32_t offRate, const void* offs, bool offs16,
                            uint32_t* steps)
{
	uint32_t jumps = 0;
	while(true) {
		if(ix.is_zoff(row)) break;
		if((row & offMask) == row) {
			uint32_t off = offs16 ? ((const uint16_t*)offs)[row >> offRate] : ((const uint32_t*)offs)[row >> offRate];
			if(off != (offs16 ? 0xffffu : H2G_MAX)) { *steps += jumps; return off + jumps; }
		}
		int c = ix.rowL(row);
		row = ix.rank(row, c);
		jumps++;
	}
	*steps += jumps;
	return jumps;
}

// LocalGFM::joinedToTextOff (gfm.h:5527, 16-bit rstarts, rejectStraddle = true) + the local -> global shift of
// getGenomeCoords_local (hi_aligner.h:5925-5934).  false = skip this element.
H2G_HD bool local_joff_to_coord(const DLocalSet& ls, const DLocalDesc* d, uint32_t joff, uint32_t rdoff, uint32_t rdlen, h2g_coord* out) {
	const uint16_t* rs = ls.words + d->rstarts_off;
	uint32_t lo = 0, hi = d->nFrag, elt = H2G_MAX, toff = 0;
	bool ok = false;
	while(true) {
		uint32_t oldelt = elt;
		elt = lo + ((hi - lo) >> 1);
		if(oldelt == elt) break;
		uint32_t lower = rs[elt * 3], upper = (elt == d->nFrag - 1) ? d->len : rs[(elt + 1) * 3];
		AL_TRACE("       frag %u/%u lower %u upper %u joff %u rdlen %u fragoff %u\n", elt, d->nFrag, lower, upper, joff, rdlen, rs[elt * 3 + 2]);
		if(lower <= joff) {
			if(upper > joff) {
				if(joff + rdlen > upper) break;          // straddles: rejected => result false
				toff = (joff - lower) + rs[elt * 3 + 2];
				ok = true;
				break;
			}
			lo = elt;
		} else hi = elt;
	}
	if(!ok) return false;                                    // `if(!result) continue;`
	const uint32_t global_toff = toff + d->localOffset;
	if(global_toff < rdoff) return false;
	out->tidx = d->tidx; out->toff = global_toff; out->joinedOff = joff + d->joinedOffset;
	return true;
}

// getGenomeCoords_local hi_aligner.h:5861-5941 on a linear local index
H2G_HD bool genome_coords_local(const LIdx& ix, uint32_t top, uint32_t bot, uint32_t rdoff, uint32_t rdlen, h2g_coord* coords,
                                uint32_t cap, uint32_t* ncoords, uint32_t* nsteps)
{
	const DLocalDesc* d = ix.d;
	const uint32_t offMask = (0xffffu << ix.ls->offRate) & 0xffffu;
	uint32_t n = 0;
	for(uint32_t e = 0; e < bot - top; e++) {
		uint32_t joff = sa_walk_idx(ix, top + e, offMask, ix.ls->offRate, ix.ls->words + d->offs_off, true, nsteps);
		h2g_coord c;
		if(!local_joff_to_coord(*ix.ls, d, joff, rdoff, rdlen, &c)) continue;
		if(n < cap) coords[n++] = c;
	}
	*ncoords = n;
	return true;
}

// ---------------------------------------------------------------------------------------- PRNG (a27)
struct Rng {   // RandomSource random_source.h:33-60
	uint32_t last;
	H2G_HD void init(uint32_t seed) { last = seed; }
	H2G_HD uint32_t nextU32() {
		last = 1664525u * last + 1013904223u;
		uint32_t ret = last >> 16;
		last = 1664525u * last + 1013904223u;
		return ret ^ last;
	}
};

// genRandSeed pat.h:55-91 with global seed 0 (name bytes come from the host)
H2G_HD uint32_t gen_rand_seed(const SeqView& fwseq, const char* name, uint32_t namelen, uint32_t seed) {
	uint32_t rseed = (seed + 101u) * 59u * 61u * 67u * 71u * 73u * 79u * 83u;
	for(uint32_t i = 0; i < fwseq.len; i++) rseed ^= ((uint32_t)fwseq.fwc[i] << ((i & 15) << 1));
	for(uint32_t i = 0; i < fwseq.len; i++) rseed ^= ((uint32_t)(fwseq.q ? fwseq.q[i] : 'I') << ((i & 3) << 3));
	for(uint32_t i = 0; i < namelen; i++) {
		int p = name[i];
		if(p == '/') break;
		rseed ^= ((uint32_t)p << ((i & 3) << 3));
	}
	return rseed;
}

// ---------------------------------------------------------------------------------------- GenomeHit helpers
H2G_HD void hit_init(h2g_ghit* h, bool fw, uint32_t rdoff, uint32_t len, uint32_t tidx, uint32_t toff, uint32_t joff) {
	h->read = 1;   // _hitcount
	h->fw = fw; h->rdoff = rdoff; h->len = len; h->trim5 = 0; h->trim3 = 0; h->tidx = tidx; h->toff = toff; h->joinedOff = joff;
	h->score = 0; h->nedits = 0; h->overflow = 0;
}
H2G_HD void hit_copy(h2g_ghit* d, const h2g_ghit* s) {
	d->read = s->read; d->fw = s->fw; d->rdoff = s->rdoff; d->len = s->len; d->trim5 = s->trim5; d->trim3 = s->trim3;
	d->tidx = s->tidx; d->toff = s->toff; d->joinedOff = s->joinedOff; d->score = s->score; d->nedits = s->nedits;
	d->overflow = s->overflow;
	for(uint32_t i = 0; i < s->nedits; i++) d->edits[i] = s->edits[i];
}
H2G_HD int base_code(uint8_t ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4; }

// Scoring::score(rdc, refm = 1 << rfc, q) scoring.h:259-269
H2G_HD int score_cell(const DScoring& sc, int rdc, int rfc, int q) {
	if(rdc > 3 || rfc > 3) return -sc.nPen;
	if(rdc == rfc) return sc.matchBonus;
	return -mm_penalty(sc, q);
}

// getLeft hi_aligner.h:919-958 (optionally with score)
H2G_HD void hit_get_left(const h2g_ghit* h, const DScoring* sc, const SeqView* seq, uint32_t* rdoff, uint32_t* len, uint32_t* toff,
                         int64_t* score)
{
	*toff = h->toff; *rdoff = h->rdoff; *len = h->len;
	if(score) *score = 0;
	for(uint32_t i = 0; i < h->nedits; i++) {
		const h2g_edit e = h->edits[i];
		if(is_stop_edit(e)) { *len = e.pos; break; }
		if(score && e.type == H2G_EDIT_MM)
			*score += score_cell(*sc, base_code(e.qchr), base_code(e.chr), seq->qual(h->rdoff + e.pos) - 33);
	}
}
// getRight hi_aligner.h:962-1014 with score
H2G_HD void hit_get_right_sc(const h2g_ghit* h, const DScoring* sc, const SeqView* seq, uint32_t* rdoff, uint32_t* len,
                             uint32_t* toff, int64_t* score)
{
	hit_get_right(h, rdoff, len, toff);
	if(!score) return;
	*score = 0;
	for(int i = (int)h->nedits - 1; i >= 0; i--) {
		const h2g_edit e = h->edits[i];
		if(is_stop_edit(e)) break;
		if(e.type == H2G_EDIT_MM)
			*score += score_cell(*sc, base_code(e.qchr), base_code(e.chr), seq->qual(h->rdoff + e.pos) - 33);
	}
}

// compatibleWith hi_aligner.h:1375-1413
H2G_HD bool hit_compatible(const h2g_ghit* a, const h2g_ghit* b, uint32_t maxIntronLen, bool no_spliced) {
	if(a == b) return false;
	if(a->fw != b->fw || a->tidx != b->tidx) return false;
	if(a->rdoff > b->rdoff) return false;
	if(a->rdoff + a->len > b->rdoff + b->len) return false;
	if(a->toff > b->toff) return false;
	uint32_t ar, al, at, br, bl, bt;
	hit_get_right(a, &ar, &al, &at);
	hit_get_left(b, nullptr, nullptr, &br, &bl, &bt, nullptr);
	if(ar > br) return false;
	if(ar + al > br + bl) return false;
	if(at > bt) return false;
	if(!no_spliced) { if(bt - at > (br - ar) + maxIntronLen) return false; }
	return true;
}

// Scoring::maxReadGaps / maxRefGaps scoring.cpp:42-98 (match bonus 0)
H2G_HD int max_gaps(int64_t minsc, int open, int ext) {
	int64_t s = 0;
	bool first = true;
	int num = 0;
	while(s >= minsc) { s -= first ? open : ext; first = false; num++; }
	return num - 1;
}

// leftAlign hi_aligner.h:3554-3610
H2G_HD void hit_left_align(h2g_ghit* h, const SeqView& seq) {
	for(uint32_t ei = 0; ei < h->nedits; ei++) {
		h2g_edit& edit = h->edits[ei];
		if(!is_gap(edit.type)) continue;
		if(edit.snp != H2G_MAX) continue;                        // known indels stay where the ALT puts them (:3562)
		uint32_t ei2 = ei + 1;
		for(; ei2 < h->nedits; ei2++) {
			const h2g_edit& e2 = h->edits[ei2];
			if(e2.type != edit.type) break;
			if(edit.type == H2G_EDIT_READ_GAP) { if(edit.pos != e2.pos) break; }
			else if(edit.pos + ei2 - ei != e2.pos) break;
		}
		ei2 -= 1;
		int b = 0;
		if(ei > 0) b = (int)h->edits[ei - 1].pos;
		int l = (int)edit.pos - 1;
		while(l > b) {
			int rdc = seq.at(h->rdoff + l);
			uint8_t rfc = (edit.type == H2G_EDIT_READ_GAP ? h->edits[ei2].chr : h->edits[ei2].qchr);
			if(rfc != base_char(rdc)) break;
			for(int ei3 = (int)ei2; ei3 > (int)ei; ei3--) {
				if(edit.type == H2G_EDIT_READ_GAP) h->edits[ei3].chr = h->edits[ei3 - 1].chr;
				else h->edits[ei3].qchr = h->edits[ei3 - 1].qchr;
				h->edits[ei3].pos -= 1;
			}
			if(edit.type == H2G_EDIT_READ_GAP) edit.chr = base_char(rdc); else edit.qchr = base_char(rdc);
			edit.pos -= 1;
			l--;
		}
		ei = ei2;
	}
}

H2G_HD void hit_push_edit(h2g_ghit* h, uint32_t pos, uint8_t chr, uint8_t qchr, uint8_t type) {
	if(h->nedits >= H2G_MAX_EDITS) { h->overflow = 1; return; }
	h2g_edit& e = h->edits[h->nedits++];
	e.pos = pos; e.chr = chr; e.qchr = qchr; e.type = type; e.pad = 0; e.snp = H2G_MAX;
}

// combineWith hi_aligner.h:1420-2025 for linear indexes without spliced alignment: plain concatenation
// (:1506-1525) or one insertion / deletion placed by the prefix/suffix score scan (:1741-1794).
H2G_HD bool hit_combine(const DRef& ref, const DScoring& sc, const SeqView& seq, h2g_ghit* a, const h2g_ghit* b, int64_t minsc,
                        uint32_t minIntronLen, bool no_spliced, int64_t* tmp1, int64_t* tmp2, const DAlts* alts = nullptr)
{
	if(a == b) return false;
	uint32_t this_rdoff, this_len, this_toff, other_rdoff, other_len, other_toff;
	int64_t this_score, other_score;
	hit_get_right_sc(a, &sc, &seq, &this_rdoff, &this_len, &this_toff, &this_score);
	hit_get_left(b, &sc, &seq, &other_rdoff, &other_len, &other_toff, &other_score);
	if(this_len != 0 && other_len != 0 && this_rdoff + this_len > other_rdoff + other_len) return false;
	const uint32_t len = other_rdoff - this_rdoff + other_len;
	const uint32_t reflen = ref.refLens[a->tidx];
	if(this_toff + len > reflen) return false;
	const uint32_t refdif = other_toff - this_toff, rddif = other_rdoff - this_rdoff;
	bool spliced = false, ins = false, del = false;
	if(refdif != rddif) {
		if(refdif > rddif) {
			if(!no_spliced && refdif - rddif >= minIntronLen) spliced = true; else del = true;
		} else ins = true;
	}
	if(spliced) return false;   // spliced alignment is not built (no_spliced_alignment mode only)
	if(!ins && !del && this_rdoff + this_len == other_rdoff) {
		const uint32_t addoff = b->rdoff - a->rdoff;
		for(uint32_t i = 0; i < b->nedits; i++) {
			hit_push_edit(a, b->edits[i].pos + addoff, b->edits[i].chr, b->edits[i].qchr, b->edits[i].type);
			if(!a->overflow) a->edits[a->nedits - 1].snp = b->edits[i].snp;
		}
		a->len += b->len;
		calculate_score(sc, seq, a);
		return true;
	}
	const uint32_t rdlen = seq.len;
	int64_t remainsc = minsc - (a->score - this_score) - (b->score - other_score);
	if(remainsc > 0) remainsc = 0;
	const int read_gaps = max_gaps(remainsc, sc.rdGapConst + sc.rdGapLinear, sc.rdGapLinear);
	const int ref_gaps = max_gaps(remainsc, sc.rfGapConst + sc.rfGapLinear, sc.rfGapLinear);
	(void)rdlen;
	if(ins) { if(refdif + ref_gaps < rddif) return false; }
	else if(del) { if(rddif + read_gaps < refdif) return false; }
	int this_ref_ext = read_gaps;
	if(this_toff + len > reflen) return false;
	if(this_toff + len + this_ref_ext > reflen) this_ref_ext = (int)(reflen - (this_toff + len));
	// refbuf[i]  = ref[this_toff + i]                              (i < len + this_ref_ext)
	// refbuf2[i] = ref[other_toff + other_len - len + i]           (window anchored at the right end)
	RefCursor rc1, rc2;
	rc1.init(&ref, a->tidx);
	rc2.init(&ref, b->tidx);
	const int64_t base2 = (int64_t)other_toff + other_len - len;
	uint32_t maxscorei = H2G_MAX;
	int64_t maxscore = INT64_MIN;
	if(ins || del) {
		if(len > 512) { a->overflow = 1; return false; }   // sc1 / sc2 capacity (reads up to 512 bp scan exactly; longer ones are flagged)
		const int inslen = ins ? (int)(rddif - refdif) : 0, dellen = del ? (int)(refdif - rddif) : 0;
		int64_t gap_penalty;
		if(ins) gap_penalty = -((int64_t)(sc.rfGapConst + sc.rfGapLinear) + (int64_t)sc.rfGapLinear * (inslen - 1));
		else    gap_penalty = -((int64_t)(sc.rdGapConst + sc.rdGapLinear) + (int64_t)sc.rdGapLinear * (dellen - 1));
		if(gap_penalty < remainsc) return false;
		int i;
		for(i = 0; i < (int)len; i++) {
			int rdc = seq.at(this_rdoff + i), rfc = rc1.get((int64_t)this_toff + i);
			tmp1[i] = i > 0 ? tmp1[i - 1] : 0;
			if(rdc != rfc) tmp1[i] += score_cell(sc, rdc, rfc, seq.qual(this_rdoff + i) - 33);
			if(tmp1[i] + gap_penalty < remainsc) break;
		}
		const int i_limit = i < (int)len ? i : (int)len;
		int i2;
		for(i2 = (int)len - 1; i2 >= 0; i2--) {
			int64_t p = base2 + i2;
			int rdc = seq.at(this_rdoff + i2), rfc = p < 0 ? 4 : rc2.get(p);
			tmp2[i2] = (uint32_t)(i2 + 1) < len ? tmp2[i2 + 1] : 0;
			if(rdc != rfc) tmp2[i2] += score_cell(sc, rdc, rfc, seq.qual(this_rdoff + i2) - 33);
			if(tmp2[i2] + gap_penalty < remainsc) break;
		}
		const int i2_limit = (i2 < inslen ? 0 : i2 - inslen);
		for(i = i2_limit, i2 = i2_limit + 1 + inslen; i < i_limit && i2 < (int)len; i++, i2++) {
			int64_t t = tmp1[i] + tmp2[i2] + gap_penalty;
			if(maxscore < t) { maxscore = t; maxscorei = (uint32_t)i; }
		}
		if(maxscore == INT64_MIN) return false;
		if(maxscore < remainsc) return false;
	}
	// keep this hit's edits up to (and including) its last gap; drop the MMs after it (:1818-1831)
	{
		bool clear = true;
		for(int i = (int)a->nedits - 1; i >= 0; i--) {
			if(is_stop_edit(a->edits[i])) { a->nedits = (uint32_t)i + 1; clear = false; break; }
		}
		if(clear) a->nedits = 0;
	}
	{
		uint32_t ins_len = 0;
		const uint32_t addoff = this_rdoff - a->rdoff;
		for(uint32_t i = 0; i < len; i++) {
			int rdc = seq.at(this_rdoff + i);
			int64_t p2 = base2 + i;
			int rfc = (i <= maxscorei) ? rc1.get((int64_t)this_toff + i) : (p2 < 0 ? 4 : rc2.get(p2));
			if(rdc != rfc) {
				hit_push_edit(a, i + addoff, base_char(rfc), base_char(rdc), H2G_EDIT_MM);
				if(alts && alts->n && !a->overflow) {                  // known SNP at this position? (:1913-1931)
					const uint32_t cpos = a->joinedOff + i + (this_toff - a->toff) - ins_len;
					for(uint32_t ai = alt_lobound(*alts, cpos); ai < alts->n; ai++) {
						const DAlt alt = alts->a[ai];
						if(alt.pos > cpos) break;
						if(alt.type != H2G_ALT_SNP_SGL) continue;
						if(alt.seq == (uint64_t)rdc) { a->edits[a->nedits - 1].snp = ai; break; }
					}
				}
			}
			if(i == maxscorei) {
				const uint32_t left = this_toff + i + 1;
				if(other_toff + other_len < len - i - 1) return false;
				const uint32_t right = other_toff + other_len - (len - i - 1);
				if(del) {
					const uint32_t skipLen = right - left;
					for(uint32_t j = 0; j < skipLen; j++) {
						int t = rc1.get((int64_t)this_toff + i + 1 + j);   // refbuf / getBase beyond it
						hit_push_edit(a, i + 1 + addoff, base_char(t), '-', H2G_EDIT_READ_GAP);
					}
				} else {
					const uint32_t skipLen = left - right;
					for(uint32_t j = 0; j < skipLen; j++) {
						int t = seq.at(this_rdoff + i + 1 + j);
						hit_push_edit(a, i + 1 + j + addoff, '-', base_char(t), H2G_EDIT_REF_GAP);
					}
					i += skipLen;
					ins_len += skipLen;
				}
			}
		}
	}
	{
		uint32_t fsi = b->nedits;
		for(uint32_t i = 0; i < b->nedits; i++) if(is_stop_edit(b->edits[i])) { fsi = i; break; }
		const uint32_t addoff = b->rdoff - a->rdoff;
		for(uint32_t i = fsi; i < b->nedits; i++) {
			hit_push_edit(a, b->edits[i].pos + addoff, b->edits[i].chr, b->edits[i].qchr, b->edits[i].type);
			if(!a->overflow) a->edits[a->nedits - 1].snp = b->edits[i].snp;
		}
	}
	if(ins || del) hit_left_align(a, seq);
	a->len = b->rdoff + b->len - a->rdoff;
	a->trim3 += b->trim3;
	calculate_score(sc, seq, a);
	return true;
}

// operator== hi_aligner.h:1156-1183
H2G_HD bool hit_equal(const h2g_ghit* a, const h2g_ghit* b) {
	if(a->fw != b->fw || a->rdoff != b->rdoff || a->len != b->len || a->tidx != b->tidx || a->toff != b->toff ||
	   a->trim5 != b->trim5 || a->trim3 != b->trim3) return false;
	if(a->nedits != b->nedits) return false;
	for(uint32_t i = 0; i < a->nedits; i++) {
		const h2g_edit e = a->edits[i], o = b->edits[i];
		if(e.type == H2G_EDIT_READ_GAP) { if(o.type != H2G_EDIT_READ_GAP) return false; }
		else if(e.type == H2G_EDIT_REF_GAP) { if(o.type != H2G_EDIT_REF_GAP) return false; }
		else if(!(e.type == o.type && e.pos == o.pos && e.chr == o.chr && e.qchr == o.qchr)) return false;
	}
	return true;
}

// ---------------------------------------------------------------------------------------- per-read workspace
struct AlnParams {
	uint32_t khits, kseeds, no_spliced, secondary;
	uint32_t minIntronLen, maxIntronLen, minAnchorLen, minAnchorLen_noncan, minK_local;
	uint32_t pseudogeneStop, anchorStop;
	uint32_t maxFragLen;     // PairedEndPolicy::maxFragLen = -X (hisat2.cpp:345)
	uint32_t bowtie2_dp;     // ReportingParams::bowtie2_dp: 0 off, 1 conditional, 2 unconditional (hisat2.cpp:529, 1770)
	uint32_t scoreMinType = 2;                                     // SimpleFunc scoreMin: 1 C, 2 L, 3 S, 4 G (simple_func.h:30-33)
	double   scoreMinConst = 0.0, scoreMinCoeff = (double)(-0.2f); // --score-min, default L,0,-0.2 (hisat2.cpp:440)
	DScoring sc;
};
// scoreMin.f<TAlScore>(len) (simple_func.h:88-110, hisat2.cpp:3380-3397: clamped to <= 0 in end-to-end mode)
H2G_HD int64_t min_score_for(const AlnParams& P, uint32_t len) {
	double X = 0.0;
	if(P.scoreMinType == 2) X = (double)len;
	else if(P.scoreMinType == 3) X = sqrt((double)len);
	else if(P.scoreMinType == 4) X = log((double)len);
	int64_t minsc = (int64_t)(P.scoreMinConst + P.scoreMinCoeff * X);
	return minsc > 0 ? 0 : minsc;
}
// the C-ABI parameter block -> AlnParams (everything that is not an option is the reference's constant)
inline AlnParams aln_params_from(const h2g_align_params& p, bool no_spliced, bool linear) {
	AlnParams P;
	P.khits = p.khits; P.kseeds = p.kseeds; P.no_spliced = no_spliced ? 1 : 0; P.secondary = p.secondary;
	P.minIntronLen = 20; P.maxIntronLen = 500000; P.minAnchorLen = 7; P.minAnchorLen_noncan = 14; P.minK_local = 8;   // tp.h, hi_aligner.h:3986
	P.pseudogeneStop = (linear && !no_spliced) ? 1 : 0; P.anchorStop = 1; P.maxFragLen = 1000;
	P.bowtie2_dp = p.bowtie2_dp;
	P.scoreMinType = p.score_min_type; P.scoreMinConst = p.score_min_const; P.scoreMinCoeff = p.score_min_coeff;
	P.sc.mmpMax = p.mm_max; P.sc.mmpMin = p.mm_min; P.sc.nPen = p.n_pen; P.sc.rdGapConst = p.rdg_const; P.sc.rdGapLinear = p.rdg_linear;
	P.sc.rfGapConst = p.rfg_const; P.sc.rfGapLinear = p.rfg_linear; P.sc.scMax = p.sc_max; P.sc.scMin = p.sc_min;
	return P;
}
inline void align_params_defaults(h2g_align_params* p, bool linear) {
	p->khits = linear ? 5 : 10;                       // hisat2.cpp:3903-3906
	p->kseeds = p->khits * 2 > 5 ? p->khits * 2 : 5;  // --max-seeds default hisat2.cpp:3174-3176
	p->no_spliced_alignment = 1; p->secondary = 0; p->bowtie2_dp = 0;
	p->mm_max = 6; p->mm_min = 2; p->n_pen = 1; p->rdg_const = 5; p->rdg_linear = 3; p->rfg_const = 5; p->rfg_linear = 3; p->sc_max = 2; p->sc_min = 1;
	p->score_min_type = 2; p->score_min_const = 0.0; p->score_min_coeff = (double)(-0.2f);
}

// One reported alignment = the arguments reportHit (hi_aligner.h:6064-6166) hands to AlnRes::init
struct AlnRec {
	uint32_t fw, tidx, toff, len, trim5, trim3, nedits, pad;
	int64_t  score;
	h2g_edit edits[H2G_MAX_EDITS];   // as stored in the AlnRes: 5'-to-3' positions of the ORIGINAL read relative to the first aligned base
};

struct Frame {
	h2g_ghit hit;
	uint32_t hitoff, hitlen;
	int64_t  maxsc, prev_score;
	uint32_t state;
	uint32_t count, lidx, extoff, extlen, ncoords, nlocal, ti;
	int32_t  ri;
	uint8_t  success, first, use_localindex, uniqueStop;
	h2g_coord coords[AL_MAX_COORDS];
	h2g_ghit  local_hits[AL_MAX_LOCALHITS];
};

#define AL_MAX_PARTIAL 24
struct PartialHit {          // BWTHit hi_aligner.h:108 (linear: node range == row range, no in-edge list)
	uint32_t top, bot, bwoff, len, hit_type, ncoords;
	h2g_coord coords[AL_MAX_GHITS];
};
struct RBHit {               // ReadBWTHit hi_aligner.h:216
	uint32_t len, cur, done, numPartialSearch, numUniqueSearch, npartial;
	PartialHit partial[AL_MAX_PARTIAL];
};

#define AL_MAX_PAIRS 32
struct MateWS {              // per-mate state of HI_Aligner + the per-mate half of AlnSinkWrap
	RBHit      rb[2];                            // _hits[rdi][fwi]
	h2g_ghit   searched[AL_MAX_SEARCHED];        // _hits_searched[rdi]
	uint32_t   nsearched;
	AlnRec     res[AL_MAX_RESULTS];              // AlnSinkWrap rs1u_ / rs2u_
	uint32_t   nres;
	int64_t    bestUnp, best2Unp;                // bestUnp1_/bestUnp2_, best2Unp*_
	int64_t    minsc;
	// initRead(rds[1], ..., rightendonly) (hi_aligner.h:3993, hisat2.cpp:3524): the lone mate 2 is searched as rdi 0 but
	// REPORTED into the sink's mate-2 list (:6192), so everything the aligner reads back by rdi — sink.bestUnp1() in
	// nextBWT / align / hybridSearch, redundant() :6311 — sees the empty mate-1 list
	uint32_t   sink_hidden;
};

struct AlignWS {
	MateWS     m[2];
	h2g_ghit   ghits[AL_MAX_GHITS];              // _genomeHits (hitcount lives in .read)
	uint32_t   nghits;
	uint8_t    ghit_done[AL_MAX_GHITS];
	// concordant pairs (AlnSinkWrap rs1_/rs2_ as indexes into m[0].res / m[1].res)
	uint8_t    pair_i[AL_MAX_PAIRS], pair_j[AL_MAX_PAIRS];
	uint32_t   npairs, insp_i, insp_j;           // _concordantIdxInspected
	int64_t    bestPair, best2Pair;
	uint64_t   localindexatts, max_localindexatts;
	uint32_t   overflow;
	uint32_t   nrank, nside, nsteps, nframes_max;   // nrank, nside adjacent: gfm_search updates both through &nrank
	h2g_ghit   tmp, tmp2;                        // scratch hits
	int64_t    sc1[512], sc2[512];               // combineWith temp_scores
	Frame      stack[AL_MAX_DEPTH];
};

// Edit::invertPoss edit.cpp:70-111 applied to the k-th element of the inverted list
H2G_HD h2g_edit inverted_edit(const h2g_ghit* h, uint32_t k, uint32_t sz, uint32_t add) {
	h2g_edit e = h->edits[h->nedits - 1 - k];
	uint32_t pos = e.pos + add;
	e.pos = (e.type == H2G_EDIT_READ_GAP) ? sz - pos : sz - pos - 1;
	return e;
}

// redundant hi_aligner.h:6311-6351
H2G_HD bool al_redundant(const MateWS* ws, const h2g_ghit* hit, uint32_t rdlen) {
	if(ws->sink_hidden) return false;
	for(uint32_t i = 0; i < ws->nres; i++) {
		const AlnRec& r = ws->res[i];
		if(r.tidx != hit->tidx || r.toff != hit->toff || r.fw != hit->fw) continue;
		if(r.nedits != hit->nedits) continue;
		uint32_t k = 0;
		for(; k < r.nedits; k++) {
			h2g_edit e = hit->fw ? hit->edits[k] : inverted_edit(hit, k, rdlen, 0);
			const h2g_edit o = r.edits[k];
			if(!(e.type == o.type && e.pos == o.pos && e.chr == o.chr && e.qchr == o.qchr)) break;
		}
		if(k >= r.nedits) return true;
	}
	return false;
}

H2G_HD bool al_is_searched(const MateWS* ws, const h2g_ghit* hit) {
	for(uint32_t i = 0; i < ws->nsearched; i++) if(hit_equal(&ws->searched[i], hit)) return true;
	return false;
}
H2G_HD void al_add_searched(AlignWS* aw, MateWS* ws, const h2g_ghit* hit) {
	if(ws->nsearched >= AL_MAX_SEARCHED) { aw->overflow |= 2; return; }
	hit_copy(&ws->searched[ws->nsearched++], hit);
}

// reportHit hi_aligner.h:6064-6166 + AlnSinkWrap::report aln_sink.h:2565-2650 (unpaired mate 1)
H2G_HD bool al_report(AlignWS* aw, MateWS* ws, const h2g_ghit* hit, uint32_t rdlen, int64_t minsc) {
	if(hit->rdoff - hit->trim5 > 0 || hit->len + hit->trim5 + hit->trim3 < rdlen) return false;
	if(hit->score < minsc) return false;
	if(ws->nres >= AL_MAX_RESULTS) { aw->overflow |= 4; return false; }
	AL_TRACE("  REPORT fw %u tidx %u toff %u len %u trim %u/%u score %lld nedits %u\n", hit->fw, hit->tidx, hit->toff, hit->len, hit->trim5, hit->trim3, (long long)hit->score, hit->nedits);
	AlnRec& r = ws->res[ws->nres++];
	r.fw = hit->fw; r.tidx = hit->tidx; r.toff = hit->toff; r.len = hit->len; r.trim5 = hit->trim5; r.trim3 = hit->trim3;
	r.nedits = hit->nedits; r.pad = 0; r.score = hit->score;
	// reportHit shifts by trim5 and inverts for !fw (hi_aligner.h:6093-6101); AlnRes::setShape then shifts the
	// stored copy by the 5' trim in read orientation (aligner_result.cpp:110-118)
	const uint32_t trim5p = hit->fw ? hit->trim5 : hit->trim3;
	for(uint32_t k = 0; k < hit->nedits; k++) {
		if(hit->fw) { r.edits[k] = hit->edits[k]; r.edits[k].pos += hit->trim5; }
		else r.edits[k] = inverted_edit(hit, k, rdlen, hit->trim5);
		r.edits[k].pos -= trim5p;
	}
	if(!ws->sink_hidden) {
		if(hit->score > ws->bestUnp) { ws->best2Unp = ws->bestUnp; ws->bestUnp = hit->score; }
		else if(hit->score > ws->best2Unp) ws->best2Unp = hit->score;
	}
	return true;
}

// ---------------------------------------------------------------------------------------- getAnchorHits (a16)
H2G_HD bool ph_empty(const PartialHit& p) { return p.bot <= p.top; }

struct AlnCtx {
	const DGfm* g;
	const DRef* ref;
	const DLocalSet* ls;
	const AlnParams* P;
	uint8_t* sw = nullptr;   // this lane's Smith-Waterman scratch (sw_scratch_bytes), only when P->bowtie2_dp != 0
	const DAlts* alts = nullptr;      // graph index: the ALT database
	struct GraphWS* gws = nullptr;    // graph index: this lane's graph scratch
	bool graph = false;               // set from a kernel template constant so that the linear kernels carry no graph code
};

// Per-lane scratch of the graph paths (allocated only for graph indexes, so the linear workspace keeps its size):
// group-walk state, ALT-extension state, and the node range + in-edge list of every partial hit (BWTHit::_node_top,
// _node_bot, _node_iedge_count hi_aligner.h:196-199), indexed [mate slot][strand][partial hit].
struct GraphPNode { uint32_t node_top, node_bot; IEdges ie; };
struct GraphWS {
	GwCtx      gw;
	AwaWS      awa;
	IEdges     ie;                 // in-edge list of the last search (waits for the coordinate call)
	uint32_t   node_top, node_bot; // its node range
	GraphPNode pnode[2][2][AL_MAX_PARTIAL];
};

// localGFMSearch / getGenomeCoords_local / globalGFMSearch / getGenomeCoords on whichever index this is.  On a graph
// index the node range and in-edge list of the last search wait in the lane's GraphWS for the coordinate call.
H2G_HD uint32_t al_local_search(const AlnCtx& C, AlignWS* ws, uint32_t lidx, const SeqView& seq, uint32_t extoff, uint32_t* extlen,
                                uint32_t* top, uint32_t* bot, bool* uniqueStop, uint32_t maxHitLen);
H2G_HD void al_local_coords(const AlnCtx& C, AlignWS* ws, uint32_t lidx, uint32_t top, uint32_t bot, uint32_t rdoff, uint32_t rdlen,
                            h2g_coord* coords, uint32_t cap, uint32_t* ncoords);
H2G_HD uint32_t al_global_search(const AlnCtx& C, AlignWS* ws, const SeqView& seq, uint32_t extoff, uint32_t* extlen, uint32_t* top,
                                 uint32_t* bot, bool* uniqueStop);
H2G_HD uint32_t al_global_coords(const AlnCtx& C, AlignWS* ws, uint32_t top, uint32_t bot, uint32_t extlen, h2g_coord* coords, uint32_t cap);

// tempHit.adjustWithALT(...) of hybridSearch_recur (spliced_aligner.h:946, 1139, 1635, 1826): always true on a linear index
H2G_HD bool al_adjust_member(const AlnCtx& C, const SeqView& seq, h2g_ghit* t, AlignWS* ws);

// GenomeHit::extend on whichever index this is
H2G_HD bool al_extend(const AlnCtx& C, const SeqView& seq, h2g_ghit* h, uint32_t mm, uint32_t ml, uint32_t mr, uint32_t* le, uint32_t* re) {
	if(!C.graph) return extend_item(*C.ref, C.P->sc, seq, h, mm, ml, mr, le, re);
	return extend_item_alts(*C.ref, *C.alts, C.P->sc, seq, h, mm, ml, mr, le, re, &C.gws->awa);
}

// hi_aligner.h:5007-5193 for one (read, strand)
H2G_HD uint32_t al_get_anchor_hits(const AlnCtx& C, const SeqView& seq, AlignWS* ws, MateWS* mw, int fwi, Rng* rnd) {
	const DGfm& g = *C.g;
	const AlnParams& P = *C.P;
	const bool graph = C.graph;
	const int slot = (int)(mw - ws->m);
	RBHit& hit = mw->rb[fwi];
	const uint32_t maxsz = P.khits > P.kseeds ? P.khits : P.kseeds;
	const uint32_t minK = g.minK;
	ws->nghits = 0;
	const uint32_t offsetSize = hit.npartial;
	for(uint32_t hi = 0; hi < offsetSize; hi++) {
		uint32_t hj = 0;
		for(; hj < offsetSize; hj++) {
			const PartialHit& pj = hit.partial[hj];
			if(ph_empty(pj) || pj.ncoords > 0 || pj.len <= minK + 2) continue;
			else break;
		}
		if(hj >= offsetSize) break;
		for(uint32_t hk = hj + 1; hk < offsetSize; hk++) {
			const PartialHit& pj = hit.partial[hj];
			const PartialHit& pk = hit.partial[hk];
			if(ph_empty(pk) || pk.ncoords > 0 || pk.len <= minK + 2) continue;
			if(pj.hit_type == pk.hit_type) {
				const uint32_t sj = pj.bot - pj.top, sk = pk.bot - pk.top;
				if(sj > sk || (sj == sk && pj.len < pk.len)) hj = hk;
			} else if(pk.hit_type > pj.hit_type) hj = hk;
		}
		PartialHit& ph = hit.partial[hj];
		const uint32_t remained = maxsz - ws->nghits;
		if(remained == 0) break;
		const GraphPNode* pn = graph ? &C.gws->pnode[slot][fwi][hj] : nullptr;
		uint32_t expected = graph ? pn->node_bot - pn->node_top : ph.bot - ph.top;
		h2g_coord* co = ph.coords;
		uint32_t nco = 0;
		const uint32_t rdoff = hit.len - ph.bwoff - ph.len;
		if(expected <= remained) {
			h2g_sa_result res;
			if(graph) genome_coords_graph_item(g, &C.gws->gw, ph.top, ph.bot, pn->node_top, pn->node_bot, &pn->ie, ph.bot - ph.top, ph.len, false, co, AL_MAX_GHITS, &res);
			else genome_coords_item(g, ph.top, ph.bot, ph.bot - ph.top, ph.len, false, co, AL_MAX_GHITS, &res);
			if(res.nsteps == H2G_MAX) { ws->overflow |= 512; res.nsteps = 0; }
			nco = res.ncoords;
			ws->nsteps += res.nsteps;
		} else if(graph) {   // random sub-sample of `remained` NODES, each with its own rows / extra in-edges (:5096-5136)
			uint32_t edgeIdx = 0, top = ph.top, added = 0;
			for(uint32_t node = pn->node_top; node < pn->node_bot; node++, expected--) {
				uint32_t bot = top + 1;
				IEdges& t = C.gws->ie;
				t.n = 0;
				if(edgeIdx < pn->ie.n && edgeIdx < H2G_IEDGE_CAP) {
					if(node - pn->node_top == pn->ie.e[edgeIdx][0]) {
						bot += pn->ie.e[edgeIdx][1];
						t.n = 1; t.e[0][0] = 0; t.e[0][1] = pn->ie.e[edgeIdx][1];
						edgeIdx++;
					}
				}
				uint32_t rndi = rnd->nextU32() % expected;
				if(rndi < remained - added) {
					h2g_sa_result res;
					if(nco < AL_MAX_GHITS) {
						genome_coords_graph_item(g, &C.gws->gw, top, bot, node, node + 1, &t, ph.bot - ph.top, ph.len, false, co + nco, AL_MAX_GHITS - nco, &res);
						if(res.nsteps == H2G_MAX) { ws->overflow |= 512; res.nsteps = 0; }
						nco += res.ncoords;
						ws->nsteps += res.nsteps;
					} else ws->overflow |= 64;
					added++;
					if(added >= remained) break;
				}
				top = bot;
			}
		} else {   // random sub-sample of `remained` rows (:5096-5136)
			uint32_t top = ph.top, added = 0;
			for(uint32_t node = ph.top; node < ph.bot; node++, expected--) {
				uint32_t bot = top + 1;
				uint32_t rndi = rnd->nextU32() % expected;
				if(rndi < remained - added) {
					h2g_sa_result res;
					if(nco < AL_MAX_GHITS) {
						genome_coords_item(g, top, bot, ph.bot - ph.top, ph.len, false, co + nco, AL_MAX_GHITS - nco, &res);
						nco += res.ncoords;
						ws->nsteps += res.nsteps;
					} else ws->overflow |= 64;
					added++;
					if(added >= remained) break;
				}
				top = bot;
			}
		}
		AL_TRACE("   anchor hj %u nco %u expected %u remained %u\n", hj, nco, expected, remained);
		ph.ncoords = nco;
		if(nco == 0) continue;                       // !hasGenomeCoords()
		const uint32_t genomeHit_size = ws->nghits;
		if(genomeHit_size + nco > maxsz) {           // coords.shufflePortion(0, size, rnd) ds.h:836
			uint32_t left = nco;
			for(uint32_t i = 0; i + 1 < nco; i++) {
				uint32_t r = rnd->nextU32() % left;
				if(r > 0) { h2g_coord t = co[i]; co[i] = co[i + r]; co[i + r] = t; }
				left--;
			}
		}
		for(uint32_t k = 0; k < nco; k++) {
			if(co[k].tidx == H2G_MAX) continue;
			const uint32_t len = ph.len;
			bool overlapped = false;
			for(uint32_t l = 0; l < genomeHit_size; l++) {
				h2g_ghit& gh = ws->ghits[l];
				if(gh.tidx != co[k].tidx || (gh.fw != 0) != seq.fw) continue;
				const uint32_t hitoff = gh.toff + hit.len - gh.rdoff;
				const uint32_t hitoff2 = co[k].toff + hit.len - rdoff;
				const int64_t diff = P.no_spliced ? 0 : (int64_t)P.maxIntronLen;
				int64_t d = (int64_t)hitoff - (int64_t)hitoff2;
				if(d < 0) d = -d;
				if(d <= diff) { overlapped = true; gh.read++; break; }   // _hitcount++
			}
			if(!overlapped) {
				if(graph) {                                        // adjustWithALT may add several (or no) hits (:5175)
					uint32_t ovf = 0;
					adjust_with_alt(g, *C.ref, *C.alts, seq, rdoff, len, co[k].tidx, co[k].toff, co[k].joinedOff, ws->ghits, &ws->nghits, AL_MAX_GHITS,
					                &C.gws->awa, &ovf);
					if(ovf) ws->overflow |= 64;
				} else if(ws->nghits < AL_MAX_GHITS) hit_init(&ws->ghits[ws->nghits++], seq.fw, rdoff, len, co[k].tidx, co[k].toff, co[k].joinedOff);
				else ws->overflow |= 64;
			}
			if(ph.hit_type == H2G_CANDIDATE_HIT && ws->nghits >= maxsz) break;
		}
		if(ph.hit_type == H2G_CANDIDATE_HIT && ws->nghits >= maxsz) break;
	}
	return ws->nghits;
}

H2G_HD bool al_adjust_member(const AlnCtx& C, const SeqView& seq, h2g_ghit* t, AlignWS* ws) {
	if(!C.graph) return true;
	uint32_t ovf = 0;
	const bool ok = adjust_with_alt_member(*C.g, *C.ref, *C.alts, seq, t, &C.gws->awa, &ovf);
	if(ovf) ws->overflow |= 1;
	return ok;
}
H2G_HD uint32_t al_local_search(const AlnCtx& C, AlignWS* ws, uint32_t lidx, const SeqView& seq, uint32_t extoff, uint32_t* extlen,
                                uint32_t* top, uint32_t* bot, bool* uniqueStop, uint32_t maxHitLen)
{
	const AlnParams& P = *C.P;
	LIdx lx; lx.ls = C.ls; lx.d = &C.ls->desc[lidx];
	if(!C.graph) return gfm_search(lx, seq, extoff, extlen, top, bot, uniqueStop, P.minK_local, maxHitLen, P.kseeds, true, &ws->nrank);
	const LGfm x = lgfm_of(*C.ls, *lx.d);
	GRange r;
	r.top = *top; r.bot = *bot; r.node_top = r.node_bot = 0;
	const uint32_t nelt = gfm_search_graph(x, lx, seq, extoff, extlen, &r, &C.gws->ie, uniqueStop, P.minK_local, maxHitLen, P.kseeds, true,
	                                       P.kseeds, &ws->nrank);
	*top = r.top; *bot = r.bot;
	C.gws->node_top = r.node_top; C.gws->node_bot = r.node_bot;
	return nelt;
}
H2G_HD void al_local_coords(const AlnCtx& C, AlignWS* ws, uint32_t lidx, uint32_t top, uint32_t bot, uint32_t rdoff, uint32_t rdlen,
                            h2g_coord* coords, uint32_t cap, uint32_t* ncoords)
{
	LIdx lx; lx.ls = C.ls; lx.d = &C.ls->desc[lidx];
	if(!C.graph) { genome_coords_local(lx, top, bot, rdoff, rdlen, coords, cap, ncoords, &ws->nsteps); return; }
	const LGfm x = lgfm_of(*C.ls, *lx.d);
	const uint32_t node_top = C.gws->node_top, node_bot = C.gws->node_bot;
	uint32_t nelt = 0, n = 0;
	*ncoords = 0;
	if(!gw_resolve(x, &C.gws->gw, top, bot, node_top, node_bot, &C.gws->ie, bot - top, &nelt)) { ws->overflow |= 512; return; }
	ws->nsteps += C.gws->gw.nsteps;
	AL_TRACE("     lcoords top %u bot %u node %u %u -> nelt %u\n", top, bot, node_top, node_bot, nelt);
	for(uint32_t e = 0; e < nelt; e++) {
		h2g_coord c;
		AL_TRACE("      off %u\n", C.gws->gw.offs[e]);
		if(!local_joff_to_coord(*C.ls, lx.d, C.gws->gw.offs[e] & 0xffffu, rdoff, rdlen, &c)) continue;
		if(n < cap) coords[n++] = c; else ws->overflow |= 512;
	}
	*ncoords = n;
}
H2G_HD uint32_t al_global_search(const AlnCtx& C, AlignWS* ws, const SeqView& seq, uint32_t extoff, uint32_t* extlen, uint32_t* top,
                                 uint32_t* bot, bool* uniqueStop)
{
	const AlnParams& P = *C.P;
	GIdx gx; gx.g = C.g;
	if(!C.graph) return gfm_search(gx, seq, extoff, extlen, top, bot, uniqueStop, C.g->minK, H2G_MAX, P.kseeds, false, &ws->nrank);
	GRange r;
	r.top = *top; r.bot = *bot; r.node_top = r.node_bot = 0;
	const uint32_t nelt = gfm_search_graph(*C.g, gx, seq, extoff, extlen, &r, &C.gws->ie, uniqueStop, C.g->minK, H2G_MAX, P.kseeds, false,
	                                       P.kseeds, &ws->nrank);
	if(nelt > 0) { *top = r.top; *bot = r.bot; }
	C.gws->node_top = r.node_top; C.gws->node_bot = r.node_bot;
	return nelt;
}
H2G_HD uint32_t al_global_coords(const AlnCtx& C, AlignWS* ws, uint32_t top, uint32_t bot, uint32_t extlen, h2g_coord* coords, uint32_t cap) {
	h2g_sa_result res;
	if(!C.graph) genome_coords_item(*C.g, top, bot, bot - top, extlen, true, coords, cap, &res);
	else {
		genome_coords_graph_item(*C.g, &C.gws->gw, top, bot, C.gws->node_top, C.gws->node_bot, &C.gws->ie, bot - top, extlen, true,
		                         coords, cap, &res);
		if(res.nsteps == H2G_MAX) { ws->overflow |= 512; res.nsteps = 0; }
	}
	ws->nsteps += res.nsteps;
	return res.ncoords;
}


// ---------------------------------------------------------------------------------------- hybridSearch_recur (a21)
enum {
	ST_ENTRY = 0,
	ST_L_WHILE, ST_L_FOR_RI, ST_L_R1, ST_L_AFTER_FOR, ST_L_FOR_TI, ST_L_R2, ST_L_AFTER_WHILE, ST_L_FOR_G, ST_L_R3, ST_L_TRIM, ST_L_R4,
	ST_L_EXT, ST_L_R5,
	ST_R_WHILE, ST_R_FOR_RI, ST_R_R1, ST_R_AFTER_FOR, ST_R_FOR_TI, ST_R_R2, ST_R_AFTER_WHILE, ST_R_FOR_G, ST_R_R3, ST_R_TRIM, ST_R_R4,
	ST_R_EXT, ST_R_R5
};

H2G_HD uint32_t local_index_of(const DLocalSet& ls, uint32_t tidx, uint32_t toff) {   // HGFM::getLocalGFM hgfm.h:1713
	uint32_t a = ls.first[tidx], b = ls.first[tidx + 1];
	uint32_t k = toff / H2G_LOCAL_INTERVAL;
	if(a + k >= b) return H2G_MAX;
	return a + k;
}
H2G_HD uint32_t local_index_prev(const DLocalSet& ls, uint32_t lidx) {   // prevLocalGFM hgfm.h:1724
	const DLocalDesc& d = ls.desc[lidx];
	if(d.localOffset < H2G_LOCAL_INTERVAL) return H2G_MAX;
	return local_index_of(ls, d.tidx, d.localOffset - H2G_LOCAL_INTERVAL);
}
H2G_HD uint32_t local_index_next(const DLocalSet& ls, uint32_t lidx) {   // nextLocalGFM hgfm.h:1735
	const DLocalDesc& d = ls.desc[lidx];
	return local_index_of(ls, d.tidx, d.localOffset + H2G_LOCAL_INTERVAL);
}

H2G_HD void sort_coords(h2g_coord* c, uint32_t n) {   // Coord::operator< ref_coord.h:79 (same ref/orient): by offset
	for(uint32_t i = 1; i < n; i++) {
		h2g_coord x = c[i];
		int j = (int)i - 1;
		while(j >= 0 && (c[j].tidx > x.tidx || (c[j].tidx == x.tidx && c[j].toff > x.toff))) { c[j + 1] = c[j]; j--; }
		c[j + 1] = x;
	}
}

// Runs hybridSearch_recur(hit, hitoff, hitlen) to completion; returns maxsc.
H2G_HD int64_t al_hybrid_search_recur(const AlnCtx& C, const SeqView& seq, AlignWS* ws, MateWS* mw, const h2g_ghit* root,
                                      uint32_t hitoff0, uint32_t hitlen0, int64_t minsc, bool alignMate)
{
	const AlnParams& P = *C.P;
	const DScoring& sc = P.sc;
	const uint32_t rdlen = seq.len, minK = C.g->minK, minK_local = P.minK_local;
	const bool no_spliced = P.no_spliced != 0;
	// spliced_aligner.h:363-366: cushion = alignMate ? rdlen * 0.03 * sc.mm(255) : 0 (no_spliced_alignment only)
	const int64_t cushion = (no_spliced && alignMate) ? (int64_t)((double)rdlen * 0.03 * (double)sc.mmpMax) : 0;
	int sp = 0;
	int64_t ret = INT64_MIN;
	{
		Frame& f = ws->stack[0];
		hit_copy(&f.hit, root);
		f.hitoff = hitoff0; f.hitlen = hitlen0; f.state = ST_ENTRY;
	}
#define AL_CALL(HITPTR, HOFF, HLEN, RESUME) do { \
		f.state = (RESUME); \
		if(sp + 1 >= AL_MAX_DEPTH) { ws->overflow |= 8; ret = INT64_MIN; } \
		else { Frame& nf = ws->stack[sp + 1]; hit_copy(&nf.hit, (HITPTR)); nf.hitoff = (HOFF); nf.hitlen = (HLEN); nf.state = ST_ENTRY; sp++; \
		       if((uint32_t)sp + 1 > ws->nframes_max) ws->nframes_max = sp + 1; } \
		goto next_iter; } while(0)
#define AL_RET(V) do { ret = (V); sp--; goto next_iter; } while(0)
#define AL_MINSC_LIVE(M) do { if(!P.secondary) { int64_t b_ = mw->bestUnp - cushion; if(b_ > (M)) (M) = b_; } } while(0)

	while(sp >= 0) {
		{
		Frame& f = ws->stack[sp];
		const h2g_ghit& hit = f.hit;
		const uint32_t hitoff = f.hitoff, hitlen = f.hitlen;
		const uint32_t dep = (uint32_t)sp;
		switch(f.state) {
		case ST_ENTRY: {
			AL_TRACE("   recur dep %u fw %u hitoff %u hitlen %u (rdoff %u len %u) toff %u score %lld nedits %u mate %d\n", dep, hit.fw, hitoff, hitlen, hit.rdoff, hit.len, hit.toff, (long long)hit.score, hit.nedits, (int)alignMate);
			f.maxsc = INT64_MIN;
			if(hit.score + cushion < minsc) AL_RET(f.maxsc);
			if(dep >= 128) AL_RET(f.maxsc);
			if(hitoff == hit.rdoff - hit.trim5 && hitlen == hit.len + hit.trim5 + hit.trim3) {
				if(al_is_searched(mw, &hit)) AL_RET(f.maxsc);
				al_add_searched(ws, mw, &hit);
			}
			if(hitoff == 0 && hitlen == rdlen) {
				if(!al_redundant(mw, &hit, rdlen)) {
					al_report(ws, mw, &hit, rdlen, minsc);
					if(hit.score > f.maxsc) f.maxsc = hit.score;
				}
				AL_RET(f.maxsc);
			} else if(hitoff > 0 && (hitoff + hitlen == rdlen || hitoff + hitoff < rdlen - hitlen)) {
				// ---------------- extend to the left (spliced_aligner.h:813-1360) ----------------
				f.use_localindex = 1;
				if(hitoff == hit.rdoff && hitoff <= minK) {
					hit_copy(&ws->tmp, &hit);
					uint32_t le, re;
					al_extend(C, seq, &ws->tmp, 1, H2G_MAX, 0, &le, &re);
					if(ws->tmp.rdoff == 0) f.use_localindex = 0;
				}
				f.lidx = local_index_of(*C.ls, hit.tidx, hit.toff);
				f.success = 0; f.first = 1; f.count = 0; f.prev_score = hit.score; f.nlocal = 0;
				f.state = ST_L_WHILE;
				goto next_iter;
			} else {
				// ---------------- extend to the right (spliced_aligner.h:1496-2050) ----------------
				f.use_localindex = 1;
				if(hit.len == hitlen && hitoff + hitlen + minK > rdlen) {
					hit_copy(&ws->tmp, &hit);
					uint32_t le, re;
					al_extend(C, seq, &ws->tmp, 1, 0, H2G_MAX, &le, &re);
					if(ws->tmp.rdoff + ws->tmp.len == rdlen) f.use_localindex = 0;
				}
				f.lidx = local_index_of(*C.ls, hit.tidx, hit.toff);
				f.success = 0; f.first = 1; f.count = 0; f.prev_score = hit.score; f.nlocal = 0;
				f.state = ST_R_WHILE;
				goto next_iter;
			}
		}
		// =============================== LEFT ===============================
		case ST_L_WHILE: {
			if(f.success) { f.state = ST_L_AFTER_WHILE; goto next_iter; }
			if(!(f.count++ < 2)) { f.state = ST_L_AFTER_WHILE; goto next_iter; }
			if(!f.use_localindex) { f.state = ST_L_AFTER_WHILE; goto next_iter; }
			if(ws->localindexatts >= ws->max_localindexatts) { f.state = ST_L_AFTER_WHILE; goto next_iter; }
			if(f.first) f.first = 0;
			else {
				f.lidx = f.lidx == H2G_MAX ? H2G_MAX : local_index_prev(*C.ls, f.lidx);
				if(f.lidx == H2G_MAX || C.ls->desc[f.lidx].len == 0) { f.state = ST_L_AFTER_WHILE; goto next_iter; }
			}
			if(f.lidx == H2G_MAX) { f.state = ST_L_AFTER_WHILE; goto next_iter; }
			uint32_t extlen = 0, top = H2G_MAX, bot = H2G_MAX;
			uint32_t extoff = hitoff - 1;
			if(extoff > 0) extoff -= 1;
			if(extoff < P.minAnchorLen) extoff = P.minAnchorLen;
			uint32_t nelt = H2G_MAX;
			const uint32_t max_nelt = 5;
			bool no_extension = false, uniqueStop = false;
			LIdx lx; lx.ls = C.ls; lx.d = &C.ls->desc[f.lidx];
			for(; extoff < rdlen; extoff++) {
				extlen = 0; uniqueStop = true;
				ws->localindexatts++;
				nelt = C.ls->desc[f.lidx].len == 0 ? 0 :
				       al_local_search(C, ws, f.lidx, seq, extoff, &extlen, &top, &bot, &uniqueStop, 0xffffu);
				if(extoff + 1 - extlen >= hitoff) { no_extension = true; break; }
				if(nelt <= max_nelt) break;
			}
			f.ncoords = 0; f.ri = -1;
			f.extoff = extoff; f.extlen = extlen; f.uniqueStop = uniqueStop;
			AL_TRACE("    L local lidx %u extoff %u extlen %u nelt %u top %u bot %u unique %d noext %d\n", f.lidx, extoff, extlen, nelt, top, bot, (int)uniqueStop, (int)no_extension);
			if(nelt > 0 && nelt <= max_nelt && extlen >= P.minAnchorLen && !no_extension) {
				al_local_coords(C, ws, f.lidx, top, bot, extoff + 1 - extlen, extlen, f.coords, AL_MAX_COORDS, &f.ncoords);
				sort_coords(f.coords, f.ncoords);
				f.ri = (int)f.ncoords - 1;
			}
			f.state = ST_L_FOR_RI;
			goto next_iter;
		}
		case ST_L_FOR_RI: {
			if(f.ri < 0) { f.state = ST_L_AFTER_FOR; goto next_iter; }
			const h2g_coord co = f.coords[f.ri];
			h2g_ghit* t = &ws->tmp;
			hit_init(t, hit.fw, f.extoff + 1 - f.extlen, f.extlen, co.tidx, co.toff, co.joinedOff);
			if(!al_adjust_member(C, seq, t, ws)) { f.ri--; goto next_iter; }
			if(!hit_compatible(t, &hit, P.maxIntronLen, no_spliced)) {
				if(f.count == 1) { f.ri--; goto next_iter; }
				f.state = ST_L_AFTER_FOR; goto next_iter;
			}
			AL_TRACE("    L coord tidx %u toff %u -> adjusted rdoff %u len %u toff %u nedits %u\n", co.tidx, co.toff, t->rdoff, t->len, t->toff, t->nedits);
			if(f.uniqueStop) { uint32_t le, re; al_extend(C, seq, t, 0, H2G_MAX, 0, &le, &re); }
			AL_TRACE("    L extended rdoff %u len %u toff %u nedits %u\n", t->rdoff, t->len, t->toff, t->nedits);
			int64_t m = minsc;
			bool combined = hit_combine(*C.ref, sc, seq, t, &hit, m, P.minIntronLen, no_spliced, ws->sc1, ws->sc2, C.alts);
			AL_TRACE("    L combined %d rdoff %u len %u score %lld nedits %u\n", (int)combined, t->rdoff, t->len, (long long)t->score, t->nedits);
			if(t->overflow) ws->overflow |= 1;
			AL_MINSC_LIVE(m);
			f.ri--;
			if(combined && t->score >= m) {
				if(t->score >= f.prev_score - sc.mmpMax) AL_CALL(t, t->rdoff, t->len + t->trim3, ST_L_R1);
				else if(f.nlocal < AL_MAX_LOCALHITS) hit_copy(&f.local_hits[f.nlocal++], t);
				else ws->overflow |= 16;
			}
			goto next_iter;
		}
		case ST_L_R1: { if(ret > f.maxsc) f.maxsc = ret; f.state = ST_L_FOR_RI; goto next_iter; }
		case ST_L_AFTER_FOR: {
			if(f.maxsc >= f.prev_score - sc.mmpMax) f.success = 1;
			f.ti = 0;
			if(!f.success && (ws->localindexatts >= ws->max_localindexatts || f.count == 2 ||
			                  (f.lidx == H2G_MAX || local_index_prev(*C.ls, f.lidx) == H2G_MAX)))
				f.state = ST_L_FOR_TI;
			else f.state = ST_L_WHILE;
			goto next_iter;
		}
		case ST_L_FOR_TI: {
			if(f.ti >= f.nlocal) { f.state = ST_L_WHILE; goto next_iter; }
			h2g_ghit* t = &f.local_hits[f.ti++];
			int64_t m = minsc;
			AL_MINSC_LIVE(m);
			if(t->score >= m) AL_CALL(t, t->rdoff, t->len + t->trim3, ST_L_R2);
			goto next_iter;
		}
		case ST_L_R2: { if(ret > f.maxsc) f.maxsc = ret; f.state = ST_L_FOR_TI; goto next_iter; }
		case ST_L_AFTER_WHILE: {
			if(f.success) AL_RET(f.maxsc);
			f.ncoords = 0; f.ri = -1;
			if(hitoff > minK && ws->localindexatts < ws->max_localindexatts) {   // global search for long introns (:1085)
				uint32_t extlen = 0, top = H2G_MAX, bot = H2G_MAX;
				const uint32_t extoff = hitoff - 1;
				bool uniqueStop = true;
				GIdx gx; gx.g = C.g;
				uint32_t nelt = al_global_search(C, ws, seq, extoff, &extlen, &top, &bot, &uniqueStop);
				f.extoff = extoff; f.extlen = extlen; f.uniqueStop = uniqueStop;
				AL_TRACE("    L global extoff %u extlen %u nelt %u top %u bot %u unique %d\n", extoff, extlen, nelt, top, bot, (int)uniqueStop);
				if(nelt > 0 && nelt <= 5 && extlen >= minK) {
					f.ncoords = al_global_coords(C, ws, top, bot, extlen, f.coords, AL_MAX_COORDS);
					if(f.ncoords > 1) sort_coords(f.coords, f.ncoords);
					f.ri = (int)f.ncoords - 1;
				}
			}
			f.state = ST_L_FOR_G;
			goto next_iter;
		}
		case ST_L_FOR_G: {
			if(f.ri < 0) { f.state = ST_L_TRIM; goto next_iter; }
			const h2g_coord co = f.coords[f.ri];
			f.ri--;
			h2g_ghit* t = &ws->tmp;
			hit_init(t, hit.fw, f.extoff + 1 - f.extlen, f.extlen, co.tidx, co.toff, co.joinedOff);
			AL_TRACE("    LG coord tidx %u toff %u joff %u rdoff %u len %u\n", co.tidx, co.toff, co.joinedOff, t->rdoff, t->len);
			if(!al_adjust_member(C, seq, t, ws)) goto next_iter;
			AL_TRACE("    LG adjusted rdoff %u len %u toff %u joff %u nedits %u\n", t->rdoff, t->len, t->toff, t->joinedOff, t->nedits);
			if(!hit_compatible(t, &hit, P.maxIntronLen, no_spliced)) goto next_iter;
			if(f.uniqueStop) { uint32_t le, re; al_extend(C, seq, t, 0, H2G_MAX, 0, &le, &re); }
			AL_TRACE("    LG extended rdoff %u len %u toff %u joff %u nedits %u\n", t->rdoff, t->len, t->toff, t->joinedOff, t->nedits);
			for(uint32_t q = 0; q < t->nedits; q++) AL_TRACE("       edit %u %c>%c type %u snp %u\n", t->edits[q].pos, t->edits[q].chr, t->edits[q].qchr, t->edits[q].type, t->edits[q].snp);
			int64_t m = minsc;
			bool combined = hit_combine(*C.ref, sc, seq, t, &hit, m, P.minIntronLen, no_spliced, ws->sc1, ws->sc2, C.alts);
			AL_TRACE("    LG combined %d rdoff %u len %u score %lld nedits %u\n", (int)combined, t->rdoff, t->len, (long long)t->score, t->nedits);
			for(uint32_t q = 0; q < t->nedits; q++) AL_TRACE("       edit %u %c>%c type %u snp %u\n", t->edits[q].pos, t->edits[q].chr, t->edits[q].qchr, t->edits[q].type, t->edits[q].snp);
			if(t->overflow) ws->overflow |= 1;
			AL_MINSC_LIVE(m);
			if(combined && t->score >= m) AL_CALL(t, t->rdoff, t->len + t->trim3, ST_L_R3);
			goto next_iter;
		}
		case ST_L_R3: { if(ret > f.maxsc) f.maxsc = ret; f.state = ST_L_FOR_G; goto next_iter; }
		case ST_L_TRIM: {
			const int64_t floor_ = f.maxsc > minsc ? f.maxsc : minsc;
			const int64_t tm = (hit.score - floor_) / sc_penalty(sc, 0);
			const uint32_t trimMax = (uint32_t)tm;
			f.state = ST_L_EXT;
			if(hit.rdoff < trimMax) {
				h2g_ghit* t = &ws->tmp;
				hit_copy(t, &hit);
				t->trim5 = hit.rdoff;                         // GenomeHit::trim5 hi_aligner.h:831
				calculate_score(sc, seq, t);
				if(t->score > f.maxsc && t->score >= minsc) AL_CALL(t, 0, t->len + t->trim5 + t->trim3, ST_L_R4);
			}
			goto next_iter;
		}
		case ST_L_R4: { if(ret > f.maxsc) f.maxsc = ret; f.state = ST_L_EXT; goto next_iter; }
		case ST_L_EXT: {
			h2g_ghit* t = &ws->tmp;
			hit_copy(t, &hit);
			int64_t m = minsc;
			const uint32_t mm = (uint32_t)((t->score - m) / sc.mmpMax);
			uint32_t nmm = 1;
			if(hitoff <= minK_local) nmm = t->rdoff < mm ? t->rdoff : mm;
			uint32_t le = 0, re = 0;
			AL_TRACE("    L ext from rdoff %u len %u toff %u joff %u nmm %u\n", t->rdoff, t->len, t->toff, t->joinedOff, nmm);
			al_extend(C, seq, t, nmm, H2G_MAX, 0, &le, &re);
			AL_TRACE("    L ext -> rdoff %u len %u toff %u joff %u score %lld nedits %u le %u\n", t->rdoff, t->len, t->toff, t->joinedOff, (long long)t->score, t->nedits, le);
			if(t->overflow) ws->overflow |= 1;
			AL_MINSC_LIVE(m);
			const uint32_t need = minK_local < hit.rdoff ? minK_local : hit.rdoff;
			if(t->score >= m && le >= need) AL_CALL(t, t->rdoff, t->len + t->trim3, ST_L_R5);
			else if(hitoff > minK_local) {
				const uint32_t jumplen = hitoff > minK ? minK : minK_local;
				const int64_t expected = hit.score - (int64_t)((hit.rdoff - hitoff) / jumplen) * sc.mmpMax - sc.mmpMax;
				if(expected >= m) AL_CALL(&hit, hitoff - jumplen, hitlen + jumplen, ST_L_R5);
			}
			AL_RET(f.maxsc);
		}
		case ST_L_R5: { if(ret > f.maxsc) f.maxsc = ret; AL_RET(f.maxsc); }
		// =============================== RIGHT ===============================
		case ST_R_WHILE: {
			if(f.success) { f.state = ST_R_AFTER_WHILE; goto next_iter; }
			if(!(f.count++ < 2)) { f.state = ST_R_AFTER_WHILE; goto next_iter; }
			if(!f.use_localindex) { f.state = ST_R_AFTER_WHILE; goto next_iter; }
			if(ws->localindexatts >= ws->max_localindexatts) { f.state = ST_R_AFTER_WHILE; goto next_iter; }
			if(f.first) f.first = 0;
			else {
				f.lidx = f.lidx == H2G_MAX ? H2G_MAX : local_index_next(*C.ls, f.lidx);
				if(f.lidx == H2G_MAX || C.ls->desc[f.lidx].len == 0) { f.state = ST_R_AFTER_WHILE; goto next_iter; }
			}
			if(f.lidx == H2G_MAX) { f.state = ST_R_AFTER_WHILE; goto next_iter; }
			uint32_t extlen = 0, top = H2G_MAX, bot = H2G_MAX;
			uint32_t extoff = hitoff + hitlen + minK_local;
			if(extoff + 1 < rdlen) extoff += 1;
			if(extoff >= rdlen) extoff = rdlen - 1;
			uint32_t nelt = H2G_MAX;
			const uint32_t max_nelt = 5;
			bool no_extension = false, uniqueStop = false;
			uint32_t maxHitLen = extoff - hitoff - hitlen;
			if(maxHitLen < minK_local) maxHitLen = minK_local;
			LIdx lx; lx.ls = C.ls; lx.d = &C.ls->desc[f.lidx];
			for(; maxHitLen < extoff + 1 && extoff < rdlen;) {
				extlen = 0; uniqueStop = false;
				ws->localindexatts++;
				nelt = C.ls->desc[f.lidx].len == 0 ? 0 :
				       al_local_search(C, ws, f.lidx, seq, extoff, &extlen, &top, &bot, &uniqueStop, maxHitLen);
				if(extoff < hitoff + hitlen) { no_extension = true; break; }
				if(nelt <= max_nelt) break;
				if(extoff + 1 < rdlen) extoff++;
				else { if(extlen < maxHitLen) break; else maxHitLen++; }
			}
			f.ncoords = 0; f.ri = 0;
			f.extoff = extoff; f.extlen = extlen; f.uniqueStop = uniqueStop;
			if(nelt > 0 && nelt <= max_nelt && extlen >= P.minAnchorLen && !no_extension) {
				al_local_coords(C, ws, f.lidx, top, bot, extoff + 1 - extlen, extlen, f.coords, AL_MAX_COORDS, &f.ncoords);
				if(f.ncoords > 1) sort_coords(f.coords, f.ncoords);
			}
			f.state = ST_R_FOR_RI;
			goto next_iter;
		}
		case ST_R_FOR_RI: {
			if(f.ri >= (int)f.ncoords) { f.state = ST_R_AFTER_FOR; goto next_iter; }
			const h2g_coord co = f.coords[f.ri];
			h2g_ghit* t = &ws->tmp;
			hit_init(t, hit.fw, f.extoff + 1 - f.extlen, f.extlen, co.tidx, co.toff, co.joinedOff);
			if(!al_adjust_member(C, seq, t, ws)) { f.ri++; goto next_iter; }
			if(!hit_compatible(&hit, t, P.maxIntronLen, no_spliced)) {
				if(f.count == 1) { f.ri++; goto next_iter; }
				f.state = ST_R_AFTER_FOR; goto next_iter;
			}
			{ uint32_t le, re; al_extend(C, seq, t, 0, 0, H2G_MAX, &le, &re); }
			h2g_ghit* cmb = &ws->tmp2;
			hit_copy(cmb, &hit);
			int64_t m = minsc;
			bool combined = hit_combine(*C.ref, sc, seq, cmb, t, m, P.minIntronLen, no_spliced, ws->sc1, ws->sc2, C.alts);
			if(cmb->overflow) ws->overflow |= 1;
			AL_MINSC_LIVE(m);
			f.ri++;
			if(combined && cmb->score >= m) {
				if(cmb->score >= f.prev_score - sc.mmpMax) AL_CALL(cmb, cmb->rdoff - cmb->trim5, cmb->len + cmb->trim5, ST_R_R1);
				else if(f.nlocal < AL_MAX_LOCALHITS) hit_copy(&f.local_hits[f.nlocal++], cmb);
				else ws->overflow |= 16;
			}
			goto next_iter;
		}
		case ST_R_R1: { if(ret > f.maxsc) f.maxsc = ret; f.state = ST_R_FOR_RI; goto next_iter; }
		case ST_R_AFTER_FOR: {
			if(f.maxsc >= f.prev_score - sc.mmpMax) f.success = 1;
			f.ti = 0;
			if(!f.success && (ws->localindexatts >= ws->max_localindexatts || f.count == 2 ||
			                  (f.lidx == H2G_MAX || local_index_next(*C.ls, f.lidx) == H2G_MAX)))
				f.state = ST_R_FOR_TI;
			else f.state = ST_R_WHILE;
			goto next_iter;
		}
		case ST_R_FOR_TI: {
			if(f.ti >= f.nlocal) { f.state = ST_R_WHILE; goto next_iter; }
			h2g_ghit* t = &f.local_hits[f.ti++];
			int64_t m = minsc;
			AL_MINSC_LIVE(m);
			if(t->score >= m) AL_CALL(t, t->rdoff - t->trim5, t->len + t->trim5, ST_R_R2);
			goto next_iter;
		}
		case ST_R_R2: { if(ret > f.maxsc) f.maxsc = ret; f.state = ST_R_FOR_TI; goto next_iter; }
		case ST_R_AFTER_WHILE: {
			if(f.success) AL_RET(f.maxsc);
			f.ncoords = 0; f.ri = 0;
			if(hitoff + hitlen + minK + 1 < rdlen && ws->localindexatts < ws->max_localindexatts) {
				uint32_t extlen = 0, top = H2G_MAX, bot = H2G_MAX;
				const uint32_t extoff = hitoff + hitlen + minK + 1;
				bool uniqueStop = true;
				GIdx gx; gx.g = C.g;
				uint32_t nelt = al_global_search(C, ws, seq, extoff, &extlen, &top, &bot, &uniqueStop);
				f.extoff = extoff; f.extlen = extlen; f.uniqueStop = uniqueStop;
				if(nelt > 0 && nelt <= 5 && extlen >= minK) {
					f.ncoords = al_global_coords(C, ws, top, bot, extlen, f.coords, AL_MAX_COORDS);
					sort_coords(f.coords, f.ncoords);
				}
			}
			f.state = ST_R_FOR_G;
			goto next_iter;
		}
		case ST_R_FOR_G: {
			if(f.ri >= (int)f.ncoords) { f.state = ST_R_TRIM; goto next_iter; }
			const h2g_coord co = f.coords[f.ri];
			f.ri++;
			h2g_ghit* t = &ws->tmp;
			hit_init(t, hit.fw, f.extoff + 1 - f.extlen, f.extlen, co.tidx, co.toff, co.joinedOff);
			if(!al_adjust_member(C, seq, t, ws)) goto next_iter;
			if(!hit_compatible(&hit, t, P.maxIntronLen, no_spliced)) goto next_iter;
			{ uint32_t le, re; al_extend(C, seq, t, 0, 0, H2G_MAX, &le, &re); }
			h2g_ghit* cmb = &ws->tmp2;
			hit_copy(cmb, &hit);
			int64_t m = minsc;
			bool combined = hit_combine(*C.ref, sc, seq, cmb, t, m, P.minIntronLen, no_spliced, ws->sc1, ws->sc2, C.alts);
			if(cmb->overflow) ws->overflow |= 1;
			AL_MINSC_LIVE(m);
			if(combined && cmb->score >= m) AL_CALL(cmb, cmb->rdoff - cmb->trim5, cmb->len + cmb->trim5, ST_R_R3);
			goto next_iter;
		}
		case ST_R_R3: { if(ret > f.maxsc) f.maxsc = ret; f.state = ST_R_FOR_G; goto next_iter; }
		case ST_R_TRIM: {
			const uint32_t trimLen = rdlen - hitoff - hit.len - hit.trim5;
			const int64_t floor_ = f.maxsc > minsc ? f.maxsc : minsc;
			const uint32_t trimMax = (uint32_t)((hit.score - floor_) / sc_penalty(sc, 0));
			f.state = ST_R_EXT;
			if(trimLen < trimMax) {
				h2g_ghit* t = &ws->tmp;
				hit_copy(t, &hit);
				t->trim3 = trimLen;                           // GenomeHit::trim3 hi_aligner.h:855
				calculate_score(sc, seq, t);
				if(t->score > f.maxsc && t->score >= minsc)
					AL_CALL(t, t->rdoff - t->trim5, t->len + t->trim5 + t->trim3, ST_R_R4);
			}
			goto next_iter;
		}
		case ST_R_R4: { if(ret > f.maxsc) f.maxsc = ret; f.state = ST_R_EXT; goto next_iter; }
		case ST_R_EXT: {
			h2g_ghit* t = &ws->tmp;
			hit_copy(t, &hit);
			int64_t m = minsc;
			const uint32_t mm = (uint32_t)((t->score - m) / sc.mmpMax);
			uint32_t nmm = 1;
			if(rdlen - hitoff - hitlen <= minK_local) {
				const uint32_t rest = rdlen - t->rdoff - t->len;
				nmm = rest < mm ? rest : mm;
			}
			uint32_t le = 0, re = 0;
			al_extend(C, seq, t, nmm, 0, H2G_MAX, &le, &re);
			if(t->overflow) ws->overflow |= 1;
			AL_MINSC_LIVE(m);
			const uint32_t rest0 = rdlen - hit.len - hit.rdoff;
			const uint32_t need = minK_local < rest0 ? minK_local : rest0;
			if(t->score >= m && re >= need) AL_CALL(t, t->rdoff - t->trim5, t->len + t->trim5, ST_R_R5);
			else if(hitoff + hitlen + minK_local < rdlen) {
				const uint32_t jumplen = hitoff + hitlen + minK < rdlen ? minK : minK_local;
				const int64_t expected = hit.score - (int64_t)((hitlen - hit.len) / jumplen) * sc.mmpMax - sc.mmpMax;
				if(expected >= m) AL_CALL(&hit, hitoff, hitlen + jumplen, ST_R_R5);
			}
			AL_RET(f.maxsc);
		}
		case ST_R_R5: { if(ret > f.maxsc) f.maxsc = ret; AL_RET(f.maxsc); }
		default: AL_RET(INT64_MIN);
		}
		}
	next_iter:;
	}
#undef AL_CALL
#undef AL_RET
#undef AL_MINSC_LIVE
	return ret;
}

// ---------------------------------------------------------------------------------------- go()
// ReadBWTHit::searchScore hi_aligner.h:320-334
H2G_HD int64_t rb_search_score(const RBHit& h, uint32_t minK) {
	int64_t score = 0;
	for(uint32_t i = 0; i < h.npartial; i++) score += (int64_t)h.partial[i].len * h.partial[i].len;
	const uint32_t act = h.numPartialSearch - h.numUniqueSearch;
	score -= (int64_t)act * minK * minK;
	score -= ((int64_t)1 << (act << 1));
	return score;
}

// reference extent of a reported alignment (AlnRes::refExtent): aligned read bases + read gaps - ref gaps
H2G_HD uint32_t rec_ref_extent(const AlnRec& r) {
	uint32_t ext = r.len;
	for(uint32_t k = 0; k < r.nedits; k++) {
		if(r.edits[k].type == H2G_EDIT_READ_GAP) ext++;
		else if(r.edits[k].type == H2G_EDIT_REF_GAP) ext--;
	}
	return ext;
}

// PairedEndPolicy::peClassifyPair pe.cpp:38-133 with the hisat2 defaults (hisat2.cpp:344-352, 3239): --fr,
// -I 0 -X 1000, no dovetail, containment and overlap allowed, expand-to-fit.  Returns true unless PE_ALS_DISCORD.
H2G_HD bool pe_concordant(int64_t off1, uint32_t len1, bool fw1, int64_t off2, uint32_t len2, bool fw2, uint32_t maxfrag_) {
	uint64_t maxfrag = maxfrag_;
	if(len1 > maxfrag) maxfrag = len1;
	if(len2 > maxfrag) maxfrag = len2;
	const uint64_t minfrag = 1;
	if(fw1 == fw2) return false;                 // PE_POLICY_FR
	const bool oneLeft = fw1;
	const int64_t fraglo = off1 < off2 ? off1 : off2;
	const int64_t h1 = off1 + len1, h2 = off2 + len2;
	const int64_t fraghi = h1 > h2 ? h1 : h2;
	const uint64_t frag = (uint64_t)(fraghi - fraglo);
	if(frag > maxfrag || frag < minfrag) return false;
	const int64_t lo1 = off1, hi1 = off1 + len1 - 1, lo2 = off2, hi2 = off2 + len2 - 1;
	const bool containment = (lo1 >= lo2 && hi1 <= hi2) || (lo2 >= lo1 && hi2 <= hi1);
	const bool olap = (lo1 <= lo2 && hi1 >= lo2) || (lo1 <= hi2 && hi1 >= hi2) || containment;
	if(!olap) { if((oneLeft && lo2 < lo1) || (!oneLeft && lo1 < lo2)) return false; }
	if((oneLeft && (hi1 > hi2 || lo2 < lo1)) || (!oneLeft && (hi2 > hi1 || lo1 < lo2))) return false;   // dovetail not allowed
	return true;
}

// pairReads hi_aligner.h:5948-6055 (non-repeat alignments, gMate1fw = true, gMate2fw = false)
H2G_HD void al_pair_reads(const AlnParams& P, AlignWS* ws, uint32_t rdlen1, uint32_t rdlen2) {
	MateWS& m1 = ws->m[0];
	MateWS& m2 = ws->m[1];
	const uint32_t start_i = ws->insp_i, start_j = ws->insp_j;
	ws->insp_i = m1.nres; ws->insp_j = m2.nres;
	for(uint32_t i = 0; i < m1.nres; i++) {
		for(uint32_t j = (i >= start_i ? 0 : start_j); j < m2.nres; j++) {
			const AlnRec& r1 = m1.res[i];
			const AlnRec& r2 = m2.res[j];
			if(r1.tidx != r2.tidx) continue;
			const uint32_t e1 = rec_ref_extent(r1), e2 = rec_ref_extent(r2);
			int64_t l = r1.toff, r = (int64_t)r1.toff + e1 - 1, l2 = r2.toff, rr2 = (int64_t)r2.toff + e2 - 1;
			if(r1.fw) { if(r2.fw) continue; }
			else {
				if(!r2.fw) continue;
				int64_t t = l; l = l2; l2 = t; t = r; r = rr2; rr2 = t;
			}
			if(l > l2) continue;
			if(r > rr2) continue;
			if(r + (int64_t)P.maxIntronLen < l2) continue;
			bool pass = true;
			if(P.no_spliced) {
				if(r1.toff < r2.toff) pass = pe_concordant(r1.toff, e1, r1.fw != 0, r2.toff, e2, r2.fw != 0, P.maxFragLen);
				else                  pass = pe_concordant(r2.toff, e2, r2.fw != 0, r1.toff, e1, r1.fw != 0, P.maxFragLen);
			}
			if(!P.no_spliced || pass) {
				int64_t threshold = ws->bestPair;
				if(m1.bestUnp >= m1.minsc && m2.bestUnp >= m2.minsc) {
					int64_t tmp = (int64_t)((double)(m1.bestUnp + m2.bestUnp) - (double)(rdlen1 + rdlen2) * 0.03 * (double)P.sc.mmpMax);
					if(tmp > threshold) threshold = tmp;
				}
				const int64_t score = r1.score + r2.score;
				if(score >= threshold || P.secondary) {   // sink.report(0, &r1, &r2) aln_sink.h:2590-2612
					if(ws->npairs < AL_MAX_PAIRS) { ws->pair_i[ws->npairs] = (uint8_t)i; ws->pair_j[ws->npairs] = (uint8_t)j; ws->npairs++; }
					else ws->overflow |= 128;
					if(score > ws->bestPair) { ws->best2Pair = ws->bestPair; ws->bestPair = score; }
					else if(score > ws->best2Pair) ws->best2Pair = score;
				}
			}
		}
	}
}

// hybridSearch spliced_aligner.h:112-322 (bowtie2_dp = 0) over ws->ghits
H2G_HD void al_hybrid_search(const AlnCtx& C, const SeqView& sv, AlignWS* ws, MateWS* mw, Rng* rnd) {
	const AlnParams& P = *C.P;
	for(uint32_t hi = 0; hi < ws->nghits; hi++) {
		uint32_t le = H2G_MAX, re = H2G_MAX;
		al_extend(C, sv, &ws->ghits[hi], 0, H2G_MAX, H2G_MAX, &le, &re);
		ws->ghit_done[hi] = 0;
	}
	for(uint32_t hi = 0; hi < ws->nghits; hi++) {
		uint32_t hj = 0;
		for(; hj < ws->nghits; hj++) if(!ws->ghit_done[hj]) break;
		if(hj >= ws->nghits) break;
		for(uint32_t hk = hj + 1; hk < ws->nghits; hk++) {
			if(ws->ghit_done[hk]) continue;
			const h2g_ghit& a = ws->ghits[hj];
			const h2g_ghit& b = ws->ghits[hk];
			if(b.read > a.read || (b.read == a.read && b.len > a.len)) hj = hk;
		}
		h2g_ghit* gh = &ws->ghits[hj];
		const int64_t maxsc = al_hybrid_search_recur(C, sv, ws, mw, gh, gh->rdoff, gh->len, mw->minsc, false);
		// spliced_aligner.h:209-317: the opt-in SwAligner pass (--bowtie2-dp 1: only when nothing reached minsc; 2: always)
		if(P.bowtie2_dp == 2 || (P.bowtie2_dp == 1 && maxsc < mw->minsc)) {
			bool found = gh->len >= sv.len;
			if(!found) {
				if(C.sw == nullptr || sv.len > H2G_SW_MAX_ROWS) ws->overflow |= 256;   // no SW scratch / read longer than the DP path holds
				else {
					SwParams SP;
					SP.sc = P.sc;
					const uint32_t refoff = gh->toff > gh->rdoff ? gh->toff - gh->rdoff : 0;
					SwOut* o = nullptr;
					sw_align_single(*C.ref, SP, sv, gh->tidx, refoff, mw->minsc, &rnd->last, C.sw, &o);
					if(o->overflow) ws->overflow |= 256;
					if(o->found) {
						// res.alres edits: setShape turned them to 5'-end coordinates, `if(!fw) invertEdits()` turns them back
						// to the aligned strand's => exactly the backtrace's own coordinates.  genomeHit.init(fw, 0, rdlen, ...)
						const uint32_t joinedOff = (uint32_t)((int64_t)gh->joinedOff + o->off - (int64_t)gh->toff);
						hit_init(gh, sv.fw, 0, sv.len, gh->tidx, (uint32_t)o->off, joinedOff);
						gh->score = o->score;
						gh->nedits = o->nedits;
						for(uint32_t e = 0; e < o->nedits; e++) gh->edits[e] = o->edits[e];
						if(C.graph && C.alts->n > 0 && gh->nedits > 0) {  // replace_edits_with_alts spliced_aligner.h:282 (re-scores)
							replace_edits_with_alts(*C.alts, gh);
							calculate_score(P.sc, sv, gh);
						}
						found = true;
					}
				}
			}
			if(found) al_hybrid_search_recur(C, sv, ws, mw, gh, gh->rdoff, gh->len, mw->minsc, false);
		}
		ws->ghit_done[hj] = 1;
	}
}

// align hi_aligner.h:5484-5573
H2G_HD bool al_align(const AlnCtx& C, const SeqView& sv, AlignWS* ws, MateWS* mw, int fwi, Rng* rnd) {
	const AlnParams& P = *C.P;
	RBHit& hit = mw->rb[fwi];
	bool any = false;
	for(uint32_t i = 0; i < hit.npartial; i++) if(!ph_empty(hit.partial[i])) { any = true; break; }
	if(!any) return false;                                    // minWidth() == max
	int64_t bestScore = mw->bestUnp;
	if(bestScore < mw->minsc) bestScore = mw->minsc;
	const uint32_t maxmm = (uint32_t)((-bestScore + P.sc.mmpMax - 1) / P.sc.mmpMax);
	const uint32_t nact = hit.numPartialSearch - hit.numUniqueSearch;
	if(!P.secondary && nact > maxmm + 0 + 1) return true;
	uint32_t numHits = al_get_anchor_hits(C, sv, ws, mw, fwi, rnd);
	if(numHits == 0) return false;
	uint64_t add = (uint64_t)((-mw->minsc) / P.sc.mmpMax) * numHits * (P.secondary ? 2 : 1);
	ws->max_localindexatts = ws->localindexatts + (add > 10 ? add : 10);
	al_hybrid_search(C, sv, ws, mw, rnd);
	return true;
}

// alignMate hi_aligner.h:5579-5770: anchor the OTHER mate (ordi) near (tidx, toff) through the local index
H2G_HD void al_align_mate(const AlnCtx& C, const SeqView& ord, AlignWS* ws, MateWS* omw, bool fw, uint32_t tidx, uint32_t toff, Rng* rnd) {
	const AlnParams& P = *C.P;
	const uint32_t rdlen = ord.len, minK_local = P.minK_local;
	ws->nghits = 0;
	uint32_t lidx = local_index_of(*C.ls, tidx, toff);
	bool first = true;
	uint32_t count = 0, max_hitlen = 0;
	while(count++ < 2) {
		if(first) first = false;
		else {
			if(ws->nghits > 0) break;
			if(lidx != H2G_MAX) lidx = fw ? local_index_next(*C.ls, lidx) : local_index_prev(*C.ls, lidx);
			if(lidx == H2G_MAX || C.ls->desc[lidx].len == 0) break;
		}
		if(lidx == H2G_MAX) break;
		LIdx lx; lx.ls = C.ls; lx.d = &C.ls->desc[lidx];
		uint32_t hitoff = rdlen - 1;
		while(hitoff >= minK_local - 1) {
			uint32_t hitlen = 0, top = H2G_MAX, bot = H2G_MAX;
			bool uniqueStop = false;
			uint32_t nelt = C.ls->desc[lidx].len == 0 ? 0 :
			                al_local_search(C, ws, lidx, ord, hitoff, &hitlen, &top, &bot, &uniqueStop, 0xffffu);
			if(nelt > 0 && nelt <= P.kseeds && hitlen > max_hitlen) {
				h2g_coord co[AL_MAX_GHITS];
				uint32_t nco = 0;
				al_local_coords(C, ws, lidx, top, bot, hitoff - hitlen + 1, hitlen, co, AL_MAX_GHITS, &nco);
				ws->nghits = 0;
				for(uint32_t ri = 0; ri < nco; ri++) {
					if(P.no_spliced) {
						if((uint64_t)co[ri].toff + (uint64_t)P.maxFragLen * 2 < toff || (uint64_t)toff + (uint64_t)P.maxFragLen * 2 < co[ri].toff) continue;
					}
					if(C.graph) {                            // adjustWithALT (:5692)
						uint32_t ovf = 0;
						adjust_with_alt(*C.g, *C.ref, *C.alts, ord, hitoff - hitlen + 1, hitlen, co[ri].tidx, co[ri].toff, co[ri].joinedOff, ws->ghits,
						                &ws->nghits, AL_MAX_GHITS, &C.gws->awa, &ovf);
						if(ovf) ws->overflow |= 64;
					} else if(ws->nghits < AL_MAX_GHITS) hit_init(&ws->ghits[ws->nghits++], ord.fw, hitoff - hitlen + 1, hitlen, co[ri].tidx, co[ri].toff, co[ri].joinedOff);
					else ws->overflow |= 64;
				}
				max_hitlen = hitlen;
			}
			if(hitlen > 0) hitoff -= (hitlen - 1);
			if(hitoff > 0) hitoff -= 1;
		}
	}
	// (genomeHits never exceeds kseeds here: nelt <= kseeds)
	for(uint32_t hi = 0; hi < ws->nghits; hi++) {
		uint32_t le = H2G_MAX, re = H2G_MAX;
		al_extend(C, ord, &ws->ghits[hi], 0, H2G_MAX, H2G_MAX, &le, &re);
		hit_copy(&ws->tmp2, &ws->ghits[hi]);
		al_hybrid_search_recur(C, ord, ws, omw, &ws->tmp2, ws->tmp2.rdoff, ws->tmp2.len, omw->minsc, true);
	}
	(void)rnd;
}

// hi_aligner.h:4048 (go) / :4644 (nextBWT) / :4868 (pickNextReadToSearch); rds[1] == nullptr for an unpaired read
H2G_HD void al_go(const AlnCtx& C, const DReads* const rds[2], uint32_t read, AlignWS* ws, Rng* rndp, int slot0 = 0)
{
	const AlnParams& P = *C.P;
	const bool paired = rds[1] != nullptr;
	const int nm = paired ? 2 : 1;
	const uint32_t minK = C.g->minK;
	Rng& rnd = *rndp;
	ws->nghits = 0; ws->overflow = 0; ws->nrank = 0; ws->nside = 0; ws->nsteps = 0; ws->nframes_max = 0;
	ws->npairs = 0; ws->insp_i = 0; ws->insp_j = 0; ws->bestPair = INT64_MIN; ws->best2Pair = INT64_MIN;
	ws->localindexatts = 0; ws->max_localindexatts = 0;
	uint32_t rdlens[2] = {0, 0};
	for(int r = 0; r < 2; r++) {
		MateWS& mw = ws->m[r ^ slot0];
		mw.nsearched = 0; mw.nres = 0; mw.bestUnp = INT64_MIN; mw.best2Unp = INT64_MIN; mw.minsc = INT64_MAX;
		mw.sink_hidden = (slot0 != 0 && nm == 1) ? 1u : 0u;
		if(r < nm) {
			SeqView v = seq_view(*rds[r], read, true);
			rdlens[r] = v.len;
			// scoreMin.f<TAlScore>(len), SIMPLE_FUNC_LINEAR 0, -0.2 (hisat2.cpp:440, simple_func.h:88)
			mw.minsc = min_score_for(P, v.len);
			for(int k = 0; k < 2; k++) {
				RBHit& h = mw.rb[k];
				h.len = v.len; h.cur = 0; h.done = 0; h.numPartialSearch = 0; h.numUniqueSearch = 0; h.npartial = 0;
			}
		}
	}
	bool found[2][2] = {{true, true}, {paired, paired}};
	while(true) {
		// ---------------- nextBWT ----------------
		int sel_r = -1, sel_f = -1;
		while(true) {
			int rdi = -1, fwi = -1;
			int64_t maxScore = INT64_MIN;
			for(int r = 0; r < nm; r++) for(int k = 0; k < 2; k++) {
				const RBHit& h = ws->m[r ^ slot0].rb[k];
				if(h.done) continue;
				int64_t cs = rb_search_score(h, minK);
				if(h.cur == 0) cs = INT64_MAX;
				if(cs > maxScore) { maxScore = cs; rdi = r; fwi = k; }
			}
			if(rdi < 0) break;
			MateWS& mw = ws->m[rdi ^ slot0];
			MateWS& ow = ws->m[(1 - rdi) ^ slot0];
			RBHit& hit = mw.rb[fwi];
			RBHit& rchit = mw.rb[1 - fwi];
			bool ret_false = false, cont = false;
			if(!P.secondary) {
				const uint32_t numSearched = hit.numPartialSearch - hit.numUniqueSearch;
				const int64_t bestScore = mw.bestUnp;
				if(bestScore >= mw.minsc) {
					const uint32_t maxmm = (uint32_t)((-bestScore + P.sc.mmpMax - 1) / P.sc.mmpMax);
					if(numSearched > maxmm + 0 + 1) {
						hit.done = 1;
						if(paired) { if(ow.bestUnp >= ow.minsc && ws->npairs > 0) ret_false = true; else cont = true; }
						else ret_false = true;
					}
				}
				if(!ret_false && !cont && rchit.done && bestScore < mw.minsc) {
					if(numSearched > (rchit.numPartialSearch - rchit.numUniqueSearch) + (P.anchorStop ? 1u : 0u)) { hit.done = 1; ret_false = true; }
				}
			}
			if(ret_false) break;
			if(cont) continue;
			SeqView sv = seq_view(*rds[rdi], read, fwi == 0);
			h2g_fm_hit fh;
			if(!C.graph) partial_search_item(*C.g, sv, hit.cur, P.pseudogeneStop != 0, P.anchorStop != 0, P.khits, &fh);
			else {
				partial_search_graph_item(*C.g, sv, hit.cur, P.pseudogeneStop != 0, P.anchorStop != 0, P.khits, P.kseeds, &fh, &C.gws->ie);
				if(hit.npartial < AL_MAX_PARTIAL) {
					GraphPNode& pn = C.gws->pnode[(int)(&mw - ws->m)][fwi][hit.npartial];
					pn.node_top = fh.node_top; pn.node_bot = fh.node_bot; pn.ie = C.gws->ie;
					if(pn.ie.n > H2G_IEDGE_CAP) ws->overflow |= 512;
				}
			}
			AL_TRACE("  psearch rdi %d fwi %d cur %u -> top %u bot %u len %u type %u cur %u done %u anchor %u\n", rdi, fwi, hit.cur, fh.top, fh.bot, fh.len, fh.hit_type, fh.cur, fh.done, fh.anchorStop);
			ws->nrank += fh.nrank; ws->nside += fh.nside;
			hit.numPartialSearch += 1; hit.numUniqueSearch += fh.numUniqueSearch; hit.cur = fh.cur;
			if(hit.npartial < AL_MAX_PARTIAL) {
				PartialHit& p = hit.partial[hit.npartial++];
				p.top = fh.top; p.bot = fh.bot; p.bwoff = fh.bwoff; p.len = fh.len; p.hit_type = fh.hit_type; p.ncoords = 0;
			} else { ws->overflow |= 32; hit.done = 1; break; }
			if(fh.done) { hit.done = 1; sel_r = rdi; sel_f = fwi; break; }
			if(!fh.pseudogeneStop) { if(hit.cur + 1 < hit.len) hit.cur++; }
			if(fh.anchorStop) { hit.done = 1; sel_r = rdi; sel_f = fwi; break; }
		}
		if(sel_r < 0) break;
		SeqView sv = seq_view(*rds[sel_r], read, sel_f == 0);
		found[sel_r][sel_f] = al_align(C, sv, ws, &ws->m[sel_r ^ slot0], sel_f, &rnd);
		AL_TRACE(" align rdi %d fwi %d -> found %d nghits %u\n", sel_r, sel_f, (int)found[sel_r][sel_f], ws->nghits);
		if(!found[0][0] && !found[0][1] && !found[1][0] && !found[1][1]) break;
		if(paired) al_pair_reads(P, ws, rdlens[0], rdlens[1]);
	}
	// no concordant pair: use each mate's alignments as anchors for the other mate (hi_aligner.h:4092-4148)
	if(paired && ws->npairs == 0 && (ws->m[0].bestUnp >= ws->m[0].minsc || ws->m[1].bestUnp >= ws->m[1].minsc)) {
		bool mate_found = false;
		const uint32_t rs_size[2] = {ws->m[0].nres, ws->m[1].nres};
		for(int i = 0; i < 2; i++) {
			for(uint32_t j = 0; j < rs_size[i]; j++) {
				const bool fw = ws->m[i].res[j].fw != 0;
				const uint32_t tidx = ws->m[i].res[j].tidx, toff = ws->m[i].res[j].toff;
				SeqView ord = seq_view(*rds[1 - i], read, !fw);   // ofw = (fw == gMate2fw ? gMate1fw : gMate2fw) = !fw
				AL_TRACE(" alignMate anchor mate %d res %u fw %d toff %u\n", i, j, (int)fw, toff);
				al_align_mate(C, ord, ws, &ws->m[1 - i], fw, tidx, toff, &rnd);
				mate_found = true;
			}
		}
		if(mate_found) al_pair_reads(P, ws, rdlens[0], rdlens[1]);
	}
}

// ---------------------------------------------------------------------------------------- finishRead selection (N1)
// AlnSinkWrap::finishRead (aln_sink.h:1939) unpaired branch -> selectByScore (aln_sink.h:2680-2760), -k mode
// (mhits unset): `select` lists the alignments to print, best first; select[0] is the primary.
H2G_HD int64_t hisat2_score(const AlnRec& r) {   // AlnScore::calculate_hisat2_score aligner_result.h:322
	int64_t score = r.score;
	if(score > INT32_MAX) score = INT32_MAX; else if(score < INT32_MIN) score = INT32_MIN;
	int64_t trim = (int64_t)r.trim5 + r.trim3;
	trim = trim > 0xffff ? 0 : 0xffff - trim;
	return (int64_t)((uint64_t)score << 32) | (0ll << 28) | (0ll << 24) | (255ll << 16) | trim;
}

H2G_HD uint32_t al_select(const MateWS* ws, const AlnParams& P, Rng* rnd, uint8_t* select) {
	const uint32_t sz = ws->nres;
	if(sz < 1) return 0;
	uint32_t num = P.khits < sz ? P.khits : sz;
	int64_t key[AL_MAX_RESULTS];
	uint32_t idx[AL_MAX_RESULTS];
	for(uint32_t i = 0; i < sz; i++) { key[i] = hisat2_score(ws->res[i]); idx[i] = i; }
	// buf.sort(); buf.reverse(): descending by (score, original offset)
	for(uint32_t i = 1; i < sz; i++) {
		int64_t k = key[i]; uint32_t x = idx[i];
		int j = (int)i - 1;
		while(j >= 0 && (key[j] < k || (key[j] == k && idx[j] < x))) { key[j + 1] = key[j]; idx[j + 1] = idx[j]; j--; }
		key[j + 1] = k; idx[j + 1] = x;
	}
	// randomise streaks of equal score (shufflePortion ds.h:836)
	uint32_t streak = 0;
	for(uint32_t i = 1; i <= sz; i++) {
		if(i < sz && key[i] == key[i - 1]) { if(streak == 0) streak = 1; streak++; }
		else {
			if(streak > 1) {
				const uint32_t begin = i - streak;
				uint32_t left = streak;
				for(uint32_t q = begin; q + 1 < begin + streak; q++) {
					uint32_t r = rnd->nextU32() % left;
					if(r > 0) { int64_t tk = key[q]; key[q] = key[q + r]; key[q + r] = tk; uint32_t ti = idx[q]; idx[q] = idx[q + r]; idx[q + r] = ti; }
					left--;
				}
			}
			streak = 0;
		}
	}
	uint32_t nsel = 0;
	for(uint32_t i = 0; i < sz; i++) { if(i >= num) break; select[nsel++] = (uint8_t)idx[i]; }
	if(!P.secondary) {
		for(uint32_t i = 0; i + 1 < nsel; i++) if(key[i] != key[i + 1]) { nsel = i + 1; break; }
	}
	return nsel;
}

// Whole per-read pipeline of the worker loop body (hisat2.cpp:3380-3640) for an unpaired read that passed
// the filters: seed the PRNG, go(), select.
struct ReadOut {
	uint32_t nres, nselect, overflow, nrank, nsteps, depth, nside;
	// AlnSetSumm::init (aligner_result.cpp:1209-1234) over ALL reported alignments, not only the selected ones: best and
	// second-best AlnScore (score, then fewer soft-trimmed bases) — the inputs of MAPQ and ZS:i.  INT32_MIN = invalid.
	int32_t  best, secbest;
	uint32_t best_trim, secbest_trim;
	uint8_t  select[AL_MAX_RESULTS];
};

// Scoring::nFilter scoring.cpp:104 with the effective default nCeil = L,0,0.15 (SeedAlignmentPolicy::parseString
// aligner_seed_policy.cpp:294-296 overrides hisat2.cpp:443) + the length filter (hisat2.cpp:3413)
H2G_HD bool read_passes_filters(const SeqView& v) {
	if(v.len < 2) return false;
	const uint32_t maxns = (uint32_t)(0.0 + (double)0.15f * (double)v.len);
	uint32_t ns = 0;
	for(uint32_t i = 0; i < v.len; i++) if(v.fwc[i] == 4) { if(++ns > maxns) return false; }
	return true;
}

H2G_HD void al_read(const AlnCtx& C, const DReads& rd, uint32_t read, const char* name, uint32_t namelen, AlignWS* ws, ReadOut* out) {
	SeqView fwv = seq_view(rd, read, true);
	Rng rnd;
	rnd.init(gen_rand_seed(fwv, name, namelen, 0));                // rnd.init(ps->bufa().seed) hisat2.cpp:3468
	const DReads* rds[2] = {&rd, nullptr};
	if(!read_passes_filters(fwv)) {                                // filt[0] false: go() is skipped (hisat2.cpp:3518)
		ws->m[0].nres = 0; ws->overflow = 0; ws->nrank = 0; ws->nsteps = 0; ws->nframes_max = 0; ws->nside = 0;
	} else
	al_go(C, rds, read, ws, &rnd);
	out->nres = ws->m[0].nres; out->overflow = ws->overflow; out->nrank = ws->nrank; out->nsteps = ws->nsteps; out->depth = ws->nframes_max; out->nside = ws->nside;
	out->nselect = al_select(&ws->m[0], *C.P, &rnd, out->select);
	int64_t b = INT64_MIN, sb = INT64_MIN;
	uint32_t bt = 0, sbt = 0;
	for(uint32_t i = 0; i < ws->m[0].nres; i++) {
		const AlnRec& r = ws->m[0].res[i];
		const uint32_t t = r.trim5 + r.trim3;
		if(b == INT64_MIN || r.score > b || (r.score == b && t < bt)) { sb = b; sbt = bt; b = r.score; bt = t; }
		else if(sb == INT64_MIN || r.score > sb || (r.score == sb && t < sbt)) { sb = r.score; sbt = t; }
	}
	out->best = b == INT64_MIN ? INT32_MIN : (int32_t)b; out->secbest = sb == INT64_MIN ? INT32_MIN : (int32_t)sb;
	out->best_trim = bt; out->secbest_trim = sbt;
}

// Paired read: rnd.init(seedA ^ seedB) (hisat2.cpp:3464-3466), go() with both mates.  The concordant /
// discordant / unpaired classification and selection of finishRead stay on the host: the
// caller replays the returned report events into its sink and continues the PRNG from `rnd_state`.
struct PairOut {
	uint32_t nres[2], npairs, overflow, nrank, nsteps, depth, nside, rnd_state, pad;
	uint8_t  pair_i[AL_MAX_PAIRS], pair_j[AL_MAX_PAIRS];
};

H2G_HD void al_pair(const AlnCtx& C, const DReads& rd1, const DReads& rd2, uint32_t read, const char* name1, uint32_t namelen1,
                    const char* name2, uint32_t namelen2, AlignWS* ws, PairOut* out) {
	SeqView v1 = seq_view(rd1, read, true), v2 = seq_view(rd2, read, true);
	Rng rnd;
	const bool f1 = read_passes_filters(v1), f2 = read_passes_filters(v2);
	const uint32_t s1 = gen_rand_seed(v1, name1, namelen1, 0), s2 = gen_rand_seed(v2, name2, namelen2, 0);
	rnd.init((f1 && f2) ? (s1 ^ s2) : s1);                       // hisat2.cpp:3463-3468
	ws->m[0].nres = 0; ws->m[1].nres = 0; ws->npairs = 0; ws->overflow = 0; ws->nrank = 0; ws->nsteps = 0; ws->nframes_max = 0; ws->nside = 0;
	if(f1 && f2) { const DReads* rds[2] = {&rd1, &rd2}; al_go(C, rds, read, ws, &rnd); }
	else if(f1)  { const DReads* rds[2] = {&rd1, nullptr}; al_go(C, rds, read, ws, &rnd, 0); }     // initRead(rds[0]) hisat2.cpp:3522
	else if(f2)  { const DReads* rds[2] = {&rd2, nullptr}; al_go(C, rds, read, ws, &rnd, 1); }     // initRead(rds[1], rightendonly) :3524
	out->nres[0] = ws->m[0].nres; out->nres[1] = ws->m[1].nres; out->npairs = ws->npairs; out->overflow = ws->overflow;
	out->nrank = ws->nrank; out->nsteps = ws->nsteps; out->depth = ws->nframes_max; out->nside = ws->nside; out->rnd_state = rnd.last; out->pad = 0;
	for(uint32_t k = 0; k < ws->npairs; k++) { out->pair_i[k] = ws->pair_i[k]; out->pair_j[k] = ws->pair_j[k]; }
}

}  // namespace h2g
