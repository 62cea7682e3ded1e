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
#ifndef AL_MAX_SEARCHED        // the *_big units raise these (second pass over overflowed reads, option sets beyond the defaults)
#define AL_MAX_SEARCHED 32     // the default workspace is sized for the common read: ~100 KB per read in flight; whatever
#define AL_MAX_RESULTS  16     // overflows it is re-run with the *_big capacities (h2g_go_big.h) by the second pass
#define AL_MAX_DEPTH    20
#define AL_MAX_LOCALHITS 4
#define AL_MAX_COORDS   12
#define AL_MAX_PARTIAL  16
#endif
#define H2G_SELECT_CAP 32      // alignments selected per read: >= the largest -k (30: --very-sensitive); fixed in every unit

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
	uint32_t sides_bytes;    // size of its side array (what a workgroup stages in LDS, h2g_ext_search)
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
H2G_HD bool local_is_linear(const DLocalDesc& d) { return d.len + 1 == d.gbwtLen || d.gbwtLen == 0; }   // GFMParams::linearFM gfm.h
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
	const uint8_t*  sbase = nullptr;   // the index's sides / ftab staged in LDS (h2g_ext_search); nullptr: read them from HBM
	const uint16_t* fbase = nullptr;
	H2G_HD const uint8_t* side_ptr(uint32_t sideNum) const { return (sbase ? sbase : ls->sides + d->sides_off) + (size_t)sideNum * 64; }
	H2G_HD uint32_t ftab_at(uint32_t i) const { return fbase ? fbase[i] : ls->words[d->ftab_off + i]; }
	H2G_HD uint32_t ftabChars() const { return ls->ftabChars; }
	H2G_HD bool is_zoff(uint32_t row) const { return d->nZ && row == d->zoff; }
	H2G_HD uint32_t side_of(uint32_t row) const { return row / 224u; }
	H2G_HD uint32_t fh(uint32_t i) const {   // ftabHi gfm.h:2618 with 16-bit words
		uint32_t v = ftab_at(i);
		if(v <= d->ftabLim) return v;
		return ls->words[d->eftab_off + ((v ^ 0xffffu) * 2 + 1)];
	}
	H2G_HD uint32_t fl(uint32_t i) const {
		uint32_t v = ftab_at(i);
		if(v <= d->ftabLim) return v;
		return ls->words[d->eftab_off + ((v ^ 0xffffu) * 2)];
	}
	H2G_HD void lohi(uint32_t fi, uint32_t* top, uint32_t* bot) const { *top = fh(fi); *bot = fl(fi + 1); }
	H2G_HD uint32_t rank(uint32_t row, int c) const {   // countBt2Side gfm.h:2958 for index_t = uint16_t
		uint32_t sideNum = row / 224u, charOff = row - sideNum * 224u;
		Side64 s = load_side64(side_ptr(sideNum));
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
		const uint8_t* p = side_ptr(sideNum);
		return (p[charOff >> 2] >> ((charOff & 3) * 2)) & 3;
	}
	H2G_HD uint32_t lf(uint32_t row) const { return rank(row, rowL(row)); }      // one step of an SA walk: mapLF(row, rowL(row))
};
// The same local index with its descriptor IN REGISTERS (round 6: the fast pass's local searches and walks).  LIdx reads DLocalDesc fields from HBM at every rank — the side
// array's offset, the '$' row, fchr[c] indexed by a character that is only known once the row's own symbol has come back — and an SA-walk step is a byte load (rowL) followed
// by the side load (rank).  Here the descriptor is read once per primitive and a walk step is ONE side load: the symbol is taken from the side in registers.  Same arithmetic.
struct LIdxR {
	const uint8_t*  sides;
	const uint16_t* ftab; const uint16_t* eftab;
	uint32_t nZ, zoff, ftabLim, fchr0, fchr1, fchr2, fchr3, ftab_chars;
	H2G_HD void init(const DLocalSet* ls, const DLocalDesc* d) {
		sides = ls->sides + d->sides_off; ftab = ls->words + d->ftab_off; eftab = ls->words + d->eftab_off;
		nZ = d->nZ; zoff = d->zoff; ftabLim = d->ftabLim; fchr0 = d->fchr[0]; fchr1 = d->fchr[1]; fchr2 = d->fchr[2]; fchr3 = d->fchr[3]; ftab_chars = ls->ftabChars;
	}
	H2G_HD uint32_t ftabChars() const { return ftab_chars; }
	H2G_HD bool is_zoff(uint32_t row) const { return nZ && row == zoff; }
	H2G_HD uint32_t side_of(uint32_t row) const { return row / 224u; }
	H2G_HD uint32_t fh(uint32_t i) const { const uint32_t v = ftab[i]; return v <= ftabLim ? v : eftab[(v ^ 0xffffu) * 2 + 1]; }
	H2G_HD uint32_t fl(uint32_t i) const { const uint32_t v = ftab[i]; return v <= ftabLim ? v : eftab[(v ^ 0xffffu) * 2]; }
	H2G_HD void lohi(uint32_t fi, uint32_t* top, uint32_t* bot) const { *top = fh(fi); *bot = fl(fi + 1); }
	H2G_HD uint32_t fchr_of(int c) const { return c == 0 ? fchr0 : c == 1 ? fchr1 : c == 2 ? fchr2 : fchr3; }
	H2G_HD uint32_t rank_in(const Side64& s, uint32_t sideNum, uint32_t charOff, int c) const {
		uint32_t cnt = 0;
#pragma unroll
		for(int k = 0; k < 7; k++) cnt += count_word(s.w[k], c, (int)charOff - 32 * k);
		if(c == 0 && nZ) {
			const uint32_t zs = zoff / 224u, zc = zoff - zs * 224u;
			if(zs == sideNum && zc < charOff) cnt--;
		}
		return (uint32_t)((s.w[7] >> (16 * c)) & 0xffffu) + cnt + fchr_of(c);
	}
	H2G_HD uint32_t rank(uint32_t row, int c) const {
		const uint32_t sideNum = row / 224u, charOff = row - sideNum * 224u;
		return rank_in(load_side64(sides + (size_t)sideNum * 64), sideNum, charOff, c);
	}
	H2G_HD int rowL(uint32_t row) const {
		const uint32_t sideNum = row / 224u, charOff = row - sideNum * 224u;
		return (sides[(size_t)sideNum * 64 + (charOff >> 2)] >> ((charOff & 3) * 2)) & 3;
	}
	H2G_HD uint32_t lf(uint32_t row) const {
		const uint32_t sideNum = row / 224u, charOff = row - sideNum * 224u;
		const Side64 s = load_side64(sides + (size_t)sideNum * 64);
		// (224 symbols: words 0 .. 6, picked by selects — an index into the array would put the side into private memory)
		const uint32_t k = charOff >> 5;
		const uint64_t a = (k & 1) ? s.w[1] : s.w[0], b = (k & 1) ? s.w[3] : s.w[2], d = (k & 1) ? s.w[5] : s.w[4];
		const uint64_t w = k >= 6 ? s.w[6] : (k >= 4 ? d : (k >= 2 ? b : a));
		const int c = (int)((w >> ((charOff & 31u) * 2u)) & 3u);
		return rank_in(s, sideNum, charOff, c);
	}
};

// A local index without a variant in its interval is a LINEAR index even inside a graph index (GFMParams::linearFM gfm.h:149:
// len + 1 == gbwtLen): it keeps the file's 128 B sides (lineRate 7) but lays them out linearly — 120 B payload (480 symbols)
// + u16 occ[4] (GFMParams::init: sideGbwtSz = sideSz - 4 * sizeof(index_t)).
struct LIdxW {
	const DLocalSet* ls;
	const DLocalDesc* d;
	H2G_HD const uint8_t* side_ptr(uint32_t sideNum) const { return ls->sides + d->sides_off + (size_t)sideNum * 128; }
	H2G_HD uint32_t ftab_at(uint32_t i) const { return ls->words[d->ftab_off + i]; }
	H2G_HD uint32_t ftabChars() const { return ls->ftabChars; }
	H2G_HD bool is_zoff(uint32_t row) const { return d->nZ && row == d->zoff; }
	H2G_HD uint32_t side_of(uint32_t row) const { return row / 480u; }
	H2G_HD uint32_t fh(uint32_t i) const { const uint32_t v = ftab_at(i); return v <= d->ftabLim ? v : ls->words[d->eftab_off + ((v ^ 0xffffu) * 2 + 1)]; }
	H2G_HD uint32_t fl(uint32_t i) const { const uint32_t v = ftab_at(i); return v <= d->ftabLim ? v : ls->words[d->eftab_off + ((v ^ 0xffffu) * 2)]; }
	H2G_HD void lohi(uint32_t fi, uint32_t* top, uint32_t* bot) const { *top = fh(fi); *bot = fl(fi + 1); }
	H2G_HD uint32_t rank(uint32_t row, int c) const {   // countBt2Side gfm.h:2958 for index_t = uint16_t on a 128 B side
		const uint32_t sideNum = row / 480u, charOff = row - sideNum * 480u;
		const uint8_t* p = side_ptr(sideNum);
		uint32_t cnt = 0;
		for(int k = 0; k < 15; k++) { uint64_t w; memcpy(&w, p + 8 * k, 8); cnt += count_word(w, c, (int)charOff - 32 * k); }
		if(c == 0 && d->nZ) {
			const uint32_t zs = d->zoff / 480u, zc = d->zoff - zs * 480u;
			if(zs == sideNum && zc < charOff) cnt--;
		}
		uint16_t occ; memcpy(&occ, p + 120 + 2 * c, 2);
		return (uint32_t)occ + cnt + d->fchr[c];
	}
	H2G_HD int rowL(uint32_t row) const {
		const uint32_t sideNum = row / 480u, charOff = row - sideNum * 480u;
		const uint8_t* p = side_ptr(sideNum);
		return (p[charOff >> 2] >> ((charOff & 3) * 2)) & 3;
	}
	H2G_HD uint32_t lf(uint32_t row) const { return rank(row, rowL(row)); }
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
		if(++jumps > (1u << 22)) break;              // never reached on a well-formed index (the text is shorter): no endless walk on the device
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
template <typename LX>
H2G_HD bool genome_coords_local(const LX& ix, uint32_t top, uint32_t bot, uint32_t rdoff, uint32_t rdlen, h2g_coord* coords,
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
	h->score = 0; h->nedits = 0; h->overflow = 0; h->splicescore = 0;
}
H2G_HD void hit_copy(h2g_ghit* d, const h2g_ghit* s) {
	d->read = s->read; d->fw = s->fw; d->rdoff = s->rdoff; d->len = s->len; d->trim5 = s->trim5; d->trim3 = s->trim3;
	d->tidx = s->tidx; d->toff = s->toff; d->joinedOff = s->joinedOff; d->score = s->score; d->nedits = s->nedits;
	d->overflow = s->overflow; d->splicescore = s->splicescore;
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
	if(h->nedits >= H2G_GHIT_EDITS) { h->overflow = 1; return; }
	h2g_edit& e = h->edits[h->nedits++];
	e.pos = pos; e.chr = chr; e.qchr = qchr; e.type = type; e.pad = 0; e.snp = H2G_MAX;
}

// combineWith hi_aligner.h:1420-2025 for linear indexes without spliced alignment: plain concatenation
// (:1506-1525) or one insertion / deletion placed by the prefix/suffix score scan (:1741-1794).
#define H2G_COMBINE_SCRATCH 512
// combineWith's temp_scores of one lane: element i of the lanes of a wave are adjacent (stride = lanes sharing the block), so a
// scan in lockstep touches a handful of lines per step instead of one line per lane
struct ScVec {
	int64_t* p; uint32_t stride;
	H2G_HD int64_t& operator[](uint32_t i) const { return p[(size_t)i * stride]; }
};
H2G_HD bool hit_combine(const DRef& ref, const DScoring& sc_in, const SeqView& seq, h2g_ghit* a, const h2g_ghit* b, int64_t minsc,
                        uint32_t minIntronLen, bool no_spliced, ScVec tmp1, ScVec tmp2, const DAlts* alts = nullptr,
                        const h2g_coord* site = nullptr /* a splice site of the database: {left, right, dir} (:1439) */,
                        uint32_t minAnchorLen = 0, uint32_t minAnchorLen_noncan = 0 /* 0: the policy's (tp.h); the database joins pass 1, 1 */)
{
	if(a == b) return false;
	DScoring sc_anchor;                                  // calculateScore sees the caller's anchor minima (:1513-1522)
	const bool anchor_override = minAnchorLen != 0;
	if(anchor_override) { sc_anchor = sc_in; sc_anchor.minAnchorLen = minAnchorLen; sc_anchor.minAnchorLen_noncan = minAnchorLen_noncan; }
	const DScoring& sc = anchor_override ? sc_anchor : sc_in;
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
	if(spliced && (sc.donor_sum == nullptr || refdif - rddif > H2G_SPL_MAXLEN)) return false;   // probscore tables are uploaded with spliced alignment on
	if(!spliced && !ins && !del && this_rdoff + this_len == other_rdoff) {
		const uint32_t addoff = b->rdoff - a->rdoff;
		for(uint32_t i = 0; i < b->nedits; i++) {
			hit_push_edit(a, b->edits[i].pos + addoff, b->edits[i].chr, b->edits[i].qchr, b->edits[i].type);
			if(!a->overflow) { a->edits[a->nedits - 1].snp = b->edits[i].snp; a->edits[a->nedits - 1].pad = b->edits[i].pad; }   // whole Edit (ALT id, splice fields)
		}
		a->len += b->len;
		calculate_score(sc, seq, a);
		return true;
	}
	const uint32_t rdlen = seq.len;
	int64_t remainsc = minsc - (a->score - this_score) - (b->score - other_score);
	if(remainsc > 0) remainsc = 0;
	// sc.maxReadGaps(remainsc + sc.canSpl(), rdlen) (:1539-1540; canSpl() = --pen-cansplice, 0 by default); 0 when an intron is placed
	const int read_gaps = spliced ? 0 : max_gaps(remainsc + sc.cp, sc.rdGapConst + sc.rdGapLinear, sc.rdGapLinear);
	const int ref_gaps = spliced ? 0 : max_gaps(remainsc + sc.cp, sc.rfGapConst + sc.rfGapLinear, sc.rfGapLinear);
	(void)rdlen;
	if(ins) { if(refdif + ref_gaps < rddif) return false; }
	else if(del) { if(rddif + read_gaps < refdif) return false; }
	int this_ref_ext = read_gaps;
	if(spliced) this_ref_ext += H2G_SPL_INTRONIC;
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
	uint32_t maxspldir = H2G_SPL_UNKNOWN;
	float maxsplscore = 0.0f, spl_probstore = 0.0f;
	if(spliced) {
		// ---- discover the splice site (:1588-1739): prefix / suffix mismatch scores, donor / acceptor dinucleotides, PWM tie-break
		if(len > H2G_COMBINE_SCRATCH) { a->overflow = 1; return false; }
		int64_t other_ref_ext = (int64_t)read_gaps + H2G_SPL_INTRONIC;
		{ const int64_t lim = (int64_t)other_toff + other_len - len; if(lim < other_ref_ext) other_ref_ext = lim; }
		int i;
		for(i = 0; i < (int)len; i++) {
			int rdc = seq.at(this_rdoff + i), rfc = rc1.get((int64_t)this_toff + i);
			tmp1[i] = i > 0 ? tmp1[i - 1] : 0;
			if(rdc != rfc) tmp1[i] += score_cell(sc, rdc, rfc, seq.qual(this_rdoff + i) - 33);
			if(tmp1[i] < remainsc) break;
		}
		int i_limit = i < (int)len ? i : (int)len;
		int i2;
		for(i2 = (int)len - 1; i2 >= 0; i2--) {
			int64_t p = base2 + i2;
			int rdc = seq.at(this_rdoff + i2), rfc = p < 0 ? 4 : rc2.get(p);
			tmp2[i2] = (uint32_t)(i2 + 1) < len ? tmp2[i2 + 1] : 0;
			if(rdc != rfc) tmp2[i2] += score_cell(sc, rdc, rfc, seq.qual(this_rdoff + i2) - 33);
			if(tmp2[i2] < remainsc) break;
		}
		int i2_limit = i2 > 0 ? i2 : 0;
		if(site) {                                       // a database site fixes the junction (:1626-1634)
			const int at = (int)(site->tidx - this_toff);
			if(i2_limit <= at) { i2_limit = at; i_limit = i2_limit + 1; }
			else i_limit = i2_limit;
		}
		const int GT = 0x23, AG = 0x02, GTrc = 0x01, AGrc = 0x13, GC = 0x21, GCrc = 0x21, AT = 0x03, AC = 0x01, ATrc = 0x03, ACrc = 0x20;
		auto rf1 = [&](int j) -> int { return rc1.get((int64_t)this_toff + j); };                     // refbuf[j]
		auto rf2 = [&](int j) -> int { const int64_t p = base2 + j; return p < 0 ? 4 : rc2.get(p); }; // refbuf2[j], j >= -other_ref_ext
		for(i = i2_limit, i2 = i2_limit + 1; i < i_limit && i2 < (int)len; i++, i2++) {
			int64_t tempscore = tmp1[i] + tmp2[i2];
			int donor = 0xff, acceptor = 0xff;
			if((uint32_t)(i + 2) < len + (uint32_t)this_ref_ext) donor = ((rf1(i + 1) << 4) | rf1(i + 2)) & 0xff;
			if((int64_t)i2 - 2 >= -other_ref_ext) acceptor = ((rf2(i2 - 2) << 4) | rf2(i2 - 1)) & 0xff;
			bool canonical = false, semi = false;
			uint32_t spldir = H2G_SPL_UNKNOWN;
			if(donor == GT && acceptor == AG) { spldir = H2G_SPL_FW; canonical = true; }
			else if(donor == AGrc && acceptor == GTrc) { spldir = H2G_SPL_RC; canonical = true; }
			else if((donor == GC && acceptor == AG) || (donor == AT && acceptor == AC)) { spldir = H2G_SPL_SEMI_FW; semi = true; }
			else if((donor == AGrc && acceptor == GCrc) || (donor == ACrc && acceptor == ATrc)) { spldir = H2G_SPL_SEMI_RC; semi = true; }
			tempscore -= (canonical ? sc.cp : sc.ncp);
			int64_t dseq = 0, aseq = 0;
			float splscore = 0.0f;
			if(canonical) {
				const int LT = (int)len + this_ref_ext, ORE = (int)other_ref_ext;
				if(spldir == H2G_SPL_FW) {
					if(i + 1 >= H2G_SPL_DONOR_EXONIC && LT > i + H2G_SPL_DONOR_INTRONIC && i2 + ORE >= H2G_SPL_ACC_INTRONIC && (int)len > i2 + H2G_SPL_ACC_EXONIC - 1) {
						for(int j = i + 1 - H2G_SPL_DONOR_EXONIC; j <= i + H2G_SPL_DONOR_INTRONIC; j++) { int b = rf1(j); if(b > 3) b = 0; dseq = dseq << 2 | b; }
						for(int j = i2 - H2G_SPL_ACC_INTRONIC; j <= i2 + H2G_SPL_ACC_EXONIC - 1; j++) { int b = rf2(j); if(b > 3) b = 0; aseq = aseq << 2 | b; }
					}
				} else {
					if(i + 1 >= H2G_SPL_ACC_EXONIC && LT > i + H2G_SPL_ACC_INTRONIC && i2 + ORE >= H2G_SPL_DONOR_INTRONIC && (int)len > i2 + H2G_SPL_DONOR_EXONIC - 1) {
						for(int j = i + H2G_SPL_ACC_INTRONIC; j >= i + 1 - H2G_SPL_ACC_EXONIC; j--) { int b = rf1(j); if(b > 3) b = 0; aseq = aseq << 2 | (b ^ 3); }
						for(int j = i2 + H2G_SPL_DONOR_EXONIC - 1; j >= i2 - H2G_SPL_DONOR_INTRONIC; j--) { int b = rf2(j); if(b > 3) b = 0; dseq = dseq << 2 | (b ^ 3); }
					}
				}
				splscore = spl_probscore(sc, dseq, aseq);
			}
			if((maxspldir == H2G_SPL_UNKNOWN && spldir == H2G_SPL_UNKNOWN && maxscore < tempscore) ||
			   (maxspldir == H2G_SPL_UNKNOWN && spldir == H2G_SPL_UNKNOWN && maxscore == tempscore && semi) ||
			   (maxspldir != H2G_SPL_UNKNOWN && spldir != H2G_SPL_UNKNOWN && (maxscore < tempscore || (maxscore == tempscore && maxsplscore < splscore))) ||
			   (maxspldir == H2G_SPL_UNKNOWN && spldir != H2G_SPL_UNKNOWN)) {
				maxscore = tempscore; maxscorei = (uint32_t)i; maxspldir = spldir; maxsplscore = splscore;
				spl_probstore = splscore;   // = probscore(donor_seq, acceptor_seq) of the Edit; only read back for canonical sites (:3774)
			}
		}
		if(maxscore == INT64_MIN) return false;
		if(!site) {   // anchor-length / intron-length veto for a novel site (:1797-1813)
			const uint32_t shorter = maxscorei + 1 < len - maxscorei - 1 ? maxscorei + 1 : len - maxscorei - 1;
			const bool noncan = maxspldir == H2G_SPL_SEMI_FW || maxspldir == H2G_SPL_SEMI_RC || maxspldir == H2G_SPL_UNKNOWN;
			if(shorter < (noncan ? sc.minAnchorLen_noncan : sc.minAnchorLen)) {
				uint32_t expected = sc.maxIntronLen;
				if(noncan) { if(shorter < 16) expected = 1u << (shorter << 1); }
				else if(shorter < 14) expected = 1u << ((shorter << 1) + 4);
				if(expected > sc.maxIntronLen) expected = sc.maxIntronLen;
				float prob = (float)(other_toff - this_toff) / (float)expected;
				if(prob > 1.0f) prob = 1.0f;
				if(prob > 0.01f) return false;
			}
		}
		if(maxscore < remainsc) return false;
	}
	if(ins || del) {
		if(len > H2G_COMBINE_SCRATCH) { a->overflow = 1; return false; }   // temp_scores capacity (reads up to 512 bp scan exactly; longer ones are flagged)
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
		int i_limit = i < (int)len ? i : (int)len;
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
	if(spliced) {   // :1832-1878 (splice_gap_off is never set in 2.2.3: no indel next to the site)
		const uint32_t addoff = this_rdoff - a->rdoff;
		for(uint32_t i = 0; i < len; i++) {
			const int rdc = seq.at(this_rdoff + i);
			const int64_t p2 = base2 + i;
			const int rfc = (i <= maxscorei) ? rc1.get((int64_t)this_toff + i) : (p2 < 0 ? 4 : rc2.get(p2));
			if(rdc != rfc) hit_push_edit(a, i + addoff, base_char(rfc), base_char(rdc), H2G_EDIT_MM);
			if(i == maxscorei) {
				const uint32_t left = this_toff + i + 1, right = other_toff + other_len - (len - i - 1);
				const uint32_t skipLen = right - left;
				if(a->nedits >= H2G_GHIT_EDITS) a->overflow = 1;
				else a->edits[a->nedits++] = make_spl_edit(i + 1 + addoff, skipLen, maxspldir, site != nullptr, spl_probstore);
			}
		}
	} else {
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
			if(!a->overflow) { a->edits[a->nedits - 1].snp = b->edits[i].snp; a->edits[a->nedits - 1].pad = b->edits[i].pad; }   // whole Edit (ALT id, splice fields)
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
		else if(e.type == H2G_EDIT_SPL) { if(!(o.type == H2G_EDIT_SPL && e.pos == o.pos && spl_len(e) == spl_len(o) && spl_dir(e) == spl_dir(o))) return false; }   // Edit::operator== edit.h:194
		else if(!(e.type == o.type && e.pos == o.pos && e.chr == o.chr && e.qchr == o.qchr)) return false;
	}
	return true;
}

// -I, --fr/--rf/--ff, --nofw/--norc: compiled into the units that define H2G_SPLICE_DB 1 (and the host instantiation), which run either mode;
// the units built without them keep the code they were verified with.  pe_flags: bit 0 gMate1fw, bit 1 gMate2fw, bit 2 gNofw, bit 3 gNorc.
#ifndef H2G_EXT_OPTS
#ifdef H2G_SPLICE_DB
#define H2G_EXT_OPTS H2G_SPLICE_DB
#else
#define H2G_EXT_OPTS 1
#endif
#endif
#define H2G_PE_DEFAULT 1u
H2G_HD uint32_t pe_flags_from(const h2g_align_params& p) {
	const uint32_t m1 = p.pe_orientation == 1 ? 0u : 1u, m2 = p.pe_orientation == 0 ? 0u : 1u;   // fr: 1,0  rf: 0,1  ff: 1,1
	return m1 | (m2 << 1) | (p.nofw ? 4u : 0u) | (p.norc ? 8u : 0u);
}
// ---------------------------------------------------------------------------------------- per-read workspace
struct AlnParams {
	uint32_t khits, kseeds, no_spliced, secondary;
	uint32_t minIntronLen, maxIntronLen, minAnchorLen, minAnchorLen_noncan, minK_local;
	uint32_t xs_only = 0;                                          // --dta-cufflinks
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
	P.minIntronLen = p.min_intronlen; P.maxIntronLen = p.max_intronlen; P.minAnchorLen = p.min_anchor_len; P.minAnchorLen_noncan = p.min_anchor_len_noncan; P.minK_local = 8;
	// xs_only also carries the options of H2G_EXT_OPTS (ctx_ext_opts below): bits 1-4 = pe_flags ^ default, bits 8.. = -I.  The kernel argument
	// block keeps its size and offsets that way, and with them the code of the units built without those options (which only ever see 0 / 1)
	P.xs_only = (p.xs_only ? 1u : 0u) | ((pe_flags_from(p) ^ H2G_PE_DEFAULT) << 1) | (p.min_frag_len << 8);   // tp.h, hi_aligner.h:3986
	P.pseudogeneStop = (linear && !no_spliced) ? 1 : 0; P.anchorStop = 1; P.maxFragLen = p.max_frag_len ? p.max_frag_len : 1000;
	P.bowtie2_dp = p.bowtie2_dp;
	P.scoreMinType = p.score_min_type; P.scoreMinConst = p.score_min_const; P.scoreMinCoeff = p.score_min_coeff;
	P.sc.mmpMax = p.mm_max; P.sc.mmpMin = p.mm_min; P.sc.nPen = p.n_pen; P.sc.rdGapConst = p.rdg_const; P.sc.rdGapLinear = p.rdg_linear;
	P.sc.rfGapConst = p.rfg_const; P.sc.rfGapLinear = p.rfg_linear; P.sc.scMax = p.sc_max; P.sc.scMin = p.sc_min;
	P.sc.minAnchorLen = P.minAnchorLen; P.sc.minAnchorLen_noncan = P.minAnchorLen_noncan; P.sc.maxIntronLen = P.maxIntronLen;
	P.sc.cp = p.pen_cansplice; P.sc.ncp = p.pen_noncansplice;
	P.sc.icpT = p.pen_canintronlen_type; P.sc.icpC = p.pen_canintronlen_const; P.sc.icpL = p.pen_canintronlen_coeff;
	P.sc.incpT = p.pen_noncanintronlen_type; P.sc.incpC = p.pen_noncanintronlen_const; P.sc.incpL = p.pen_noncanintronlen_coeff;
	return P;
}
inline void align_params_defaults(h2g_align_params* p, bool linear) {
	p->khits = linear ? 5 : 10;                       // hisat2.cpp:3903-3906
	p->kseeds = p->khits * 2 > 5 ? p->khits * 2 : 5;  // --max-seeds default hisat2.cpp:3174-3176
	p->no_spliced_alignment = 1; p->secondary = 0; p->bowtie2_dp = 0;
	p->mm_max = 6; p->mm_min = 2; p->n_pen = 1; p->rdg_const = 5; p->rdg_linear = 3; p->rfg_const = 5; p->rfg_linear = 3; p->sc_max = 2; p->sc_min = 1;
	p->score_min_type = 2; p->score_min_const = 0.0; p->score_min_coeff = (double)(-0.2f);
	p->no_temp_splicesite = 0; p->first_read_id = 0;
	p->min_intronlen = 20; p->max_intronlen = 500000; p->pen_cansplice = 0; p->pen_noncansplice = 12;   // hisat2.cpp:493-499
	p->pen_canintronlen_type = 4; p->pen_canintronlen_const = -8.0; p->pen_canintronlen_coeff = 1.0;
	p->pen_noncanintronlen_type = 4; p->pen_noncanintronlen_const = -8.0; p->pen_noncanintronlen_coeff = 1.0;
	p->min_anchor_len = 7; p->min_anchor_len_noncan = 14; p->xs_only = 0; p->use_haplotype = 0; p->max_alts_tried = 16; p->max_frag_len = 1000; p->min_frag_len = 0; p->pe_orientation = 0; p->nofw = 0; p->norc = 0;
}

// One reported alignment = the arguments reportHit (hi_aligner.h:6064-6166) hands to AlnRes::init
struct AlnRec {
	uint32_t fw, tidx, toff, len, trim5, trim3, nedits, splicescore;
	int64_t  score;
	h2g_edit edits[H2G_GHIT_EDITS];  // as stored in the AlnRes: 5'-to-3' positions of the ORIGINAL read relative to the first aligned base
};

struct Frame {                 // one activation of hybridSearch_recur (spliced_aligner.h:331); `state` = the machine pc to resume at
	h2g_ghit hit;
	uint32_t hitoff, hitlen;
	int64_t  maxsc, prev_score;
	uint32_t state;
	uint32_t count, lidx, extoff, extlen, ncoords, nlocal, ti;
	int32_t  ri;
	uint8_t  success, first, use_localindex, uniqueStop;
	uint32_t top, bot, nelt, noext, maxHitLen;   // locals of the local / global search loops that live across machine ops
	h2g_coord coords[AL_MAX_COORDS];
	h2g_ghit  local_hits[AL_MAX_LOCALHITS];
};

struct PartialHit {          // BWTHit hi_aligner.h:108 (linear: node range == row range, no in-edge list)
	uint32_t top, bot, bwoff, len, hit_type, ncoords;
	h2g_coord coords[AL_MAX_GHITS];
};
struct RBHit {               // ReadBWTHit hi_aligner.h:216 (the BWTHit list itself lives in MateArr: this header stays in the hot part)
	uint32_t len, cur, done, numPartialSearch, numUniqueSearch, npartial;
	PartialHit* partial;     // -> MateArr::partial[strand]
};

#define AL_MAX_PAIRS 32
struct MateArr {             // the big lists of one mate (cold part of the workspace)
	PartialHit partial[2][AL_MAX_PARTIAL];       // _hits[rdi][fwi]
	h2g_ghit   searched[AL_MAX_SEARCHED];        // _hits_searched[rdi]
	AlnRec     res[AL_MAX_RESULTS];              // AlnSinkWrap rs1u_ / rs2u_
};
struct MateWS {              // per-mate state of HI_Aligner + the per-mate half of AlnSinkWrap: scalars + pointers into MateArr
	RBHit      rb[2];
	h2g_ghit*  searched;
	AlnRec*    res;
	uint32_t   nsearched, nres;
	int64_t    bestUnp, best2Unp;                // bestUnp1_/bestUnp2_, best2Unp*_
	int64_t    minsc;
	// initRead(rds[1], ..., rightendonly) (hi_aligner.h:3993, hisat2.cpp:3524): the lone mate 2 is searched as rdi 0 but
	// REPORTED into the sink's mate-2 list (:6192), so everything the aligner reads back by rdi — sink.bestUnp1() in
	// nextBWT / align / hybridSearch, redundant() :6311 — sees the empty mate-1 list
	uint32_t   sink_hidden;
};

// Locals of go() / nextBWT / align / getAnchorHits / hybridSearch / alignMate that live across machine ops (h2g_machine.h)
struct GoVars {
	uint32_t paired, nm, slot0, read;
	uint32_t rd_sel[2];              // which read set (0 = mate-1 file, 1 = mate-2 file) go()'s rds[r] is
	uint32_t rdlens[2];
	uint8_t  found[2][2];
	int32_t  sel_r, sel_f;           // nextBWT's selection = the (read, strand) align() works on
	int32_t  nb_rdi, nb_fwi;
	uint32_t sv_rdi, sv_fw, mw_slot; // the SeqView / MateWS the current phase works on
	uint32_t rnd;                    // RandomSource::last
	// getAnchorHits
	uint32_t gh_hi, gh_hj, gh_remained, gh_expected, gh_nco, gh_node, gh_node_end, gh_top, gh_bot, gh_added, gh_edgeIdx, gh_gsize, gh_k, gh_rdoff;
	// hybridSearch
	uint32_t hs_hi, hs_hj, hs_found;
	// mate phase / alignMate
	uint32_t mp_i, mp_j, mp_rs[2], mate_found;
	uint32_t am_lidx, am_first, am_count, am_maxhitlen, am_hitoff, am_hitlen, am_hi, am_fw, am_tidx, am_toff, am_ri, am_nco;
	// hybridSearch_recur driver
	int32_t  sp;
	uint32_t rc_ret_pc, rc_alignMate, pr_ret_pc;
	int64_t  ret, rc_minsc, rc_cushion, fs_tscore;
};

// Per-read workspace.  Every primitive of a read touches a handful of its scalars: those come first, packed into a few cache
// lines (a scattered 4-byte access costs a whole HBM line), the big lists behind them.
struct AlignWS {
	GoVars     gv;
	h2g_fm_hit fh;                               // result of the last partial search
	MateWS     m[2];
	uint32_t   nghits;
	uint8_t    ghit_done[AL_MAX_GHITS];
	// concordant pairs (AlnSinkWrap rs1_/rs2_ as indexes into m[0].res / m[1].res)
	uint32_t   npairs, insp_i, insp_j;           // _concordantIdxInspected
	int64_t    bestPair, best2Pair;
	uint64_t   localindexatts, max_localindexatts;
	uint32_t   overflow;
	uint32_t   nrank, nside, nsteps, nframes_max;   // nrank, nside adjacent: gfm_search updates both through &nrank
	uint8_t    pair_i[AL_MAX_PAIRS], pair_j[AL_MAX_PAIRS];
	// ---- lists
	h2g_ghit   tmp, tmp2;                        // scratch hits
	h2g_ghit   ghits[AL_MAX_GHITS];              // _genomeHits (hitcount lives in .read)
	h2g_coord  am_co[AL_MAX_GHITS];              // alignMate's coordinate list
	Frame      stack[AL_MAX_DEPTH];
	MateArr    marr[2];
};

// Edit::invertPoss edit.cpp:70-111 applied to the k-th element of the inverted list
H2G_HD h2g_edit inverted_edit(const h2g_ghit* h, uint32_t k, uint32_t sz, uint32_t add) {
	h2g_edit e = h->edits[h->nedits - 1 - k];
	uint32_t pos = e.pos + add;
	e.pos = (e.type == H2G_EDIT_READ_GAP || e.type == H2G_EDIT_SPL) ? sz - pos : sz - pos - 1;
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
			if(e.type == H2G_EDIT_SPL && ((e.pad ^ o.pad) & 0x7f)) break;
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
H2G_HD bool al_report(AlignWS* aw, MateWS* ws, const h2g_ghit* hit, uint32_t rdlen, int64_t minsc, bool xs_only = false) {
	if(xs_only) {   // reportHit hi_aligner.h:6101 over GenomeHit::splicing_dir :1128: a spliced alignment whose strand is unknown or mixed
		uint32_t dir = H2G_SPL_UNKNOWN; bool spliced = false, mixed = false;
		for(uint32_t i = 0; i < hit->nedits && !mixed; i++) {
			if(hit->edits[i].type != H2G_EDIT_SPL) continue;
			spliced = true;
			const uint32_t d = spl_dir(hit->edits[i]);
			if(dir == H2G_SPL_UNKNOWN) dir = d;
			else if(d != H2G_SPL_UNKNOWN) {
				const bool a_fw = dir == H2G_SPL_FW || dir == H2G_SPL_SEMI_FW, a_rc = dir == H2G_SPL_RC || dir == H2G_SPL_SEMI_RC;
				if(a_fw && d != H2G_SPL_FW && d != H2G_SPL_SEMI_FW) mixed = true;
				if(a_rc && d != H2G_SPL_RC && d != H2G_SPL_SEMI_RC) mixed = true;
			}
		}
		if(spliced && (mixed || dir == H2G_SPL_UNKNOWN)) return false;
	}
	if(hit->rdoff - hit->trim5 > 0 || hit->len + hit->trim5 + hit->trim3 < rdlen) return false;
	if(hit->score < minsc) return false;
	if(ws->nres >= AL_MAX_RESULTS) { aw->overflow |= 4; return false; }
	AL_TRACE("  REPORT fw %u tidx %u toff %u len %u trim %u/%u score %lld nedits %u\n", hit->fw, hit->tidx, hit->toff, hit->len, hit->trim5, hit->trim3, (long long)hit->score, hit->nedits);
	for(uint32_t q_ = 0; q_ < hit->nedits; q_++) AL_TRACE("      edit pos %u type %u chr %u qchr %u pad %u snp %x\n", hit->edits[q_].pos, hit->edits[q_].type, hit->edits[q_].chr, hit->edits[q_].qchr, hit->edits[q_].pad, hit->edits[q_].snp);
	AlnRec& r = ws->res[ws->nres++];
	r.fw = hit->fw; r.tidx = hit->tidx; r.toff = hit->toff; r.len = hit->len; r.trim5 = hit->trim5; r.trim3 = hit->trim3;
	r.nedits = hit->nedits; r.splicescore = hit->splicescore; r.score = hit->score;
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

#define H2G_COMBINE_MAXLEN 512   // reads up to 512 bp scan exactly in combineWith; longer ones are flagged
struct AlnCtx {
	const DGfm* g;
	const DRef* ref;
	const DLocalSet* ls;
	const AlnParams* P;
	uint8_t* sw = nullptr;   // this lane's Smith-Waterman scratch (sw_scratch_bytes), only when P->bowtie2_dp != 0
	int64_t* sc = nullptr;   // this lane's combineWith temp_scores: 2 x H2G_COMBINE_MAXLEN elements, sc_stride apart (scratch of one primitive)
	uint32_t sc_stride = 1;
	const DAlts* alts = nullptr;      // graph index: the ALT database
	uint32_t rdid_base = 0;           // id of read 0 of the batch (Read::rdid): what the database's visibility window is measured against
	const DSpliceDB* ssdb = nullptr;  // splice sites read from a file (h2g_index_set_splice_sites); nullptr or n == 0: ssdb.empty()
	struct GraphWS* gws = nullptr;    // graph index: this lane's scratch for one primitive (group walk, ALT extension)
	struct GraphSlot* gsl = nullptr;  // graph index: the graph state of the read being worked on
	bool graph = false;               // set from a kernel template constant so that the linear kernels carry no graph code
#if H2G_EXT_OPTS
	uint32_t pe_flags = H2G_PE_DEFAULT, min_frag_len = 0;
#endif
};
#if H2G_EXT_OPTS
H2G_HD void ctx_ext_opts(AlnCtx& C, const AlnParams& P) { C.pe_flags = ((P.xs_only >> 1) & 15u) ^ H2G_PE_DEFAULT; C.min_frag_len = P.xs_only >> 8; }
#endif

// Per-lane scratch of the graph paths (allocated only for graph indexes, so the linear workspace keeps its size):
// group-walk state, ALT-extension state, and the node range + in-edge list of every partial hit (BWTHit::_node_top,
// _node_bot, _node_iedge_count hi_aligner.h:196-199), indexed [mate slot][strand][partial hit].
struct GraphPNode { uint32_t node_top, node_bot; IEdges ie; };
struct GraphWS {               // scratch of ONE primitive execution: per lane of the kernel, not per read
	GwCtx      gw;
	AwaWS      awa;
};
struct GraphSlot {             // graph state that lives as long as the read: per read in flight
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
	if(local_is_linear(*lx.d)) {
		// a local index without a variant in its interval is a LINEAR index even inside a graph index (GFMParams::linearFM:
		// len + 1 == gbwtLen; 64 B sides, no F / M bits): nodes are rows, no in-edges (localGFMSearch hi_aligner.h:6751 over mapLF)
		LIdxW lw; lw.ls = C.ls; lw.d = lx.d;
		const uint32_t nelt = gfm_search(lw, seq, extoff, extlen, top, bot, uniqueStop, P.minK_local, maxHitLen, P.kseeds, true, &ws->nrank);
		C.gsl->node_top = *top; C.gsl->node_bot = *bot; C.gsl->ie.n = 0;
		return nelt;
	}
	const LGfm x = lgfm_of(*C.ls, *lx.d);
	GRange r;
	r.top = *top; r.bot = *bot; r.node_top = r.node_bot = 0;
	const uint32_t nelt = gfm_search_graph(x, lx, seq, extoff, extlen, &r, &C.gsl->ie, uniqueStop, P.minK_local, maxHitLen, P.kseeds, true,
	                                       P.kseeds, &ws->nrank);
	*top = r.top; *bot = r.bot;
	C.gsl->node_top = r.node_top; C.gsl->node_bot = r.node_bot;
	return nelt;
}
H2G_HD void al_local_coords(const AlnCtx& C, AlignWS* ws, uint32_t lidx, uint32_t top, uint32_t bot, uint32_t rdoff, uint32_t rdlen,
                            h2g_coord* coords, uint32_t cap, uint32_t* ncoords)
{
	LIdx lx; lx.ls = C.ls; lx.d = &C.ls->desc[lidx];
	if(!C.graph) { genome_coords_local(lx, top, bot, rdoff, rdlen, coords, cap, ncoords, &ws->nsteps); return; }
	if(local_is_linear(*lx.d)) { LIdxW lw; lw.ls = C.ls; lw.d = lx.d; genome_coords_local(lw, top, bot, rdoff, rdlen, coords, cap, ncoords, &ws->nsteps); return; }
	const LGfm x = lgfm_of(*C.ls, *lx.d);
	const uint32_t node_top = C.gsl->node_top, node_bot = C.gsl->node_bot;
	uint32_t nelt = 0, n = 0;
	*ncoords = 0;
	if(!gw_resolve(x, &C.gws->gw, top, bot, node_top, node_bot, &C.gsl->ie, bot - top, &nelt)) { { ws->overflow |= 512; AL_TRACE("  cap512 at %s:%d\n", __FILE__, __LINE__); } return; }
	ws->nsteps += C.gws->gw.nsteps;
	AL_TRACE("     lcoords top %u bot %u node %u %u -> nelt %u\n", top, bot, node_top, node_bot, nelt);
	for(uint32_t e = 0; e < nelt; e++) {
		h2g_coord c;
		AL_TRACE("      off %u\n", C.gws->gw.offs[e]);
		if(!local_joff_to_coord(*C.ls, lx.d, C.gws->gw.offs[e] & 0xffffu, rdoff, rdlen, &c)) continue;
		if(n < cap) coords[n++] = c; else { ws->overflow |= 512; AL_TRACE("  cap512 at %s:%d\n", __FILE__, __LINE__); }
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
	const uint32_t nelt = gfm_search_graph(*C.g, gx, seq, extoff, extlen, &r, &C.gsl->ie, uniqueStop, C.g->minK, H2G_MAX, P.kseeds, false,
	                                       P.kseeds, &ws->nrank);
	if(nelt > 0) { *top = r.top; *bot = r.bot; }
	C.gsl->node_top = r.node_top; C.gsl->node_bot = r.node_bot;
	return nelt;
}
H2G_HD uint32_t al_global_coords(const AlnCtx& C, AlignWS* ws, uint32_t top, uint32_t bot, uint32_t extlen, h2g_coord* coords, uint32_t cap) {
	h2g_sa_result res;
	if(!C.graph) genome_coords_item(*C.g, top, bot, bot - top, extlen, true, coords, cap, &res);
	else {
		genome_coords_graph_item(*C.g, &C.gws->gw, top, bot, C.gsl->node_top, C.gsl->node_bot, &C.gsl->ie, bot - top, extlen, true,
		                         coords, cap, &res);
		if(res.nsteps == H2G_MAX) { { ws->overflow |= 512; AL_TRACE("  cap512 at %s:%d\n", __FILE__, __LINE__); } res.nsteps = 0; }
	}
	ws->nsteps += res.nsteps;
	return res.ncoords;
}


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
		if(r.edits[k].type == H2G_EDIT_SPL) ext += spl_len(r.edits[k]);
		else if(r.edits[k].type == H2G_EDIT_READ_GAP) ext++;
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

#if H2G_EXT_OPTS
// the same with the policy of --fr / --rf / --ff and -I (pe.cpp:38-133); (off1, len1, fw1) is the LEFT alignment (hi_aligner.h:6020-6029)
H2G_HD bool pe_concordant_ext(int64_t off1, uint32_t len1, bool fw1, int64_t off2, uint32_t len2, bool fw2, uint32_t maxfrag_, uint32_t minfrag_, uint32_t pe_flags) {
	uint64_t maxfrag = maxfrag_;
	if(len1 > maxfrag) maxfrag = len1;
	if(len2 > maxfrag) maxfrag = len2;
	const uint64_t minfrag = minfrag_ < 1 ? 1 : minfrag_;
	const bool m1fw = (pe_flags & 1u) != 0, m2fw = (pe_flags & 2u) != 0;
	bool oneLeft;
	if(m1fw == m2fw) { if(fw1 != fw2) return false; oneLeft = m1fw ? fw1 : !fw1; }   // PE_POLICY_FF / _RR
	else { if(fw1 == fw2) return false; oneLeft = m1fw ? fw1 : !fw1; }              // PE_POLICY_FR / _RF
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
#endif

// pairReads hi_aligner.h:5948-6055 (non-repeat alignments; without H2G_EXT_OPTS: gMate1fw = true, gMate2fw = false)
H2G_HD void al_pair_reads(const AlnParams& P, AlignWS* ws, uint32_t rdlen1, uint32_t rdlen2
#if H2G_EXT_OPTS
                          , uint32_t pe_flags = H2G_PE_DEFAULT, uint32_t min_frag_len = 0
#endif
                          ) {
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
#if H2G_EXT_OPTS
			const bool m1fw = (pe_flags & 1u) != 0, m2fw = (pe_flags & 2u) != 0;
			if((r1.fw != 0) == m1fw) { if((r2.fw != 0) != m2fw) continue; }
			else {
				if((r2.fw != 0) == m2fw) continue;
				int64_t t = l; l = l2; l2 = t; t = r; r = rr2; rr2 = t;
			}
#else
			if(r1.fw) { if(r2.fw) continue; }
			else {
				if(!r2.fw) continue;
				int64_t t = l; l = l2; l2 = t; t = r; r = rr2; rr2 = t;
			}
#endif
			if(l > l2) continue;
			if(r > rr2) continue;
			if(r + (int64_t)P.maxIntronLen < l2) continue;
			bool pass = true;
			if(P.no_spliced) {
#if H2G_EXT_OPTS
				if(r1.toff < r2.toff) pass = pe_concordant_ext(r1.toff, e1, r1.fw != 0, r2.toff, e2, r2.fw != 0, P.maxFragLen, min_frag_len, pe_flags);
				else                  pass = pe_concordant_ext(r2.toff, e2, r2.fw != 0, r1.toff, e1, r1.fw != 0, P.maxFragLen, min_frag_len, pe_flags);
#else
				if(r1.toff < r2.toff) pass = pe_concordant(r1.toff, e1, r1.fw != 0, r2.toff, e2, r2.fw != 0, P.maxFragLen);
				else                  pass = pe_concordant(r2.toff, e2, r2.fw != 0, r1.toff, e1, r1.fw != 0, P.maxFragLen);
#endif
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
// ---------------------------------------------------------------------------------------- finishRead selection (N1)
// AlnSinkWrap::finishRead (aln_sink.h:1939) unpaired branch -> selectByScore (aln_sink.h:2680-2760), -k mode
// (mhits unset): `select` lists the alignments to print, best first; select[0] is the primary.
H2G_HD int64_t hisat2_score(const AlnRec& r) {   // AlnScore::calculate_hisat2_score aligner_result.h:322
	int64_t score = r.score;
	if(score > INT32_MAX) score = INT32_MAX; else if(score < INT32_MIN) score = INT32_MIN;
	int64_t trim = (int64_t)r.trim5 + r.trim3;
	trim = trim > 0xffff ? 0 : 0xffff - trim;
	// transcript score (reportHit hi_aligner.h:6100-6143 over GenomeHit::spliced() :1086): 2 = spliced through database sites only
	// ("known transcripts"), 1 = spliced, 0 = not (the near-a-splice-site case needs --avoid-pseudogene)
	int64_t tscore = 0;
	bool all_known = true;
	for(uint32_t k = 0; k < r.nedits; k++) if(r.edits[k].type == H2G_EDIT_SPL) { tscore = 1; all_known = all_known && spl_known(r.edits[k]); }
	if(tscore && all_known) tscore = 2;
	int64_t spl = (int64_t)r.splicescore / 100;
	spl = spl > 255 ? 0 : 255 - spl;
	return (int64_t)((uint64_t)score << 32) | (0ll << 28) | (tscore << 24) | (spl << 16) | trim;
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
	for(uint32_t i = 0; i < sz; i++) { if(i >= num || nsel >= H2G_SELECT_CAP) break; select[nsel++] = (uint8_t)idx[i]; }
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
	uint32_t best_h2, secbest_h2;
	uint8_t  select[H2G_SELECT_CAP];   // fixed: the same layout in every translation unit whatever its AL_MAX_RESULTS
};

// Scoring::nFilter scoring.cpp:104 with the effective default nCeil = L,0,0.15 (SeedAlignmentPolicy::parseString
// aligner_seed_policy.cpp:294-296 overrides hisat2.cpp:443) + the length filter (hisat2.cpp:3413)
H2G_HD bool read_passes_filters(const SeqView& v) {
	if(v.len < 2) return false;
	const uint32_t maxns = (uint32_t)(0.0 + (double)0.15f * (double)v.len);
	uint32_t ns = 0;
	if(v.pk) {                       // packed read: the N mask words (SeqView::at)
		for(uint32_t w = 0; w < (v.len + 31) / 32; w++) ns += (uint32_t)__builtin_popcount(v.pk[(H2G_PK_WORDS + w) * v.pk_stride]);
		return ns <= maxns;
	}
	for(uint32_t i = 0; i < v.len; i++) if(v.fwc[i] == 4) { if(++ns > maxns) return false; }
	return true;
}
// Paired read: rnd.init(seedA ^ seedB) (hisat2.cpp:3464-3466), go() with both mates.  The concordant /
// discordant / unpaired classification and selection of finishRead stay on the host: the
// caller replays the returned report events into its sink and continues the PRNG from `rnd_state`.
struct PairOut {
	uint32_t nres[2], npairs, overflow, nrank, nsteps, depth, nside, rnd_state, pad;
	uint8_t  pair_i[AL_MAX_PAIRS], pair_j[AL_MAX_PAIRS];
};

}  // namespace h2g

#include "h2g_machine.h"
