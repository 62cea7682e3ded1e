/*
 * h2o.c — CPU ORACLE (test infrastructure; see h2o.h).  Plain C restatement of the
 * HISAT2 2.2.3 seed-and-extend hot path; every function cites the reference lines it
 * follows.  Pinned by tests/test_oracle_golden.py against vectors produced by the real
 * reference classes (oracle/ref_probe.cpp) and against the reference's own SwAligner
 * known-answer cases (aligner_sw.cpp:1470-2727, lifted into tests/golden/sw_kat.json).
 */
#include "h2o.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ file helpers */
typedef struct { uint8_t* d; size_t n, pos; } rbuf;

static int slurp(const char* base, const char* ext, rbuf* b) {
	char fn[4096];
	snprintf(fn, sizeof fn, "%s.%s", base, ext);
	FILE* f = fopen(fn, "rb");
	if(!f) return -1;
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	b->d = (uint8_t*)malloc((size_t)n + 16);
	b->n = (size_t)n;
	b->pos = 0;
	if(fread(b->d, 1, (size_t)n, f) != (size_t)n) { fclose(f); return -1; }
	fclose(f);
	return 0;
}
static uint32_t rd_u32(rbuf* b) { uint32_t v; memcpy(&v, b->d + b->pos, 4); b->pos += 4; return v; }
static uint32_t rd_u16(rbuf* b) { uint16_t v; memcpy(&v, b->d + b->pos, 2); b->pos += 2; return v; }
static uint32_t rd_w(rbuf* b, int wsz) { return wsz == 4 ? rd_u32(b) : rd_u16(b); }
static uint32_t* rd_arr(rbuf* b, int wsz, size_t n) {
	uint32_t* a = (uint32_t*)malloc((n + 1) * sizeof(uint32_t));
	for(size_t i = 0; i < n; i++) a[i] = rd_w(b, wsz);
	return a;
}

/* GFMParams::init  gfm.h:138-185 */
static void params_init(h2o_params* p, uint32_t len, uint32_t gbwtLen, uint32_t numNodes,
                        int32_t lineRate, int32_t offRate, int32_t ftabChars, uint32_t eftabLen, int wsz)
{
	memset(p, 0, sizeof *p);
	p->wsz = wsz;
	uint32_t wmax = wsz == 4 ? 0xffffffffu : 0xffffu;
	p->linear = ((uint32_t)((len + 1) & wmax) == gbwtLen || gbwtLen == 0);
	p->len = len;
	p->gbwtLen = gbwtLen == 0 ? len + 1 : gbwtLen;
	p->numNodes = numNodes == 0 ? len + 1 : numNodes;
	uint32_t gbwtSz = p->linear ? p->gbwtLen / 4 + 1 : p->gbwtLen / 2 + 1;
	p->lineRate = lineRate; p->offRate = offRate; p->ftabChars = ftabChars; p->eftabLen = eftabLen;
	p->offMask = (wmax << offRate) & wmax;
	p->ftabLen = (1u << (ftabChars * 2)) + 1;
	p->offsLen = (p->numNodes + (1u << offRate) - 1) >> offRate;
	p->sideSz = 1u << lineRate;
	if(p->linear) { p->sideGbwtSz = p->sideSz - wsz * 4; p->sideGbwtLen = p->sideGbwtSz << 2; }
	else          { p->sideGbwtSz = p->sideSz - wsz * 6; p->sideGbwtLen = p->sideGbwtSz << 1; }
	p->numSides = (gbwtSz + p->sideGbwtSz - 1) / p->sideGbwtSz;
	p->gbwtTotLen = p->numSides * p->sideSz;
}

/* body shared by GFM::readIntoMemory gfm.h:6039-6332 and LocalGFM::readIntoMemory hgfm.h:1150-1400 */
static void read_gfm_body(rbuf* b, h2o_gfm* g, int wsz) {
	g->nPat = rd_w(b, wsz);
	g->plen = rd_arr(b, wsz, g->nPat);
	g->nFrag = rd_w(b, wsz);
	g->rstarts = rd_arr(b, wsz, (size_t)g->nFrag * 3);
	g->gfm = (uint8_t*)malloc(g->p.gbwtTotLen + 16);
	memcpy(g->gfm, b->d + b->pos, g->p.gbwtTotLen);
	b->pos += g->p.gbwtTotLen;
	g->nZ = rd_w(b, wsz);
	g->zOffs = rd_arr(b, wsz, g->nZ);
	for(int i = 0; i < 5; i++) g->fchr[i] = rd_w(b, wsz);
	g->ftab = rd_arr(b, wsz, g->p.ftabLen);
	g->eftab = rd_arr(b, wsz, g->p.eftabLen);
}

int h2o_index_load(const char* base, h2o_index** out) {
	h2o_index* ix = (h2o_index*)calloc(1, sizeof *ix);
	rbuf b1, b2, b3, b4, b5, b6;
	if(slurp(base, "1.ht2", &b1) || slurp(base, "2.ht2", &b2) || slurp(base, "3.ht2", &b3) ||
	   slurp(base, "4.ht2", &b4)) { free(ix); return -1; }
	/* .1.ht2 header  gfm.h:5917-6000 */
	if(rd_u32(&b1) != 1) return -2;          /* endianness sentinel */
	rd_u32(&b1);                             /* version */
	uint32_t len = rd_u32(&b1), gbwtLen = rd_u32(&b1), numNodes = rd_u32(&b1);
	int32_t lineRate = (int32_t)rd_u32(&b1); rd_u32(&b1);
	int32_t offRate = (int32_t)rd_u32(&b1), ftabChars = (int32_t)rd_u32(&b1);
	uint32_t eftabLen = rd_u32(&b1); rd_u32(&b1); /* flags */
	params_init(&ix->g.p, len, gbwtLen, numNodes, lineRate, offRate, ftabChars, eftabLen, 4);
	read_gfm_body(&b1, &ix->g, 4);
	/* names  gfm.h:6317-6332 */
	ix->names = (char**)calloc(ix->g.nPat + 1, sizeof(char*));
	{
		uint32_t k = 0;
		size_t s = b1.pos;
		while(b1.pos < b1.n && k < ix->g.nPat) {
			char c = (char)b1.d[b1.pos];
			if(c == '\n' || c == '\0') {
				size_t l = b1.pos - s;
				ix->names[k] = (char*)malloc(l + 1);
				memcpy(ix->names[k], b1.d + s, l);
				ix->names[k][l] = 0;
				k++;
				s = b1.pos + 1;
				if(c == '\0') break;
			}
			b1.pos++;
		}
	}
	/* .2.ht2  gfm.h:6334-6432 */
	rd_u32(&b2);
	ix->g.offs = rd_arr(&b2, 4, ix->g.p.offsLen);
	/* .3.ht2 / .4.ht2  reference.cpp:101-190 */
	h2o_ref* r = &ix->r;
	if(rd_u32(&b3) != 1) return -2;
	r->nrecs = rd_u32(&b3);
	r->rec_off = (uint32_t*)malloc(4 * (r->nrecs + 1));
	r->rec_len = (uint32_t*)malloc(4 * (r->nrecs + 1));
	r->rec_first = (uint8_t*)malloc(r->nrecs + 1);
	r->refRecOffs = (uint32_t*)malloc(4 * (r->nrecs + 2));
	r->refOffs = (uint32_t*)malloc(4 * (r->nrecs + 2));
	r->refLens = (uint32_t*)malloc(4 * (r->nrecs + 2));
	uint64_t cumsz = 0, cumlen = 0;
	for(uint32_t i = 0; i < r->nrecs; i++) {
		r->rec_off[i] = rd_u32(&b3);
		r->rec_len[i] = rd_u32(&b3);
		r->rec_first[i] = b3.d[b3.pos++] ? 1 : 0;
		if(r->rec_first[i]) {
			r->refRecOffs[r->nrefs] = i;
			r->refOffs[r->nrefs] = (uint32_t)cumsz;
			if(r->nrefs > 0) r->refLens[r->nrefs - 1] = (uint32_t)cumlen;
			cumlen = 0;
			r->nrefs++;
		}
		cumsz += r->rec_len[i];
		cumlen += r->rec_off[i];
		cumlen += r->rec_len[i];
	}
	r->refRecOffs[r->nrefs] = r->nrecs;
	r->refOffs[r->nrefs] = (uint32_t)cumsz;
	r->refLens[r->nrefs - 1] = (uint32_t)cumlen;
	r->bufSz = cumsz;
	r->buf = b4.d; /* keep */
	/* .5/.6.ht2  hgfm.h:2560-2640, 1130-1400 */
	if(slurp(base, "5.ht2", &b5) == 0 && slurp(base, "6.ht2", &b6) == 0) {
		rd_u32(&b5); rd_u32(&b6);
		ix->nlocal = rd_u32(&b5);
		int32_t llr = (int32_t)rd_u32(&b5); rd_u32(&b5);
		int32_t lor = (int32_t)rd_u32(&b5), lfc = (int32_t)rd_u32(&b5); rd_u32(&b5);
		ix->local = (h2o_gfm*)calloc(ix->nlocal + 1, sizeof(h2o_gfm));
		ix->local_first = (uint32_t*)calloc(ix->g.nPat + 2, 4);
		uint32_t ntext = 0;
		for(uint32_t i = 0; i < ix->nlocal; i++) {
			h2o_gfm* l = &ix->local[i];
			l->tidx = rd_u32(&b5); l->localOffset = rd_u32(&b5); l->joinedOffset = rd_u32(&b5);
			uint32_t llen = rd_u16(&b5), lgl = rd_u16(&b5), lnn = rd_u16(&b5), lel = rd_u16(&b5);
			params_init(&l->p, llen, lgl, lnn, llr, lor, lfc, lel, 2);
			while(ntext <= l->tidx) ix->local_first[ntext++] = i;
			if(llen == 0) continue;
			read_gfm_body(&b5, l, 2);
			l->offs = rd_arr(&b6, 2, l->p.offsLen);
		}
		while(ntext <= ix->g.nPat) ix->local_first[ntext++] = ix->nlocal;
		free(b5.d); free(b6.d);
	}
	/* .7.ht2: ALTs  gfm.h:728-905 (haplotypes/repeats behind them are not needed: use_haplotype defaults to false) */
	{
		rbuf b7;
		if(slurp(base, "7.ht2", &b7) == 0 && b7.n >= 8) {
			rd_u32(&b7);
			uint32_t n = rd_u32(&b7);
			h2o_alt* a = (h2o_alt*)calloc(2 * (size_t)n + 1, sizeof *a);
			uint32_t* ord = (uint32_t*)calloc(2 * (size_t)n + 1, 4);
			uint32_t m = 0;
			for(uint32_t i = 0; i < n && b7.pos + 20 <= b7.n; i++) {
				a[m].pos = rd_u32(&b7); a[m].type = rd_u32(&b7); a[m].len = rd_u32(&b7);
				memcpy(&a[m].seq, b7.d + b7.pos, 8); b7.pos += 8;
				m++;
			}
			uint32_t n0 = m;
			for(uint32_t i = 0; i < n0; i++) {
				if(a[i].type == H2O_ALT_SNP_DEL) {              /* :879-885 */
					a[m] = a[i]; a[m].pos = a[i].pos + a[i].len - 1; a[m].seq = (a[i].seq & ~0xffull) | 1; m++;
				} else if(a[i].type == H2O_ALT_SPLICESITE) {      /* :874-878 */
					a[m] = a[i]; a[m].pos = a[i].len; a[m].len = a[i].pos; m++;
				}
			}
			for(uint32_t i = 0; i < m; i++) ord[i] = i;
			/* buf.sort() of (ALT, original index) pairs :887-903; insertion sort keeps it simple (ALT::operator< alt.h:88) */
			for(uint32_t i = 1; i < m; i++) {
				h2o_alt x = a[i]; uint32_t xo = ord[i]; int j = (int)i - 1;
				while(j >= 0) {
					const h2o_alt* y = &a[j];
					int lt;                                         /* x < y ? */
					if(x.pos != y->pos) lt = x.pos < y->pos;
					else if(x.type != y->type) {
						if(x.type == H2O_ALT_SNP_INS) lt = 1; else if(y->type == H2O_ALT_SNP_INS) lt = 0; else lt = x.type < y->type;
					} else if(x.len != y->len) lt = x.len < y->len;
					else if(x.seq != y->seq) lt = x.seq < y->seq;
					else lt = xo < ord[j];
					if(!lt) break;
					a[j + 1] = a[j]; ord[j + 1] = ord[j]; j--;
				}
				a[j + 1] = x; ord[j + 1] = xo;
			}
			free(ord);
			ix->alts = a; ix->nalts = m;
			free(b7.d);
		}
	}
	/* _minK  hi_aligner.h:3979-3984 */
	{ uint32_t gl = ix->g.p.len; ix->minK = 0; while(gl > 0) { gl >>= 2; ix->minK++; } }
	free(b1.d); free(b2.d); free(b3.d);
	*out = ix;
	return 0;
}

void h2o_index_free(h2o_index* ix) { (void)ix; /* test-lifetime object */ }

void h2o_scoring_default(h2o_scoring* s) { /* scoring.h:29-87 */
	s->mmpMax = 6; s->mmpMin = 2; s->nPen = 1; s->rdGapConst = 5; s->rdGapLinear = 3;
	s->rfGapConst = 5; s->rfGapLinear = 3; s->scMax = 2; s->scMin = 1; s->matchBonus = 0;
}

/* ------------------------------------------------------------------ rank (a3-a5) */
static const uint64_t c_table[4] = { /* gfm.h:75-80 */
	0xffffffffffffffffull, 0xaaaaaaaaaaaaaaaaull, 0x5555555555555555ull, 0x0000000000000000ull };

static inline int countInU64(int c, uint64_t dw) { /* gfm.h:566-578 */
	uint64_t x0 = dw ^ c_table[c];
	uint64_t x1 = x0 >> 1;
	uint64_t x2 = x1 & 0x5555555555555555ull;
	uint64_t x3 = x0 & x2;
	return __builtin_popcountll(x3);
}

/* countUpTo, POPCNT path  gfm.h:3166-3201 */
static uint32_t countUpTo(const uint8_t* side, int by_, int bp, int c) {
	uint32_t cCnt = 0;
	int i = 0;
	int by = by_ + (bp > 0 ? 1 : 0);
	for(; i < by; i += 8) {
		uint64_t w;
		memcpy(&w, side + i, 8);
		if(i + 8 < by) {
			cCnt += countInU64(c, w);
		} else {
			uint32_t by_shift = 8 - (by - i);
			uint32_t bp_shift = (bp > 0 ? 4 - bp : 0);
			uint32_t shift = (by_shift << 3) + (bp_shift << 1);
			w <<= shift;
			uint32_t add = countInU64(c, w);
			if(c == 0) add -= (shift >> 1);
			cCnt += add;
			break;
		}
	}
	return cCnt;
}

uint32_t h2o_rank(const h2o_gfm* g, uint32_t row, int c) { /* SideLocus::initFromRow gfm.h:376-393; countBt2Side gfm.h:2958-3001 */
	const h2o_params* p = &g->p;
	uint32_t sideNum = row / p->sideGbwtLen, charOff = row % p->sideGbwtLen;
	const uint8_t* side = g->gfm + (size_t)sideNum * p->sideSz;
	int by = charOff >> 2, bp = charOff & 3;
	uint32_t cCnt = countUpTo(side, by, bp, c);
	if(c == 0) {
		for(uint32_t i = 0; i < g->nZ; i++) { /* '$' stored as A: gfm.h:2967-2979 with postReadInit :2783 */
			uint32_t zs = g->zOffs[i] / p->sideGbwtLen, zc = g->zOffs[i] % p->sideGbwtLen;
			if(zs == sideNum && zc < charOff) cCnt--;
		}
	}
	const uint8_t* acgt8 = side + p->sideGbwtSz + (p->linear ? 0 : 2 * p->wsz);
	uint32_t occ;
	if(p->wsz == 4) { memcpy(&occ, acgt8 + 4 * c, 4); }
	else { uint16_t o; memcpy(&o, acgt8 + 2 * c, 2); occ = o; }
	return occ + cCnt + g->fchr[c];
}

int h2o_rowL(const h2o_gfm* g, uint32_t row) { /* gfm.h:3615-3630 */
	const h2o_params* p = &g->p;
	uint32_t sideNum = row / p->sideGbwtLen, charOff = row % p->sideGbwtLen;
	const uint8_t* side = g->gfm + (size_t)sideNum * p->sideSz;
	return (side[charOff >> 2] >> ((charOff & 3) * 2)) & 3;
}

/* ------------------------------------------------------------------ graph index (a2/a7/a9) */
/* 128 B graph side (gfm.h:160-176, 6204): [0,52) 2-bit gbwt chars (208), [52,78) F bits, [78,104) M bits,
 * then index_t F_loc, M_occ, occ[A,C,G,T].  Bit j of byte i is position 8*i+j. */
static uint32_t pop64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }
static uint32_t countUpTo_bits(const uint8_t* bits, int by, int bp) { /* gfm.h:3384-3449 */
	uint32_t cCnt = 0;
	int n = by + (bp > 0 ? 1 : 0);
	for(int i = 0; i < n; i += 8) {
		uint64_t w;
		memcpy(&w, bits + i, 8);
		if(i + 8 < n) cCnt += pop64(w);
		else {
			uint32_t by_shift = 8 - (n - i);
			uint32_t bp_shift = (bp > 0 ? 8 - bp : 0);
			w <<= (by_shift << 3) + bp_shift;
			cCnt += pop64(w);
			break;
		}
	}
	return cCnt;
}
static uint32_t side_u32(const h2o_gfm* g, uint32_t sideNum, uint32_t k) { /* k: 0 F_loc, 1 M_occ (index_t / local_index_t words) */
	const uint8_t* p = g->gfm + (size_t)sideNum * g->p.sideSz + g->p.sideGbwtSz + (size_t)g->p.wsz * k;
	if(g->p.wsz == 4) { uint32_t v; memcpy(&v, p, 4); return v; }
	uint16_t v; memcpy(&v, p, 2); return v;
}
uint32_t h2o_rank_M(const h2o_gfm* g, uint32_t row) { /* initFromRow_bit gfm.h:428; rank_M :4100; countMSide :3146 */
	const h2o_params* p = &g->p;
	uint32_t sideNum = row / p->sideGbwtLen, charOff = row % p->sideGbwtLen;
	const uint8_t* side = g->gfm + (size_t)sideNum * p->sideSz;
	const uint8_t* M = side + (p->sideGbwtSz - (p->sideGbwtSz >> 2));
	return countUpTo_bits(M, (int)(charOff >> 3), (int)(charOff & 7)) + side_u32(g, sideNum, 1);
}
uint32_t h2o_select_F(const h2o_gfm* g, uint32_t row, uint32_t count) { /* gfm.h:4113-4167 */
	const h2o_params* p = &g->p;
	uint32_t sideNum = row / p->sideGbwtLen, charOff = row % p->sideGbwtLen;
	const uint32_t bitsPerSide = p->sideGbwtSz << 1;
	while(1) {
		const uint8_t* F = g->gfm + (size_t)sideNum * p->sideSz + (p->sideGbwtSz >> 1);
		uint32_t by = charOff >> 3, bp = charOff & 7;
		uint32_t remaining = bitsPerSide - charOff;
		uint32_t minSide = count < remaining ? count : remaining;
		uint64_t bits;
		memcpy(&bits, F + by, 8);
		uint32_t advance = 64;
		if(bp > 0) { bits >>= bp; advance -= bp; }
		if(minSide < advance) { advance = minSide; bits <<= (64 - minSide); }
		count -= pop64(bits);
		if(count == 0) { charOff += advance - 1; break; }
		if(charOff + advance == bitsPerSide) { sideNum++; charOff = 0; }
		else charOff += advance;
	}
	return sideNum * p->sideGbwtLen + charOff;
}
uint32_t h2o_in_edge_count(const h2o_gfm* g, uint32_t top, uint32_t bot, uint32_t* iedges, uint32_t cap)
{ /* gfm.h:4172-4213: (node index relative to the first node, extra in-edges) for nodes with >1 in-edge */
	const h2o_params* p = &g->p;
	uint32_t n = 0, curr_node = 0, num0s = 0;
	int first = 1;
	for(uint32_t row = top; row < bot; row++) {
		if(first) first = 0;
		else {
			uint32_t sideNum = row / p->sideGbwtLen, charOff = row % p->sideGbwtLen;
			const uint8_t* F = g->gfm + (size_t)sideNum * p->sideSz + (p->sideGbwtSz >> 1);
			int bit = (F[charOff >> 3] >> (charOff & 7)) & 1;
			if(bit) { curr_node++; num0s = 0; }
			else {
				num0s++;
				if(num0s == 1) { if(n < cap) iedges[2 * n] = curr_node; n++; }
				if(n <= cap) iedges[2 * (n - 1) + 1] = num0s;
			}
		}
	}
	return n;
}
/* F-row of node `node` (0-based) given a row whose side seeds the backward scan over the (F_loc, M_occ) side
 * headers; shared by mapGLF :3785-3810 and mapGLF1 :3975-3998 */
static uint32_t node_to_Frow(const h2o_gfm* g, uint32_t locRow, uint32_t node, uint32_t* F_loc_out, uint32_t* M_occ_out) {
	uint32_t sideNum = locRow / g->p.sideGbwtLen;
	uint32_t F_loc, M_occ;
	while(1) {
		F_loc = side_u32(g, sideNum, 0);
		M_occ = side_u32(g, sideNum, 1);
		if(M_occ <= node) break;
		sideNum--;
	}
	if(M_occ > 0) F_loc++;
	*F_loc_out = F_loc; *M_occ_out = M_occ;
	if(node + 1 > M_occ) return h2o_select_F(g, F_loc, node + 1 - M_occ);
	return F_loc;
}
/* mapGLF gfm.h:3759-3837.  Returns 0 (and top=bot=0) when the range is empty. */
int h2o_map_glf(const h2o_gfm* g, uint32_t top, uint32_t bot, int c, uint32_t k, uint32_t* otop, uint32_t* obot,
                uint32_t* ontop, uint32_t* onbot, uint32_t* iedges, uint32_t cap, uint32_t* niedges)
{
	uint32_t t = h2o_rank(g, top, c), b = h2o_rank(g, bot, c);
	*niedges = 0;
	if(g->p.linear) { *otop = *ontop = t; *obot = *onbot = b; return t < b; }
	*otop = *obot = 0;
	if(t + 1 >= g->p.gbwtLen || t >= b) return 0;
	uint32_t node_top = h2o_rank_M(g, t + 1) - 1;
	uint32_t F_loc, M_occ;
	uint32_t ft = node_to_Frow(g, t + 1, node_top, &F_loc, &M_occ);
	uint32_t node_bot = h2o_rank_M(g, b);
	/* :3812-3827 — the bottom uses the side of `bot` directly (no backward scan) */
	uint32_t bs = b / g->p.sideGbwtLen;
	uint32_t bF = side_u32(g, bs, 0), bM = side_u32(g, bs, 1);
	if(bM > 0) bF++;
	uint32_t fb = (node_bot + 1 > bM) ? h2o_select_F(g, bF, node_bot + 1 - bM) : bF;
	*otop = ft; *obot = fb; *ontop = node_top; *onbot = node_bot;
	if(iedges && node_bot - node_top <= k && node_bot - node_top < fb - ft)
		*niedges = h2o_in_edge_count(g, ft, fb, iedges, cap);
	return 1;
}
/* mapGLF1 gfm.h:3957-4028 (mapLF1 :3892).  Returns 0 when the row cannot be extended by c. */
int h2o_map_glf1(const h2o_gfm* g, uint32_t row, int c, uint32_t* otop, uint32_t* obot, uint32_t* ontop, uint32_t* onbot)
{
	*otop = *obot = 0;
	if(h2o_rowL(g, row) != c) return 0;
	for(uint32_t i = 0; i < g->nZ; i++) if(row == g->zOffs[i]) return 0;
	uint32_t t = h2o_rank(g, row, c);
	if(g->p.linear) { *otop = *ontop = t; *obot = *onbot = t + 1; return 1; }
	uint32_t node_top = h2o_rank_M(g, t + 1) - 1;
	uint32_t F_loc, M_occ;
	uint32_t ft = node_to_Frow(g, t + 1, node_top, &F_loc, &M_occ);
	uint32_t node_bot = node_top + 1;
	uint32_t fb = (node_bot + 1 > M_occ) ? h2o_select_F(g, F_loc, node_bot + 1 - M_occ) : F_loc;
	*otop = ft; *obot = fb; *ontop = node_top; *onbot = node_bot;
	return 1;
}

/* ------------------------------------------------------------------ ftab (a10) */
static uint32_t ftabHi(const h2o_gfm* g, uint32_t i) { /* gfm.h:2618-2634 */
	uint32_t lim = g->p.linear ? g->p.len : g->p.gbwtLen;
	uint32_t wmax = g->p.wsz == 4 ? 0xffffffffu : 0xffffu;
	if(g->ftab[i] <= lim) return g->ftab[i];
	uint32_t ef = g->ftab[i] ^ wmax;
	return g->eftab[ef * 2 + 1];
}
static uint32_t ftabLo(const h2o_gfm* g, uint32_t i) { /* gfm.h:2696-2712 */
	uint32_t lim = g->p.linear ? g->p.len : g->p.gbwtLen;
	uint32_t wmax = g->p.wsz == 4 ? 0xffffffffu : 0xffffu;
	if(g->ftab[i] <= lim) return g->ftab[i];
	uint32_t ef = g->ftab[i] ^ wmax;
	return g->eftab[ef * 2];
}
int h2o_ftab_lohi(const h2o_gfm* g, const uint8_t* seq, uint32_t off, uint32_t* top, uint32_t* bot) {
	/* ftabSeqToInt (fw index, rev=false) gfm.h:2569-2594; ftabLoHi gfm.h:2670-2685 */
	uint32_t fi = 0;
	for(int i = 0; i < g->p.ftabChars; i++) {
		int c = seq[off + i];
		if(c > 3) return 0;
		fi = (fi << 2) | (uint32_t)c;
	}
	*top = ftabHi(g, fi);
	*bot = ftabLo(g, fi + 1);
	return 1;
}

/* ------------------------------------------------------------------ SA offset (a14) */
static int is_zoff(const h2o_gfm* g, uint32_t row) {
	for(uint32_t i = 0; i < g->nZ; i++) if(row == g->zOffs[i]) return 1;
	return 0;
}
/* Linear-index meaning of GWState::init/advance + tryOffset (group_walk.h:509-560, 1035-1336;
 * gfm.h:2719-2733) == getOffset (gfm.h:5682-5716): walk LF until a sampled row or '$'. */
uint32_t h2o_get_offset(const h2o_gfm* g, uint32_t row, uint32_t* steps) {
	uint32_t jumps = 0;
	while(1) {
		if(is_zoff(g, row)) break;
		if((row & g->p.offMask) == row) {
			uint32_t off = g->offs[row >> g->p.offRate];
			uint32_t wmax = g->p.wsz == 4 ? 0xffffffffu : 0xffffu;
			if(off != wmax) { if(steps) *steps = jumps; return off + jumps; }
		}
		int c = h2o_rowL(g, row);
		row = h2o_rank(g, row, c);
		jumps++;
	}
	if(steps) *steps = jumps;
	return jumps;
}

int h2o_joined_to_text(const h2o_gfm* g, uint32_t qlen, uint32_t off, uint32_t* tidx, uint32_t* textoff,
                       uint32_t* tlen, int rejectStraddle, int* straddled) /* gfm.h:5527-5600 */
{
	uint32_t top = 0, bot = g->nFrag, elt = H2O_MAX;
	while(1) {
		uint32_t oldelt = elt;
		elt = top + ((bot - top) >> 1);
		if(oldelt == elt) { *tidx = H2O_MAX; return 0; }
		uint32_t lower = g->rstarts[elt * 3];
		uint32_t upper = (elt == g->nFrag - 1) ? g->p.len : g->rstarts[(elt + 1) * 3];
		if(lower <= off) {
			if(upper > off) {
				if(off + qlen > upper) {
					*straddled = 1;
					if(rejectStraddle) { *tidx = H2O_MAX; return 0; }
				}
				*tidx = g->rstarts[elt * 3 + 1];
				uint32_t fragoff = off - g->rstarts[elt * 3];
				*textoff = fragoff + g->rstarts[elt * 3 + 2];
				break;
			} else top = elt;
		} else bot = elt;
	}
	*tlen = g->plen[*tidx];
	return 1;
}

/* ------------------------------------------------------------------ reference (a17) */
/* Meaning of BitPairReference::getStretch reference.cpp:486-650 (== getStretchNaive :404-470):
 * dest[i] = base of text tidx at toff+i (0..3), 4 for N / outside the sequence. */
void h2o_get_stretch(const h2o_ref* r, uint32_t tidx, int64_t toff, uint32_t count, uint8_t* dest) {
	memset(dest, 4, count);
	uint64_t reci = r->refRecOffs[tidx], recf = r->refRecOffs[tidx + 1];
	uint64_t bufOff = r->refOffs[tidx];
	int64_t off = 0;
	for(uint64_t i = reci; i < recf; i++) {
		off += r->rec_off[i];
		int64_t lo = off, hi = off + r->rec_len[i];     /* unambiguous stretch [lo,hi) */
		int64_t a = toff > lo ? toff : lo, b = toff + count < hi ? toff + (int64_t)count : hi;
		for(int64_t t = a; t < b; t++) {
			uint64_t bo = bufOff + (uint64_t)(t - lo);
			dest[t - toff] = (r->buf[bo >> 2] >> ((bo & 3) << 1)) & 3;
		}
		bufOff += r->rec_len[i];
		off = hi;
		if(off >= toff + (int64_t)count) break;
	}
}

/* ------------------------------------------------------------------ partialSearch (a11) */
void h2o_partial_search(const h2o_index* ix, const uint8_t* seq, uint32_t len, uint32_t cur_in,
                        int pseudogeneStop_in, int anchorStop_in, uint32_t khits, h2o_bwthit* o)
{
	uint32_t n = 0;
	h2o_partial_search_graph(ix, seq, len, cur_in, pseudogeneStop_in, anchorStop_in, khits,
	                         5 > khits * 2 ? 5 : khits * 2, o, NULL, 0, &n);
}

/* hi_aligner.h:6361-6600 for linear and graph indexes; iedges[2*cap] receives BWTHit::_node_iedge_count */
void h2o_partial_search_graph(const h2o_index* ix, const uint8_t* seq, uint32_t len, uint32_t cur_in,
                              int pseudogeneStop_in, int anchorStop_in, uint32_t khits, uint32_t kseeds,
                              h2o_bwthit* o, uint32_t* iedges, uint32_t cap, uint32_t* niedges)
{ /* mapLF gfm.h:3739; mapGLF :3759; mapGLF1 :3957; mapLF1 :3892 */
	enum { IE_CAP = 64 };
	uint32_t tmp_ie[2 * IE_CAP], cur_ie[2 * IE_CAP], tmp_n = 0, cur_n = 0;
	*niedges = 0;
	const h2o_gfm* g = &ix->g;
	const uint32_t ftabLen = (uint32_t)g->p.ftabChars, minK = ix->minK;
	int pseudogeneStop_ = pseudogeneStop_in, anchorStop_ = anchorStop_in;
	int pseudogeneStop = 0, anchorStop = 0;
	memset(o, 0, sizeof *o);
	o->top = o->bot = o->node_top = o->node_bot = H2O_MAX;
	o->hit_type = H2O_CANDIDATE_HIT;
	o->numPartialSearch = 1;
	uint32_t cur = cur_in, offset = cur_in, dep = cur_in;
	uint32_t left = len - dep;
	o->bwoff = offset;
	if(left < ftabLen + 1) {                                  /* :6403-6417 */
		cur = len; o->len = cur - offset; o->cur = cur; o->done = 1; return;
	}
	for(uint32_t i = 0; i < ftabLen; i++) {                  /* :6419-6437 */
		int c = seq[len - dep - 1 - i];
		if(c > 3) {
			cur += (i + 1); o->len = cur - offset; o->cur = cur; if(cur >= len) o->done = 1; return;
		}
	}
	uint32_t top = 0, bot = 0;
	h2o_ftab_lohi(g, seq, len - dep - ftabLen, &top, &bot);  /* :6440 */
	dep += ftabLen;
	if(top >= bot) {                                         /* :6442-6457 */
		cur = dep; o->len = cur - offset; o->cur = cur; if(cur >= len) o->done = 1; return;
	}
	uint32_t same_range = 0, similar_range = 0;
	uint32_t ntop = 0, nbot = 0;                              /* node_range, initially (0,0) */
	while(dep < len) {                                       /* :6459-6539 */
		int c = seq[len - dep - 1];
		uint32_t ttop = 0, tbot = 0, tntop = 0, tnbot = 0;
		tmp_n = 0;
		if(c <= 3) {
			if(bot - top > 1) {                               /* bloc.valid(): mapLF / mapGLF on both loci */
				o->nrank += 2;
				uint32_t s0 = top / g->p.sideGbwtLen, s1 = bot / g->p.sideGbwtLen;
				o->nside += (s0 == s1) ? 1 : 2;
				if(g->p.linear) {
					ttop = tntop = h2o_rank(g, top, c);
					tbot = tnbot = h2o_rank(g, bot, c);
				} else {
					h2o_map_glf(g, top, bot, c, kseeds, &ttop, &tbot, &tntop, &tnbot, tmp_ie, IE_CAP, &tmp_n);
				}
			} else {                                         /* mapGLF1 -> mapLF1 */
				o->nrank += 1;
				o->nside += 1;
				if(h2o_map_glf1(g, top, c, &ttop, &tbot, &tntop, &tnbot)) {
					if(ttop + 1 < tbot) { tmp_ie[0] = 0; tmp_ie[1] = tbot - ttop - 1; tmp_n = 1; } /* :6476-6482 */
				}
			}
		}
		if(ttop >= tbot) break;
		uint32_t nt = tnbot - tntop, no = nbot - ntop;
		if(pseudogeneStop_) {                                /* :6488-6503 */
			if(nt < no && no <= (5u < khits ? 5u : khits)) {
				if(dep - offset >= minK + 6 && similar_range >= 5) {
					o->numUniqueSearch++; pseudogeneStop = 1; break;
				}
			}
			if(nt != 1) {
				if(nt + 2 >= no) similar_range++;
				else if(nt + 4 < no) similar_range = 0;
			} else pseudogeneStop_ = 0;
		}
		if(anchorStop_) {                                    /* :6505-6519 */
			if(nt != 1 && no == nt) {
				same_range++;
				if(same_range >= 5) anchorStop_ = 0;
			} else same_range = 0;
			if(dep - offset >= minK + 8 && nt >= 4) anchorStop_ = 0;
		}
		top = ttop; bot = tbot; ntop = tntop; nbot = tnbot;
		cur_n = tmp_n;                                       /* :6522-6527 */
		memcpy(cur_ie, tmp_ie, sizeof(uint32_t) * 2 * (tmp_n < IE_CAP ? tmp_n : IE_CAP));
		dep++;
		if(anchorStop_) {                                    /* :6530-6536 */
			if(dep - offset >= minK + 12 && bot - top == 1) {
				o->numUniqueSearch++; anchorStop = 1; break;
			}
		}
	}
	if(top < bot) {                                          /* :6542-6598 */
		uint32_t hit_type = H2O_CANDIDATE_HIT;
		if(anchorStop) hit_type = H2O_ANCHOR_HIT;
		else if(pseudogeneStop) hit_type = H2O_PSEUDOGENE_HIT;
		int report = ntop < nbot;   /* no LF step taken => node_range (0,0) => not reported */
		if(nbot - ntop < bot - top && cur_n == 0) report = 0;  /* :6551-6553 */
		if(report) {
			o->top = top; o->bot = bot; o->node_top = ntop; o->node_bot = nbot;
			*niedges = cur_n;
			for(uint32_t e = 0; e < cur_n && e < cap; e++) { iedges[2 * e] = cur_ie[2 * e]; iedges[2 * e + 1] = cur_ie[2 * e + 1]; }
		}
		o->len = dep - offset;
		o->hit_type = hit_type;
		cur = dep;
		if(cur >= len) {
			if(hit_type == H2O_CANDIDATE_HIT) o->numUniqueSearch++;
			o->done = 1;
		}
		o->cur = cur;
	} else {
		/* unreachable: range only shrinks to empty through the break above, which keeps
		 * the previous non-empty range */
		o->cur = cur;
	}
	o->pseudogeneStop = pseudogeneStop;
	o->anchorStop = anchorStop;
}

/* ------------------------------------------------------------------ getGenomeCoords (a14) */
int h2o_genome_coords(const h2o_index* ix, uint32_t top, uint32_t bot, uint32_t maxelt, uint32_t rdlen,
                      int rejectStraddle, h2o_coord* coords, uint32_t* ncoords, int* straddled, uint32_t* nsteps)
{ /* hi_aligner.h:5774-5855 (linear: node range == [top,bot)) */
	*straddled = 0;
	uint32_t nelt = bot - top;
	if(nelt > maxelt) nelt = maxelt;
	uint32_t n = *ncoords;
	for(uint32_t e = 0; e < nelt; e++) {
		uint32_t st = 0;
		uint32_t joff = h2o_get_offset(&ix->g, top + e, &st);
		if(nsteps) *nsteps += st;
		uint32_t tidx = 0, toff = 0, tlen = 0;
		int st2 = 0;
		h2o_joined_to_text(&ix->g, rdlen, joff, &tidx, &toff, &tlen, rejectStraddle, &st2);
		*straddled |= st2;
		if(tidx == H2O_MAX) { *ncoords = n; return 0; }
		coords[n].tidx = st2 ? H2O_MAX : tidx;
		coords[n].toff = toff;
		coords[n].joinedOff = joff;
		n++;
	}
	*ncoords = n;
	return 1;
}

/* ------------------------------------------------------------------ extend (a18, a19) */
static int mm_pen(const h2o_scoring* sc, int q) { /* Scoring::initPens COST_MODEL_QUAL scoring.h:117-124 */
	if(q < 0) q = 0;
	int ii = q < 40 ? q : 40;
	float frac = (float)ii / 40.0f;
	return sc->mmpMin + (int)(frac * (sc->mmpMax - sc->mmpMin));
}

static int sc_pen(const h2o_scoring* sc, int q) { /* scoring.h:312-318 */
	if(q <= 33) return sc->scMin;
	q -= 33;
	if(q > 40) q = 40;
	return (int)((q / 40.0f) * (sc->scMax - sc->scMin) + sc->scMin);
}

int64_t h2o_calculate_score(const h2o_scoring* sc, const char* qual, h2o_ghit* h) { /* hi_aligner.h:3711-3891 (no splice edits) */
	int64_t score = 0;
	uint32_t mm = 0;
	for(uint32_t i = 0; i < h->nedits; i++) {
		const h2o_edit* e = &h->edits[i];
		if(e->snp != H2O_MAX) continue;                      /* known-variant edits cost nothing (:3737, :3846, :3858) */
		if(e->type == H2O_EDIT_MM) {
			int q = qual[h->rdoff + e->pos] - 33;
			/* Scoring::score(rdc, refm, q) scoring.h:259-269 */
			int rdc = e->qchr == 'A' ? 0 : e->qchr == 'C' ? 1 : e->qchr == 'G' ? 2 : e->qchr == 'T' ? 3 : 4;
			if(rdc > 3) score -= sc->nPen;
			else if(e->chr == 'N') score += sc->matchBonus;   /* mask 15 contains every base */
			else score -= mm_pen(sc, q);
			mm++;
		} else if(e->type == H2O_EDIT_READ_GAP) {
			int open = !(i > 0 && h->edits[i - 1].type == H2O_EDIT_READ_GAP && h->edits[i - 1].pos == e->pos);
			score -= open ? (sc->rdGapConst + sc->rdGapLinear) : sc->rdGapLinear;
		} else if(e->type == H2O_EDIT_REF_GAP) {
			int open = !(i > 0 && h->edits[i - 1].type == H2O_EDIT_REF_GAP && h->edits[i - 1].pos + 1 == e->pos);
			score -= open ? (sc->rfGapConst + sc->rfGapLinear) : sc->rfGapLinear;
		}
	}
	/* soft-clip penalty :3868-3874 — Scoring::sc(q) scoring.h:312-318 (takes the raw quality char) */
	for(uint32_t i = 0; i < h->trim5; i++) score -= sc_pen(sc, qual[i]);
	for(uint32_t i = 0; i < h->trim3; i++) score -= sc_pen(sc, qual[i]);
	score += (int64_t)(h->len - mm) * sc->matchBonus;
	h->score = score;
	return score;
}

static void edits_insert_front(h2o_edit* e, uint32_t* n, h2o_edit x) {
	if(*n >= H2O_MAX_EDITS) return;
	memmove(e + 1, e, *n * sizeof *e);
	e[0] = x; (*n)++;
}

/* alignWithALTs (hi_aligner.h:683-783) + alignWithALTs_recur without ALTs (:2763-2853 left, :3168-3216 right) */
static uint32_t align_no_alts(const h2o_index* ix, uint32_t joinedOff, const uint8_t* rdseq, uint32_t base_rdoff,
                              uint32_t rdoff, uint32_t rdlen, uint32_t tidx, int rfoff, uint32_t rflen, int left,
                              h2o_edit* edits, uint32_t* nedits_io, uint32_t mm, uint32_t* numNs)
{
	(void)joinedOff;
	int best_rdoff = (int)rdoff;
	if(numNs) *numNs = 0;
	uint32_t nedits = *nedits_io;
	h2o_edit tmp[H2O_MAX_EDITS];
	uint32_t ntmp = nedits;
	memcpy(tmp, edits, nedits * sizeof *tmp);
	uint32_t rdoff_add = rdoff - base_rdoff;
	do { /* _recur, dep 0 */
		if(rfoff < -16) break;
		uint32_t contig_len = ix->r.refLens[tidx];
		if(rfoff >= (int64_t)contig_len) break;
		if(rfoff >= 0 && (uint64_t)rfoff + rflen > contig_len) rflen = contig_len - rfoff;
		else if(rfoff < 0 && rflen > contig_len) rflen = contig_len;
		if(rflen == 0) break;
		uint8_t rfbuf[1100];
		if(rflen > 1024) rflen = 1024;
		{
			/* rfseq = raw_refbuf + 16 + off + min(rfoff,0) over a buffer pre-filled with 4 */
			memset(rfbuf, 4, sizeof rfbuf);
			int s = rfoff > 0 ? rfoff : 0;
			uint32_t cnt = rfoff > 0 ? rflen : rflen + rfoff;
			h2o_get_stretch(&ix->r, tidx, s, cnt, rfbuf + 32);
		}
		const uint8_t* rfseq = rfbuf + 32 + (rfoff < 0 ? rfoff : 0);
		if(left) {
			uint32_t tmp_mm = 0, mm_tmp_numNs = 0;
			int mm_min_rd_i = (int)rdoff;
			for(int rf_i = (int)rflen - 1; rf_i >= 0 && mm_min_rd_i >= 0; rf_i--, mm_min_rd_i--) {
				int rf_bp = rfseq[rf_i], rd_bp = rdseq[mm_min_rd_i];
				if(rf_bp != rd_bp || rd_bp == 4) {
					if(tmp_mm >= mm) break;
					tmp_mm++;
					h2o_edit e = { (uint32_t)mm_min_rd_i, (uint8_t)"ACGTN"[rf_bp], (uint8_t)"ACGTN"[rd_bp], H2O_EDIT_MM, 0, H2O_MAX };
					edits_insert_front(tmp, &ntmp, e);
				}
				if(rf_bp == 4) mm_tmp_numNs++;
			}
			if(mm_min_rd_i < best_rdoff) {
				best_rdoff = mm_min_rd_i;
				memcpy(edits, tmp, ntmp * sizeof *tmp); *nedits_io = ntmp;
				if(numNs) *numNs = mm_tmp_numNs;
			}
		} else {
			uint32_t tmp_mm = 0, mm_max_rd_i = 0;
			for(uint32_t rf_i = 0; rf_i < rflen && mm_max_rd_i < rdlen; rf_i++, mm_max_rd_i++) {
				int rf_bp = rfseq[rf_i], rd_bp = rdseq[rdoff + mm_max_rd_i];
				if(rf_bp != rd_bp || rd_bp == 4) {
					if(tmp_mm >= mm) break;
					tmp_mm++;
					if(ntmp < H2O_MAX_EDITS) {
						h2o_edit e = { mm_max_rd_i + rdoff_add, (uint8_t)"ACGTN"[rf_bp], (uint8_t)"ACGTN"[rd_bp], H2O_EDIT_MM, 0, H2O_MAX };
						tmp[ntmp++] = e;
					}
				}
			}
			if((int)(mm_max_rd_i + rdoff) > best_rdoff) {
				best_rdoff = (int)(mm_max_rd_i + rdoff);
				memcpy(edits, tmp, ntmp * sizeof *tmp); *nedits_io = ntmp;
			}
		}
	} while(0);
	uint32_t extlen = left ? rdoff - best_rdoff : best_rdoff - rdoff;   /* :741-750 */
	uint32_t ne = *nedits_io;
	if(extlen > 0 && ne > 0) {                                            /* :751-779 */
		const h2o_edit* f = &edits[0];
		if(f->pos + extlen == base_rdoff + 1) {
			if(f->type == H2O_EDIT_READ_GAP || f->type == H2O_EDIT_REF_GAP) extlen = 0;
			if(f->type == H2O_EDIT_MM && f->chr == 'N') extlen = 0;
		}
		const h2o_edit* b = &edits[ne - 1];
		if(extlen > 0 && b->pos == rdoff - base_rdoff + extlen - 1) {
			if(b->type == H2O_EDIT_READ_GAP || b->type == H2O_EDIT_REF_GAP) extlen = 0;
		}
		if(extlen == 0 && ne > nedits) {
			if(left) memmove(edits, edits + (ne - nedits), nedits * sizeof *edits);
			*nedits_io = nedits;
		}
	}
	return extlen;
}

/* ---- ALT-aware extension: alignWithALTs (hi_aligner.h:683-783) + alignWithALTs_recur (:2763-3550) for SNP ALTs
 * (single / insertion / deletion).  Splice-site and exon ALTs are skipped (not present in a --snp-only index);
 * haplotypes are not used (use_haplotype defaults to false, hisat2.cpp:522). */
typedef struct {
	const h2o_index* ix;
	const uint8_t* rdseq;
	uint32_t tidx, mm, maxAltsTried, numALTsTried;
	int left;
	int best_rdoff;
	h2o_edit tmp[H2O_MAX_EDITS]; uint32_t ntmp;       /* tmp_edits */
	h2o_edit* best; uint32_t* nbest;                   /* edits */
	uint32_t* numNs;
	int overflow;
	/* candidate_edits (ELList<Edit,128,4>*): alternative edit lists reaching the same best offset; NULL when cand == NULL */
	h2o_edit (*cand)[H2O_MAX_EDITS]; uint32_t* cand_n; uint32_t ncand, cand_cap;
} awa_ctx;
static void awa_cand_push(awa_ctx* x) {
	if(!x->cand) return;
	if(x->ncand >= x->cand_cap) { x->overflow = 1; return; }
	memcpy(x->cand[x->ncand], x->tmp, x->ntmp * sizeof x->tmp[0]); x->cand_n[x->ncand] = x->ntmp; x->ncand++;
}
static void awa_push_front(awa_ctx* x, h2o_edit e) { if(x->ntmp >= H2O_MAX_EDITS) { x->overflow = 1; return; } memmove(x->tmp + 1, x->tmp, x->ntmp * sizeof e); x->tmp[0] = e; x->ntmp++; }
static void awa_push_back(awa_ctx* x, h2o_edit e) { if(x->ntmp >= H2O_MAX_EDITS) { x->overflow = 1; return; } x->tmp[x->ntmp++] = e; }
static void awa_erase_front(awa_ctx* x, uint32_t n) { memmove(x->tmp, x->tmp + n, (x->ntmp - n) * sizeof x->tmp[0]); x->ntmp -= n; }
static void awa_commit(awa_ctx* x) { memcpy(x->best, x->tmp, x->ntmp * sizeof x->tmp[0]); *x->nbest = x->ntmp; }
static uint32_t alt_lobound(const h2o_index* ix, uint32_t pos) {   /* EList::bsearchLoBound with a type-NONE key */
	uint32_t lo = 0, hi = ix->nalts;
	while(lo < hi) { uint32_t m = (lo + hi) >> 1; if(ix->alts[m].pos < pos) lo = m + 1; else hi = m; }
	return lo;
}
static const uint8_t* awa_fetch(const h2o_index* ix, uint32_t tidx, int rfoff, uint32_t rflen, uint8_t* buf /* >= rflen + 64 */) {
	memset(buf, 4, rflen + 64);
	int s = rfoff > 0 ? rfoff : 0;
	int cnt = rfoff > 0 ? (int)rflen : (int)rflen + rfoff;
	if(cnt > 0) h2o_get_stretch(&ix->r, tidx, s, (uint32_t)cnt, buf + 32);
	return buf + 32 + (rfoff < 0 ? rfoff : 0);
}
#define AWA_BUF 1400
static uint32_t awa_recur(awa_ctx* x, uint32_t joinedOff, uint32_t rdoff_add, uint32_t rdoff, uint32_t rdlen,
                          const uint8_t* rfseq, int rfoff, uint32_t rflen, uint32_t tmp_numNs, uint32_t dep, uint32_t prev_alt_type)
{
	const h2o_index* ix = x->ix;
	const h2o_alt* alts = ix->alts;
	const uint8_t* rdseq = x->rdseq;
	(void)prev_alt_type;
	if(x->numALTsTried > x->maxAltsTried + dep) return 0;
	if(rfoff < -16) return 0;
	uint32_t contig_len = ix->r.refLens[x->tidx];
	if(rfoff >= (int64_t)contig_len) return 0;
	if(rfoff >= 0 && (uint64_t)rfoff + rflen > contig_len) rflen = contig_len - rfoff;
	else if(rfoff < 0 && rflen > contig_len) rflen = contig_len;
	if(rflen == 0) return 0;
	uint8_t buf[AWA_BUF], buf2[AWA_BUF];
	if(rflen + 64 > AWA_BUF) { x->overflow = 1; return 0; }
	if(rfseq == NULL) rfseq = awa_fetch(ix, x->tidx, rfoff, rflen, buf);
	if(x->left) {
		uint32_t tmp_mm = 0, mm_tmp_numNs = 0;
		int min_rd_i = (int)rdoff, mm_min_rd_i = (int)rdoff;
		for(int rf_i = (int)rflen - 1; rf_i >= 0 && mm_min_rd_i >= 0; rf_i--, mm_min_rd_i--) {
			int rf_bp = rfseq[rf_i], rd_bp = rdseq[mm_min_rd_i];
			if(rf_bp != rd_bp || rd_bp == 4) {
				if(tmp_mm == 0) min_rd_i = mm_min_rd_i;
				if(tmp_mm >= x->mm) break;
				tmp_mm++;
				h2o_edit e = { (uint32_t)mm_min_rd_i, (uint8_t)"ACGTN"[rf_bp], (uint8_t)"ACGTN"[rd_bp], H2O_EDIT_MM, 0, H2O_MAX };
				awa_push_front(x, e);
			}
			if(rf_bp == 4) { if(tmp_mm == 0) tmp_numNs++; mm_tmp_numNs++; }
		}
		if(tmp_mm == 0) min_rd_i = mm_min_rd_i;
		if(mm_min_rd_i < x->best_rdoff) { x->best_rdoff = mm_min_rd_i; awa_commit(x); if(x->numNs) *x->numNs = mm_tmp_numNs; }
		if(mm_min_rd_i < 0) return rdlen;
		if(tmp_mm > 0) { awa_erase_front(x, tmp_mm); tmp_mm = 0; }
		int a_first = 0, a_second = 0;
		if(ix->nalts > 0) {
			uint32_t rd_diff = rdoff - (uint32_t)mm_min_rd_i;
			rd_diff = rd_diff > 16 ? rd_diff - 16 : 0;
			uint32_t cpos = rd_diff >= joinedOff ? joinedOff : joinedOff - rd_diff;
			a_first = a_second = (int)alt_lobound(ix, cpos);
			if(a_first >= (int)ix->nalts) a_first = a_second = a_second - 1;
			for(; a_first >= 0; a_first--) {
				const h2o_alt* alt = &alts[a_first];
				if(alt->type == H2O_ALT_SNP_SGL || alt->type == H2O_ALT_SNP_DEL || alt->type == H2O_ALT_SNP_INS) {
					if(alt->type == H2O_ALT_SNP_DEL && !(alt->seq & 0xff)) continue;
					if((uint64_t)alt->pos + rdlen < joinedOff) break;
				} else if(alt->type == H2O_ALT_SPLICESITE) {
					if(alt->pos < alt->len) continue;
					if((uint64_t)alt->pos + rdlen - 1 < joinedOff) break;
				} else continue;
			}
		}
		const uint32_t orig_nedits = x->ntmp;
		for(; a_second > a_first; a_second--) {
			h2o_alt alt = alts[a_second];
			if(alt.pos >= joinedOff) continue;
			if(alt.type == H2O_ALT_SPLICESITE || alt.type == H2O_ALT_EXON) continue;   /* not built: splice-site ALTs */
			if(alt.type == H2O_ALT_SNP_DEL) {
				if(!(alt.seq & 0xff)) continue;
				alt.pos = alt.pos - alt.len + 1;
			}
			int alt_compatible = 0;
			int rf_i = (int)rflen - 1, rd_i = (int)rdoff, diff = 0;
			if(alt.type == H2O_ALT_SNP_SGL) diff = (int)(joinedOff - alt.pos - 1);
			else if(alt.type == H2O_ALT_SNP_DEL) {
				if(alt.pos + alt.len >= joinedOff) continue;
				diff = (int)(joinedOff - (alt.pos + alt.len));
			} else if(alt.type == H2O_ALT_SNP_INS) diff = (int)(joinedOff - alt.pos);
			else continue;
			if(rf_i < diff || rd_i < diff) continue;
			rf_i -= diff; rd_i -= diff;
			int rd_bp = rdseq[rd_i];
			if(rd_i < min_rd_i) {
				if(alt.type == H2O_ALT_SNP_INS) { if(rd_i + 1 >= min_rd_i) continue; }
				break;
			}
			if(alt.type == H2O_ALT_SNP_SGL) {
				if(rd_bp == (int)alt.seq) {
					int rf_bp = rfseq[rf_i];
					h2o_edit e = { (uint32_t)rd_i, (uint8_t)"ACGTN"[rf_bp], (uint8_t)"ACGTN"[rd_bp], H2O_EDIT_MM, 0, (uint32_t)a_second };
					awa_push_front(x, e);
					rd_i--; rf_i--;
					alt_compatible = 1;
				}
			} else if(alt.type == H2O_ALT_SNP_DEL) {
				if(rfoff + rf_i > (int)alt.len) {
					if(rf_i > (int)alt.len) {
						for(uint32_t i = 0; i < alt.len; i++) {
							int rf_bp = rfseq[rf_i - (int)i];
							h2o_edit e = { (uint32_t)(rd_i + 1), (uint8_t)"ACGTN"[rf_bp], '-', H2O_EDIT_READ_GAP, 0, (uint32_t)a_second };
							awa_push_front(x, e);
						}
					} else {                                       /* long deletions: refetch further left */
						int new_rfoff = rfoff - (int)alt.len;
						uint32_t new_rflen = (uint32_t)(rf_i + (int)alt.len + 10);
						if(new_rflen + 64 > AWA_BUF) { x->overflow = 1; return 0; }
						const uint8_t* new_rfseq = awa_fetch(ix, x->tidx, new_rfoff, new_rflen, buf2);
						for(int i = 0; i < (int)alt.len; i++) {
							int rf_bp = new_rfseq[rf_i - i + (int)alt.len];
							h2o_edit e = { (uint32_t)(rd_i + 1), (uint8_t)"ACGTN"[rf_bp], '-', H2O_EDIT_READ_GAP, 0, (uint32_t)a_second };
							awa_push_front(x, e);
						}
					}
					rf_i -= (int)alt.len;
					alt_compatible = 1;
				}
			} else {                                               /* insertion */
				if(rd_i > (int)alt.len) {
					int same_seq = 1;
					for(uint32_t i = 0; i < alt.len; i++) {
						rd_bp = rdseq[rd_i - (int)i];
						int snp_bp = (int)((alt.seq >> (i << 1)) & 3);
						if(rd_bp != snp_bp) { same_seq = 0; break; }
						h2o_edit e = { (uint32_t)(rd_i - (int)i), '-', (uint8_t)"ACGTN"[rd_bp], H2O_EDIT_REF_GAP, 0, (uint32_t)a_second };
						awa_push_front(x, e);
					}
					if(same_seq) { rd_i -= (int)alt.len; alt_compatible = 1; }
				}
			}
			if(alt_compatible) {
				x->numALTsTried++;
				if(rd_i < 0) { x->best_rdoff = rd_i; awa_commit(x); return rdlen; }
				uint32_t next_joinedOff = alt.pos;
				int next_rfoff = rfoff, next_rdoff = rd_i, next_rflen = rf_i + 1, next_rdlen = rd_i + 1;
				const uint8_t* next_rfseq = rfseq;
				if(next_rflen < next_rdlen) {
					int add_len = next_rdlen + 10 - next_rflen;
					if(next_rfoff < add_len) add_len = next_rfoff;
					next_rfoff -= add_len; next_rflen += add_len; next_rfseq = NULL;
				}
				uint32_t alignedLen = awa_recur(x, next_joinedOff, rdoff_add, (uint32_t)next_rdoff, (uint32_t)next_rdlen, next_rfseq, next_rfoff,
				                                (uint32_t)next_rflen, tmp_numNs, dep + 1, alt.type);
				if(alignedLen == (uint32_t)next_rdlen) return rdlen;
			}
			if(orig_nedits < x->ntmp) awa_erase_front(x, x->ntmp - orig_nedits);
		}
		return 0;
	} else {
		uint32_t tmp_mm = 0, max_rd_i = 0, mm_max_rd_i = 0, mm_tmp_numNs = 0;
		for(uint32_t rf_i = 0; rf_i < rflen && mm_max_rd_i < rdlen; rf_i++, mm_max_rd_i++) {
			int rf_bp = rfseq[rf_i], rd_bp = rdseq[rdoff + mm_max_rd_i];
			if(rf_bp != rd_bp || rd_bp == 4) {
				if(tmp_mm == 0) max_rd_i = mm_max_rd_i;
				if(tmp_mm >= x->mm) break;
				tmp_mm++;
				h2o_edit e = { mm_max_rd_i + rdoff_add, (uint8_t)"ACGTN"[rf_bp], (uint8_t)"ACGTN"[rd_bp], H2O_EDIT_MM, 0, H2O_MAX };
				awa_push_back(x, e);
			}
			if(rf_bp == 4) { if(tmp_mm == 0) tmp_numNs++; mm_tmp_numNs++; }
		}
		if(tmp_mm == 0) max_rd_i = mm_max_rd_i;
		if((int)(mm_max_rd_i + rdoff) > x->best_rdoff) { x->best_rdoff = (int)(mm_max_rd_i + rdoff); awa_commit(x); if(x->numNs) *x->numNs = mm_tmp_numNs; x->ncand = 0; }
		else if((int)(mm_max_rd_i + rdoff) == x->best_rdoff) awa_cand_push(x);
		if(mm_max_rd_i == rflen) return mm_max_rd_i;
		if(ix->nalts == 0) return 0;                               /* bsearchLoBound on an empty list: first >= size */
		uint32_t a_first, a_second;
		{
			uint32_t rd_diff = max_rd_i > 16 ? max_rd_i - 16 : 0;
			a_first = a_second = alt_lobound(ix, joinedOff + rd_diff);
			if(a_first >= ix->nalts) return 0;
			for(; a_second < ix->nalts; a_second++) {
				const h2o_alt* alt = &alts[a_second];
				if(alt->type == H2O_ALT_SPLICESITE) { if(alt->pos > alt->len) continue; }
				if(alt->type == H2O_ALT_SNP_DEL) { if(alt->seq & 0xff) continue; }
				if(alt->pos > joinedOff + max_rd_i) break;
			}
		}
		if(mm_max_rd_i == rdlen) return mm_max_rd_i;               /* no splice-site ALTs to look further for */
		if(tmp_mm > 0) { x->ntmp -= tmp_mm; tmp_mm = 0; }
		const uint32_t orig_nedits = x->ntmp;
		for(; a_first < a_second; a_first++) {
			const h2o_alt* alt = &alts[a_first];
			if(alt->type == H2O_ALT_SPLICESITE || alt->type == H2O_ALT_EXON) continue;
			if(alt->type == H2O_ALT_SNP_DEL) { if(alt->seq & 0xff) continue; }
			int alt_compatible = 0;
			uint32_t rf_i, rd_i;
			rf_i = rd_i = alt->pos - joinedOff;
			if(rd_i >= rdlen) continue;
			int rf_bp = rfseq[rf_i], rd_bp = rdseq[rdoff + rd_i];
			if(alt->type == H2O_ALT_SNP_SGL) {
				if(rd_bp == (int)alt->seq) {
					h2o_edit e = { rd_i + rdoff_add, (uint8_t)"ACGTN"[rf_bp], (uint8_t)"ACGTN"[rd_bp], H2O_EDIT_MM, 0, a_first };
					awa_push_back(x, e);
					rd_i++; rf_i++;
					alt_compatible = 1;
				}
			} else if(alt->type == H2O_ALT_SNP_DEL) {
				int try_del = rd_i > 0;
				if(rd_i == 0 && dep > 0) { if(x->ntmp > 0 && x->tmp[x->ntmp - 1].type != H2O_EDIT_READ_GAP) try_del = 1; }
				if(try_del) {
					if(rf_i + alt->len <= rflen) {
						for(uint32_t i = 0; i < alt->len; i++) {
							rf_bp = rfseq[rf_i + i];
							h2o_edit e = { rd_i + rdoff_add, (uint8_t)"ACGTN"[rf_bp], '-', H2O_EDIT_READ_GAP, 0, a_first };
							awa_push_back(x, e);
						}
					} else {                                       /* long deletions */
						uint32_t new_rflen = rf_i + alt->len + 10;
						if(new_rflen + 64 > AWA_BUF) { x->overflow = 1; return 0; }
						const uint8_t* new_rfseq = awa_fetch(ix, x->tidx, rfoff, new_rflen, buf2);
						for(uint32_t i = 0; i < alt->len; i++) {
							rf_bp = new_rfseq[rf_i + i];
							h2o_edit e = { rd_i + rdoff_add, (uint8_t)"ACGTN"[rf_bp], '-', H2O_EDIT_READ_GAP, 0, a_first };
							awa_push_back(x, e);
						}
					}
					rf_i += alt->len;
					alt_compatible = 1;
				}
			} else if(alt->type == H2O_ALT_SNP_INS) {
				if(rd_i + alt->len <= rdlen && rf_i > 0) {
					int same_seq = 1;
					for(uint32_t i = 0; i < alt->len; i++) {
						rd_bp = rdseq[rdoff + rd_i + i];
						int snp_bp = (int)((alt->seq >> ((alt->len - i - 1) << 1)) & 3);
						if(rd_bp != snp_bp) { same_seq = 0; break; }
						h2o_edit e = { rd_i + i + rdoff_add, '-', (uint8_t)"ACGTN"[rd_bp], H2O_EDIT_REF_GAP, 0, a_first };
						awa_push_back(x, e);
					}
					if(same_seq) { rd_i += alt->len; alt_compatible = 1; }
				}
			}
			if(alt_compatible) {
				x->numALTsTried++;
				if(rd_i == rdlen) {
					if(x->best_rdoff < (int)(rdoff + rd_i)) x->ncand = 0;
					awa_cand_push(x);
					x->best_rdoff = (int)(rdoff + rd_i); awa_commit(x); return rd_i;
				}
				uint32_t next_joinedOff = 0;
				int next_rfoff = rfoff + (int)rf_i;
				uint32_t next_rdoff = rdoff + rd_i;
				const uint8_t* next_rfseq = rfseq + rf_i;
				uint32_t next_rflen = rflen - rf_i, next_rdlen = rdlen - rd_i;
				if(alt->type == H2O_ALT_SNP_SGL) next_joinedOff = alt->pos + 1;
				else if(alt->type == H2O_ALT_SNP_DEL) { next_joinedOff = alt->pos + alt->len; if(rflen <= rf_i) next_rflen = 0; }
				else next_joinedOff = alt->pos;
				if(next_rflen < next_rdlen) { next_rflen = next_rdlen + 10; next_rfseq = NULL; }
				uint32_t alignedLen = awa_recur(x, next_joinedOff, rdoff_add + rd_i, next_rdoff, next_rdlen, next_rfseq, next_rfoff, next_rflen,
				                                tmp_numNs, dep + 1, alt->type);
				if(alignedLen > 0) { if(rd_i + alignedLen == rdlen) return rd_i + alignedLen; }
			}
			if(orig_nedits < x->ntmp) x->ntmp = orig_nedits;
		}
		return 0;
	}
}
/* alignWithALTs hi_aligner.h:683-783 */
static uint32_t align_with_alts_c(const h2o_index* ix, uint32_t joinedOff, const uint8_t* rdseq, uint32_t base_rdoff,
                                  uint32_t rdoff, uint32_t rdlen, uint32_t tidx, int rfoff, uint32_t rflen, int left,
                                  h2o_edit* edits, uint32_t* nedits_io, uint32_t mm, uint32_t* numNs,
                                  h2o_edit (*cand)[H2O_MAX_EDITS], uint32_t* cand_n, uint32_t cand_cap, uint32_t* ncand);
static uint32_t align_with_alts(const h2o_index* ix, uint32_t joinedOff, const uint8_t* rdseq, uint32_t base_rdoff,
                                uint32_t rdoff, uint32_t rdlen, uint32_t tidx, int rfoff, uint32_t rflen, int left,
                                h2o_edit* edits, uint32_t* nedits_io, uint32_t mm, uint32_t* numNs)
{
	return align_with_alts_c(ix, joinedOff, rdseq, base_rdoff, rdoff, rdlen, tidx, rfoff, rflen, left, edits, nedits_io, mm, numNs, NULL, NULL, 0, NULL);
}
static uint32_t align_with_alts_c(const h2o_index* ix, uint32_t joinedOff, const uint8_t* rdseq, uint32_t base_rdoff,
                                  uint32_t rdoff, uint32_t rdlen, uint32_t tidx, int rfoff, uint32_t rflen, int left,
                                  h2o_edit* edits, uint32_t* nedits_io, uint32_t mm, uint32_t* numNs,
                                  h2o_edit (*cand)[H2O_MAX_EDITS], uint32_t* cand_n, uint32_t cand_cap, uint32_t* ncand)
{
	awa_ctx x;
	x.cand = cand; x.cand_n = cand_n; x.cand_cap = cand_cap; x.ncand = 0;
	x.ix = ix; x.rdseq = rdseq; x.tidx = tidx; x.mm = mm; x.maxAltsTried = 16; x.numALTsTried = 0; x.left = left;
	x.best_rdoff = (int)rdoff; x.best = edits; x.nbest = nedits_io; x.numNs = numNs; x.overflow = 0;
	if(numNs) *numNs = 0;
	const uint32_t nedits = *nedits_io;
	x.ntmp = nedits;
	memcpy(x.tmp, edits, nedits * sizeof x.tmp[0]);
	awa_recur(&x, joinedOff, rdoff - base_rdoff, rdoff, rdlen, NULL, rfoff, rflen, 0, 0, 0);
	if(ncand) *ncand = x.ncand;
	uint32_t extlen = left ? rdoff - (uint32_t)x.best_rdoff : (uint32_t)x.best_rdoff - rdoff;
	uint32_t ne = *nedits_io;
	if(extlen > 0 && ne > 0) {                                            /* :751-779 */
		const h2o_edit* f = &edits[0];
		if(f->pos + extlen == base_rdoff + 1) {
			if(f->type == H2O_EDIT_READ_GAP || f->type == H2O_EDIT_REF_GAP) extlen = 0;
			if(f->type == H2O_EDIT_MM && f->chr == 'N') extlen = 0;
		}
		const h2o_edit* b = &edits[ne - 1];
		if(extlen > 0 && b->pos == rdoff - base_rdoff + extlen - 1) {
			if(b->type == H2O_EDIT_READ_GAP || b->type == H2O_EDIT_REF_GAP) extlen = 0;
		}
		if(extlen == 0 && ne > nedits) {
			if(left) memmove(edits, edits + (ne - nedits), nedits * sizeof *edits);
			*nedits_io = nedits;
		}
	}
	return extlen;
}

/* findOffDiffs hi_aligner.h:2545-2640: offset differences that indel ALTs near [start, end) can introduce */
typedef struct { uint32_t first; int second; } offdiff_t;
static int offdiff_lt(offdiff_t a, offdiff_t b) { return a.first != b.first ? a.first < b.first : a.second < b.second; }
static uint32_t find_off_diffs(const h2o_index* ix, uint32_t start, uint32_t end, offdiff_t* od, uint32_t cap, uint32_t* nod) {
	uint32_t n = 0;
	od[n].first = 0; od[n].second = 0; n++;
	*nod = n;
	if(ix->g.p.linear) return n;
	const h2o_alt* alts = ix->alts;
	uint32_t a1 = alt_lobound(ix, start), a2 = a1;
	for(; a2 < ix->nalts; a2++) {
		const h2o_alt* alt = &alts[a2];
		if(alt->type == H2O_ALT_SPLICESITE && alt->pos > alt->len) continue;
		if(alt->type == H2O_ALT_SNP_DEL && (alt->seq & 0xff)) continue;
		if(alt->pos >= end) break;
	}
	if(a1 >= a2) return n;
#define IS_GAPALT(a) (((a)->type == H2O_ALT_SNP_DEL && !((a)->seq & 0xff)) || (a)->type == H2O_ALT_SNP_INS)
	for(uint32_t s2 = a2; s2 > a1; s2--) {
		const h2o_alt* alt = &alts[s2 - 1];
		if(!IS_GAPALT(alt)) continue;
		int off = alt->type == H2O_ALT_SNP_DEL ? (int)alt->len : -(int)alt->len;
		if(n < cap) { od[n].first = (uint32_t)(off < 0 ? -off : off); od[n].second = off > 0 ? 1 : -1; n++; }
	}
	if(n > 1) {                                            /* sort + unique */
		for(uint32_t i = 1; i < n; i++) { offdiff_t x = od[i]; int j = (int)i - 1; while(j >= 0 && offdiff_lt(x, od[j])) { od[j + 1] = od[j]; j--; } od[j + 1] = x; }
		uint32_t w = 1;
		for(uint32_t i = 1; i < n; i++) if(od[i].first != od[w - 1].first || od[i].second != od[w - 1].second) od[w++] = od[i];
		n = w;
	}
	const uint32_t single = n;
	for(uint32_t s2 = a2; s2 > a1; s2--) {
		const h2o_alt* alt = &alts[s2 - 1];
		if(!IS_GAPALT(alt)) continue;
		int off = alt->type == H2O_ALT_SNP_DEL ? (int)alt->len : -(int)alt->len;
		for(uint32_t s3 = s2 - 1; s3 > a1; s3--) {
			const h2o_alt* alt2 = &alts[s3 - 1];
			if(!IS_GAPALT(alt2)) continue;
			if(alt2->type == H2O_ALT_SNP_DEL) { if(alt2->pos + alt2->len >= alt->pos) continue; off += (int)alt2->len; }
			else { if(alt2->pos >= alt->pos) continue; off -= (int)alt2->len; }
			int found = 0;
			for(uint32_t i = 0; i < n; i++) if(off == (int)od[i].first * od[i].second) { found = 1; break; }
			if(!found && n < cap) { od[n].first = (uint32_t)(off < 0 ? -off : off); od[n].second = off > 0 ? 1 : -1; n++; }
		}
	}
#undef IS_GAPALT
	*nod = n;
	return single;
}
static int ghit_equal(const h2o_ghit* a, const h2o_ghit* b) { /* GenomeHit::operator== hi_aligner.h:1156-1183 */
	if(a->fw != b->fw || a->rdoff != b->rdoff || a->len != b->len || a->tidx != b->tidx || a->toff != b->toff || a->trim5 != b->trim5 ||
	   a->trim3 != b->trim3 || a->nedits != b->nedits) return 0;
	for(uint32_t i = 0; i < a->nedits; i++) {
		const h2o_edit *e = &a->edits[i], *o = &b->edits[i];
		if(e->type == H2O_EDIT_READ_GAP) { if(o->type != H2O_EDIT_READ_GAP) return 0; }
		else if(e->type == H2O_EDIT_REF_GAP) { if(o->type != H2O_EDIT_REF_GAP) return 0; }
		else if(e->type != o->type || e->pos != o->pos || e->chr != o->chr || e->qchr != o->qchr) return 0;
	}
	return 1;
}
/* static GenomeHit::adjustWithALT hi_aligner.h:2239-2390 (no splice-site ALTs: findSSOffs yields the single (0,0)).
 * Appends to hits[*nhits..cap); returns whether any hit was added. */
int h2o_adjust_with_alt(const h2o_index* ix, const uint8_t* seq, int fw, uint32_t rdoff, uint32_t len, uint32_t tidx, uint32_t toff,
                        uint32_t joinedOff, h2o_ghit* hits, uint32_t* nhits, uint32_t cap)
{
	const uint32_t n0 = *nhits;
	if(*nhits >= cap) return 0;
	h2o_ghit* gh = &hits[*nhits];
	memset(gh, 0, sizeof *gh);
	gh->fw = (uint32_t)fw; gh->rdoff = rdoff; gh->len = len; gh->tidx = tidx; gh->toff = toff; gh->joinedOff = joinedOff;
	if(ix->g.p.linear) { (*nhits)++; return 1; }
	(*nhits)++;
	const uint32_t width = 1u << (ix->g.p.offRate + 2);
	offdiff_t od[64]; uint32_t nod = 0;
	const uint32_t single = find_off_diffs(ix, gh->joinedOff >= width ? gh->joinedOff - width : 0, gh->joinedOff + width, od, 64, &nod);
	const uint32_t max_od = 4;                              /* max(4, maxAltsTried / 4) */
	if(nod - single > max_od) nod = single + max_od;
	const uint32_t orig_joinedOff = gh->joinedOff, orig_toff = gh->toff;
	int found2 = 0;
	static h2o_edit cand[8][H2O_MAX_EDITS]; uint32_t cand_n[8], ncand = 0;
	for(uint32_t o = 0; o < nod && !found2; o++) {
		if(od[o].second >= 0) { gh->joinedOff = orig_joinedOff + od[o].first; gh->toff = orig_toff + od[o].first; }
		else { if(orig_toff < od[o].first) continue; gh->joinedOff = orig_joinedOff - od[o].first; gh->toff = orig_toff - od[o].first; }
		gh->nedits = 0;
		ncand = 0;
		uint32_t alignedLen = align_with_alts_c(ix, gh->joinedOff, seq, gh->rdoff, gh->rdoff, gh->len, gh->tidx, (int)gh->toff, gh->len + 10, 0,
		                                        gh->edits, &gh->nedits, 0, NULL, cand, cand_n, 8, &ncand);
		if(alignedLen == gh->len) {
			found2 = 1;
			for(uint32_t i = 0; i + 1 < *nhits; i++) if(ghit_equal(&hits[i], gh)) found2 = 0;
			if(found2) {
				for(uint32_t e = 0; e < ncand; e++) {
					if(*nhits >= cap) break;
					h2o_ghit* c = &hits[*nhits];
					*c = hits[*nhits - 1];
					memcpy(c->edits, cand[e], cand_n[e] * sizeof(h2o_edit)); c->nedits = cand_n[e];
					(*nhits)++;
					for(uint32_t i = 0; i + 1 < *nhits; i++) if(ghit_equal(&hits[i], c)) { (*nhits)--; break; }
				}
			}
		} else gh->nedits = 0;
	}
	if(!found2) (*nhits)--;                                 /* genomeHits.pop_back() */
	return *nhits > n0;
}

int h2o_extend(const h2o_index* ix, const h2o_scoring* sc, const uint8_t* seq, const char* qual, uint32_t rdlen,
               h2o_ghit* h, uint32_t* leftext, uint32_t* rightext, uint32_t mm)
{ /* GenomeHit::extend hi_aligner.h:2031-2232 */
	uint32_t max_leftext = *leftext, max_rightext = *rightext;
	*leftext = 0; *rightext = 0;
	if(max_leftext > 0 && h->rdoff > 0) {
		if(h->toff <= 0) return 0;
		int rl = (int)h->toff - (int)h->rdoff;
		uint32_t reflen = h->rdoff + 10;
		rl -= (int)(reflen - h->rdoff);
		if(rl < 0) { reflen += rl; rl = 0; }
		uint32_t numNs = 0, num_prev = h->nedits;
		uint32_t best_ext = (ix->nalts ? align_with_alts : align_no_alts)(ix, h->joinedOff, seq, h->rdoff - 1, h->rdoff - 1, h->rdoff, h->tidx,
		                                  rl, reflen, 1, h->edits, &h->nedits, mm, &numNs);
		if(h->len == 0 && mm == 0 && h->nedits > 0) { h->nedits = 0; return 0; }
		if(best_ext > 0) {
			*leftext = best_ext;
			uint32_t added = h->nedits - num_prev;
			int ref_ext = (int)best_ext;
			for(uint32_t i = 0; i < added; i++) {
				if(h->edits[i].type == H2O_EDIT_REF_GAP) ref_ext--;
				else if(h->edits[i].type == H2O_EDIT_READ_GAP) ref_ext++;
			}
			h->rdoff -= best_ext;
			h->toff -= ref_ext;
			h->len += best_ext;
			h->joinedOff -= (ref_ext - (int)numNs);
			for(uint32_t i = 0; i < h->nedits; i++) {
				if(i < added) h->edits[i].pos -= h->rdoff;
				else h->edits[i].pos += best_ext;
			}
		}
	}
	if(max_rightext > 0 && h->rdoff + h->len < rdlen) {
		/* getRight :962-1000 without gap/splice edits: whole hit */
		uint32_t right_rdoff = h->rdoff, right_len = h->len, right_toff = h->toff;
		for(int i = (int)h->nedits - 1; i >= 0; i--) {
			const h2o_edit* e = &h->edits[i];
			if(e->type == H2O_EDIT_READ_GAP || e->type == H2O_EDIT_REF_GAP) {
				right_rdoff = h->rdoff + e->pos;
				right_len = h->len - e->pos;
				if(e->type == H2O_EDIT_REF_GAP) { right_rdoff++; right_len--; }
				uint32_t roff = h->toff + h->len; /* getRightOff :1020-1035 */
				for(uint32_t k = 0; k < h->nedits; k++) {
					if(h->edits[k].type == H2O_EDIT_READ_GAP) roff++;
					else if(h->edits[k].type == H2O_EDIT_REF_GAP) roff--;
				}
				right_toff = roff - right_len;
				break;
			}
		}
		uint32_t rl = right_toff + right_len;
		uint32_t rr = rdlen - (right_rdoff + right_len);
		uint32_t tlen = ix->r.refLens[h->tidx];
		if(rl < tlen) {
			uint32_t reflen = rr + 10;
			if(rl + reflen > tlen) reflen = tlen - rl;
			int ref_ext = (int)h->len;
			for(uint32_t ei = 0; ei < h->nedits; ei++) {
				const h2o_edit* e = &h->edits[ei];
				if(e->type == H2O_EDIT_REF_GAP) ref_ext--;
				else if(e->type == H2O_EDIT_READ_GAP) ref_ext++;
				else if(e->type == H2O_EDIT_MM && e->chr == 'N') ref_ext--;
			}
			uint32_t best_ext = (ix->nalts ? align_with_alts : align_no_alts)(ix, h->joinedOff + ref_ext, seq, h->rdoff, h->rdoff + h->len,
			                                  rdlen - (h->rdoff + h->len), h->tidx, (int)rl, reflen, 0,
			                                  h->edits, &h->nedits, mm, NULL);
			if(h->len == 0 && mm == 0 && h->nedits > 0) { h->nedits = 0; return 0; }
			if(best_ext > 0) { *rightext = best_ext; h->len += best_ext; }
		}
	}
	h2o_calculate_score(sc, qual, h);
	return *leftext > 0 || *rightext > 0;
}

/* ------------------------------------------------------------------ batch baseline */
uint64_t h2o_seed_extend_batch(const h2o_index* ix, const uint8_t* seqs, const uint32_t* offs, uint32_t nreads,
                               int pseudogeneStop, uint32_t khits, uint64_t* counters)
{
	h2o_scoring sc;
	h2o_scoring_default(&sc);
	uint64_t sum = 0, nrank = 0, nsteps = 0, next = 0;
	uint8_t rc[1024];
	char qual[1024];
	memset(qual, 'I', sizeof qual);
	for(uint32_t r = 0; r < nreads; r++) {
		const uint8_t* fwseq = seqs + offs[r];
		uint32_t len = offs[r + 1] - offs[r];
		if(len > 1000) continue;
		for(uint32_t i = 0; i < len; i++) { uint8_t c = fwseq[len - 1 - i]; rc[i] = c < 4 ? 3 - c : 4; }
		for(int fwi = 0; fwi < 2; fwi++) {
			const uint8_t* seq = fwi == 0 ? fwseq : rc;
			h2o_bwthit bh;
			h2o_partial_search(ix, seq, len, 0, pseudogeneStop, 1, khits, &bh);
			nrank += bh.nrank;
			sum = sum * 1000003u + bh.top + 31 * bh.bot + 977 * bh.len + bh.hit_type;
			if(bh.top == H2O_MAX || bh.bot - bh.top > 16 || bh.len <= ix->minK + 2) continue;
			h2o_coord co[16];
			uint32_t nco = 0, st = 0;
			int straddled = 0;
			h2o_genome_coords(ix, bh.top, bh.bot, bh.bot - bh.top, bh.len, 0, co, &nco, &straddled, &st);
			nsteps += st;
			for(uint32_t k = 0; k < nco; k++) {
				sum = sum * 1000003u + co[k].tidx + 7 * co[k].toff;
				if(co[k].tidx == H2O_MAX) continue;
				h2o_ghit gh;
				memset(&gh, 0, sizeof gh);
				gh.fw = fwi == 0; gh.rdoff = len - bh.bwoff - bh.len; gh.len = bh.len;
				gh.tidx = co[k].tidx; gh.toff = co[k].toff; gh.joinedOff = co[k].joinedOff;
				uint32_t le = H2O_MAX, re = H2O_MAX;
				h2o_extend(ix, &sc, seq, qual, len, &gh, &le, &re, 0);
				next++;
				sum = sum * 1000003u + gh.rdoff + 3 * gh.len + 5 * gh.toff;
			}
		}
	}
	if(counters) { counters[0] = nrank; counters[1] = nsteps; counters[2] = next; }
	return sum;
}

/* ------------------------------------------------------------------ Smith-Waterman (a23-a25) */
/* The end-to-end DP of SwAligner as HISAT2 calls it from hybridSearch (spliced_aligner.h:209-262), 8-bit and (minsc < -254,
 * alignNucleotidesEnd2EndSseI16 aligner_swsse_ee_i16.cpp:793-1170 with its gather :1216 and backtrace :1324) 16-bit cells:
 *   frameSeedExtensionRect dp_framer.cpp:81-130; initRef aligner_sw.cpp:137-253;
 *   alignNucleotidesEnd2EndSseU8 aligner_swsse_ee_u8.cpp:791-1172 (Farrar striped fill + lazy-F fix-up; the
 *   H/E/F bytes it leaves in SSEMatrix equal the plain saturating recurrences below);
 *   gatherCellsNucleotidesEnd2EndSseU8 :1202-1234; SwAligner::nextAlignment aligner_sw.cpp:709-870;
 *   backtraceNucleotidesEnd2EndSseU8 :1309-1900 (tie-breaks are the deterministic `#if 1` branches). */
static inline uint32_t subsu(uint32_t a, uint32_t b) { return a > b ? a - b : 0; }   /* unsigned saturating subtract (cells are <= TOP) */
static inline uint32_t maxu(uint32_t a, uint32_t b) { return a > b ? a : b; }
static uint32_t lcg_next(uint32_t* last) { /* RandomSource::nextU32 random_source.h:52-61 */
	*last = 1664525u * *last + 1013904223u;
	uint32_t ret = *last >> 16;
	*last = 1664525u * *last + 1013904223u;
	return ret ^ *last;
}
static int mmpen_q(const h2o_scoring* sc, int q) { /* Scoring::initPens COST_MODEL_QUAL scoring.h:117-124 */
	if(q < 0) q = 0;
	int ii = q < 40 ? q : 40;
	float frac = (float)ii / 40.0f;
	return sc->mmpMin + (int)(frac * (float)(sc->mmpMax - sc->mmpMin));
}
static const char MASK2DNA[] = "?ACMGRSVTWYHKDBNN";   /* alphabet.cpp:71-89 */

/* fill + gather + first nextAlignment over the rectangle whose columns are rf[0..ncol) (0..3, 4 = N; owned: freed here), column 0 at reference
 * offset o->refl, `triml` columns trimmed on the left, core diagonals o->corel..o->corer; *ns_out = Ns of the reported alignment */
static int sw_rect(const h2o_scoring* sc, const uint8_t* seq, const char* qual, uint32_t nrow, uint8_t* rf, uint32_t ncol, int64_t triml,
                   int64_t minsc, int nceil, int gapbar, uint32_t* rnd, h2o_sw_result* o, int* ns_out);

int h2o_sw_align(const h2o_index* ix, const h2o_scoring* sc, const uint8_t* seq, const char* qual, uint32_t rdlen,
                 uint32_t tidx, uint32_t refoff, int64_t minsc, int nceil, int gapbar, uint32_t* rnd, h2o_sw_result* o)
{
	memset(o, 0, sizeof *o);
	o->best = -99999;
	/* frame: maxgap = min(max(10, 10), 10); maxns = 0 => trim to the reference */
	const int64_t maxgap = 10, reflen = ix->r.refLens[tidx];
	int64_t refl = (int64_t)refoff - 2 * maxgap, refr = (int64_t)refoff + (rdlen - 1) + 2 * maxgap;
	int64_t triml = 0, trimr = 0;
	if(refr >= reflen) trimr = refr - (reflen - 1);
	if(refl < 0) triml = -refl;
	o->refl = refl + triml; o->refr = refr - trimr; o->refl_pretrim = refl; o->refr_pretrim = refr;
	o->corel = maxgap; o->corer = maxgap + 2 * maxgap;
	const uint32_t ncol = (uint32_t)(o->refr - o->refl + 1);
	uint8_t* rf = (uint8_t*)malloc(ncol + 1);
	h2o_get_stretch(&ix->r, tidx, o->refl, ncol, rf);       /* 0..3, 4 = N / outside */
	int ns;
	return sw_rect(sc, seq, qual, rdlen, rf, ncol, triml, minsc, nceil, gapbar, rnd, o, &ns);
}

/* The same DP over a window of a plain reference string: columns refl..refr of ref[0..reflen) (codes 0..4), N outside the string (the padding
 * of the reference's own SwAligner test driver, aligner_sw.cpp:1230-1245), nothing trimmed, core diagonals corel..corer (col - row, as DPRect
 * counts them).  For the known-answer cases of aligner_sw.cpp:1470-2727 (tests/golden/sw_kat.json). */
int h2o_sw_align_window(const h2o_scoring* sc, const uint8_t* seq, const char* qual, uint32_t rdlen, const uint8_t* ref, uint32_t reflen,
                        int64_t refl, int64_t refr, int64_t corel, int64_t corer, int64_t minsc, int nceil, int gapbar, uint32_t* rnd,
                        h2o_sw_result* o, int* ns_out)
{
	memset(o, 0, sizeof *o);
	o->best = -99999;
	o->refl = o->refl_pretrim = refl; o->refr = o->refr_pretrim = refr;
	o->corel = corel; o->corer = corer;
	const uint32_t ncol = (uint32_t)(refr - refl + 1);
	uint8_t* rf = (uint8_t*)malloc(ncol + 1);
	for(uint32_t j = 0; j < ncol; j++) { const int64_t p = refl + (int64_t)j; rf[j] = (p < 0 || p >= (int64_t)reflen) ? 4 : ref[p]; }
	return sw_rect(sc, seq, qual, rdlen, rf, ncol, 0, minsc, nceil, gapbar, rnd, o, ns_out);
}

static int sw_rect(const h2o_scoring* sc, const uint8_t* seq, const char* qual, uint32_t nrow, uint8_t* rf, uint32_t ncol, int64_t triml,
                   int64_t minsc, int nceil, int gapbar, uint32_t* rnd, h2o_sw_result* o, int* ns_out)
{
	const int64_t rfi = o->refl;
	*ns_out = 0;
	/* cell width: SwAligner::align takes the 8-bit fill when minsc >= -254, else the 16-bit one (aligner_sw.cpp:496;
	 * alignNucleotidesEnd2EndSseI16 aligner_swsse_ee_i16.cpp + its gather / backtrace).  The i16 cells (signed saturating, 0x7fff = score 0,
	 * 0x8000 = "minus infinity", the gap barrier added twice = forced to 0x8000) are the u8 recurrences at 16 bits: with the cell + 0x8000
	 * read as unsigned, TOP = 0xffff is score 0 and every subtraction saturates at 0 */
	const uint32_t TOP = minsc >= -254 ? 0xffu : 0xffffu;
	uint16_t* H = (uint16_t*)calloc((size_t)nrow * ncol, 2);
	uint16_t* E = (uint16_t*)calloc((size_t)nrow * ncol, 2);
	uint16_t* F = (uint16_t*)calloc((size_t)nrow * ncol, 2);
	uint16_t* M = (uint16_t*)calloc((size_t)nrow * ncol, 2);
	const uint32_t rdgapo = (uint32_t)(sc->rdGapConst + sc->rdGapLinear), rdgape = (uint32_t)sc->rdGapLinear;
	const uint32_t rfgapo = (uint32_t)(sc->rfGapConst + sc->rfGapLinear), rfgape = (uint32_t)sc->rfGapLinear;
#define AT(m, i, j) m[(size_t)(i) * ncol + (j)]
	uint32_t lrmax = 0;
	for(uint32_t j = 0; j < ncol; j++) {
		const int refc = rf[j];
		for(uint32_t i = 0; i < nrow; i++) {
			const uint32_t gb = (i < (uint32_t)gapbar || (nrow - i - 1) < (uint32_t)gapbar) ? TOP : 0;
			const int readc = seq[i], q = (qual ? qual[i] : 'I') - 33;
			uint32_t pen;                                   /* query profile :76-147 == -Scoring::score scoring.h:259 */
			if(readc > 3 || refc > 3) pen = (uint32_t)sc->nPen;
			else pen = readc == refc ? 0 : (uint32_t)mmpen_q(sc, q);
			const uint32_t e = j == 0 ? 0 : maxu(subsu(AT(E, i, j - 1), rdgape), subsu(subsu(AT(H, i, j - 1), rdgapo), gb));
			const uint32_t f = i == 0 ? 0 : subsu(maxu(subsu(AT(F, i - 1, j), rfgape), subsu(AT(H, i - 1, j), rfgapo)), gb);
			const uint32_t diag = i == 0 ? TOP : (j == 0 ? 0 : AT(H, i - 1, j - 1));
			AT(E, i, j) = (uint16_t)e;
			AT(F, i, j) = (uint16_t)f;
			AT(H, i, j) = (uint16_t)maxu(maxu(subsu(diag, pen), e), f);
		}
		if(AT(H, nrow - 1, j) > lrmax) lrmax = AT(H, nrow - 1, j);
	}
	int64_t best = (int64_t)lrmax - (int64_t)TOP;
	o->best = best;
	int found = !(best < minsc) && lrmax != 0;
	o->found_align = 0;
	if(found) {
		/* gather :1202-1234 + sort (DpBtCandidate::operator< aligner_sw_nuc.h:149: score desc, row desc, col desc) */
		uint32_t ncand = 0;
		uint32_t* cand = (uint32_t*)malloc(4 * ncol);
		for(uint32_t j = 0; j < ncol; j++) if((int64_t)AT(H, nrow - 1, j) - (int64_t)TOP >= minsc) cand[ncand++] = j;
		for(uint32_t a = 1; a < ncand; a++) {            /* insertion sort: score desc, then col desc */
			uint32_t c = cand[a]; int b = (int)a - 1;
			while(b >= 0 && (AT(H, nrow - 1, cand[b]) < AT(H, nrow - 1, c) ||
			                 (AT(H, nrow - 1, cand[b]) == AT(H, nrow - 1, c) && cand[b] < c))) { cand[b + 1] = cand[b]; b--; }
			cand[b + 1] = c;
		}
		o->found_align = ncand > 0;
		/* nextAlignment aligner_sw.cpp:709-870 */
		typedef struct { uint32_t nedsz, celsz, row, col, gaps, readGaps, refGaps; int64_t score; int ns, ct; } frame_t;
		frame_t* stack = (frame_t*)malloc(sizeof(frame_t) * 4096);
		uint32_t* cells = (uint32_t*)malloc(8 * 4096);
		for(uint32_t ci = 0; ci < ncand && !o->found; ci++) {
			uint32_t row = nrow - 1, col = cand[ci];
			const int64_t escore = (int64_t)AT(H, row, col) - (int64_t)TOP;
			if(escore < minsc) continue;
			if(AT(M, row, col) & 1) continue;                /* BT_CAND_FATE_FILT_START */
			uint32_t reseed = lcg_next(rnd) + 1;
			*rnd = reseed;
			/* ---- backtrace */
			uint32_t nstack = 0, ncells = 0, ned = 0, gaps = 0, readGaps = 0, refGaps = 0;
			int64_t score = 0; int ns = 0;
			const uint32_t origCol = col;
			int ct = 0;                                      /* 0 = H, 1 = E, 2 = F */
			int ok = 0, fail = 0;
			while(1) {
				const int readc = seq[row];
				const int refm = 1 << rf[col];
				int empty = 0, reportedThru, canMoveThru = 1, branch = 0, cur = -1;
				uint16_t* mk = &AT(M, row, col);
				reportedThru = (*mk & 1) != 0;
				if(reportedThru) canMoveThru = 0;
				else if(row > 0) {
					const int gapsAllowed = !(row < (uint32_t)gapbar || (nrow - row - 1) < (uint32_t)gapbar);
					if(ct == 1) {                            /* E: came from the left */
						const int64_t sc_cur = (int64_t)AT(E, row, col) - (int64_t)TOP;
						int mask = 0, origMask;
						const int64_t sc_h_left = (int64_t)AT(H, row, col - 1) - (int64_t)TOP, sc_e_left = (int64_t)AT(E, row, col - 1) - (int64_t)TOP;
						if(sc_h_left - rdgapo == sc_cur) mask |= 1;
						if(sc_e_left - rdgape == sc_cur) mask |= 2;
						origMask = mask;
						if(*mk & (1 << 7)) mask = (*mk >> 8) & 3;
#define EMASK(v) (*mk = (uint16_t)((*mk & ~(7 << 7)) | (1 << 7) | ((v) << 8)))
#define FMASK(v) (*mk = (uint16_t)((*mk & ~(7 << 10)) | (1 << 10) | ((v) << 11)))
#define HMASK(v) (*mk = (uint16_t)((*mk & ~(31 << 1)) | (1 << 1) | ((v) << 2)))
						if(mask == 3) { cur = 3 /* READ_OPEN */; EMASK(2); branch = 1; }
						else if(mask == 2) { cur = 4 /* RDGAP_EXTEND */; EMASK(0); }
						else if(mask == 1) { cur = 3; EMASK(0); }
						else { empty = 1; canMoveThru = (origMask == 0); }
					} else if(ct == 2) {                     /* F: came from above */
						const int64_t sc_h_up = (int64_t)AT(H, row - 1, col) - (int64_t)TOP, sc_f_up = (int64_t)AT(F, row - 1, col) - (int64_t)TOP;
						const int64_t sc_cur = (int64_t)AT(F, row, col) - (int64_t)TOP;
						int mask = 0, origMask;
						if(sc_h_up - rfgapo == sc_cur) mask |= 1;
						if(sc_f_up - rfgape == sc_cur) mask |= 2;
						origMask = mask;
						if(*mk & (1 << 10)) mask = (*mk >> 11) & 3;
						if(mask == 3) { cur = 1 /* REF_OPEN */; FMASK(2); branch = 1; }
						else if(mask == 2) { cur = 2 /* RFGAP_EXTEND */; FMASK(0); }
						else if(mask == 1) { cur = 1; FMASK(0); }
						else { empty = 1; canMoveThru = (origMask == 0); }
					} else {
						const int64_t sc_cur = (int64_t)AT(H, row, col) - (int64_t)TOP;
						const int64_t sc_f_up = (int64_t)AT(F, row - 1, col) - (int64_t)TOP, sc_h_up = (int64_t)AT(H, row - 1, col) - (int64_t)TOP;
						const int hasl = col > 0;
						const int64_t sc_h_left = hasl ? (int64_t)AT(H, row, col - 1) - (int64_t)TOP : 0;
						const int64_t sc_e_left = hasl ? (int64_t)AT(E, row, col - 1) - (int64_t)TOP : 0;
						const int64_t sc_h_upleft = hasl ? (int64_t)AT(H, row - 1, col - 1) - (int64_t)TOP : 0;
						const int q = (qual ? qual[row] : 'I') - 33;
						int64_t sc_diag;                         /* Scoring::score(readc, refm, q) */
						if(readc > 3 || refm > 15) sc_diag = -sc->nPen;
						else sc_diag = (refm & (1 << readc)) ? 0 : -mmpen_q(sc, q);
						int mask = 0, origMask;
						if(gapsAllowed) {
							if(sc_cur == sc_h_up - rfgapo) mask |= 1;
							if(hasl && sc_cur == sc_h_left - rdgapo) mask |= 2;
							if(sc_cur == sc_f_up - rfgape) mask |= 4;
							if(hasl && sc_cur == sc_e_left - rdgape) mask |= 8;
						}
						if(hasl && sc_cur == sc_h_upleft + sc_diag) mask |= 16;
						origMask = mask;
						if(*mk & (1 << 1)) mask = (*mk >> 2) & 31;
						const int opts = __builtin_popcount((unsigned)mask);
						int select = -1;
						if(opts == 1) { select = __builtin_ctz((unsigned)mask); HMASK(0); }
						else if(opts > 1) {
							if(mask & 16) select = 4; else if(mask & 1) select = 0; else if(mask & 4) select = 2;
							else if(mask & 2) select = 1; else select = 3;
							mask &= ~(1 << select);
							HMASK(mask);
							branch = 1;
						}
						if(select == 4) cur = 0; else if(select == 0) cur = 1; else if(select == 1) cur = 3;
						else if(select == 2) cur = 2; else if(select == 3) cur = 4;
						else { empty = 1; canMoveThru = (origMask == 0); }
					}
				}
				*mk |= 1;                                    /* setReportedThrough */
				if(!canMoveThru) {
					if(nstack > 0) {
						frame_t* fr = &stack[--nstack];
						ncells = fr->celsz; ned = fr->nedsz; row = fr->row; col = fr->col; gaps = fr->gaps;
						readGaps = fr->readGaps; refGaps = fr->refGaps; score = fr->score; ns = fr->ns; ct = fr->ct;
						continue;
					}
					fail = 1; break;
				}
				if(empty || row == 0) { cells[2 * ncells] = row; cells[2 * ncells + 1] = col; ncells++; ok = 1; break; }
				if(branch) {
					frame_t* fr = &stack[nstack++];
					fr->nedsz = ned; fr->celsz = ncells; fr->row = row; fr->col = col; fr->gaps = gaps; fr->readGaps = readGaps;
					fr->refGaps = refGaps; fr->score = score; fr->ns = ns; fr->ct = ct;
				}
				cells[2 * ncells] = row; cells[2 * ncells + 1] = col; ncells++;
				h2o_edit* e = &o->edits[ned < H2O_MAX_EDITS ? ned : H2O_MAX_EDITS - 1];
				switch(cur) {
				case 0: {                                    /* SW_BT_OALL_DIAG */
					const int m = (refm >= 16 || readc > 3) ? -1 : ((refm >> readc) & 1);
					ct = 0;
					if(m != 1) {
						e->pos = row; e->chr = (uint8_t)MASK2DNA[refm]; e->qchr = (uint8_t)"ACGTN"[readc]; e->type = H2O_EDIT_MM; e->pad = 0; e->snp = H2O_MAX; ned++;
						const int q = (qual ? qual[row] : 'I') - 33;
						score -= (readc > 3 || refm > 15) ? sc->nPen : mmpen_q(sc, q);
					}
					if(m == -1) ns++;
					row--; col--;
					break; }
				case 1: case 2:                              /* REF_OPEN / RFGAP_EXTEND: move up */
					e->pos = row; e->chr = '-'; e->qchr = (uint8_t)"ACGTN"[readc]; e->type = H2O_EDIT_REF_GAP; e->pad = 0; e->snp = H2O_MAX; ned++;
					row--; ct = cur == 1 ? 0 : 2; score -= cur == 1 ? rfgapo : rfgape; gaps++; refGaps++;
					break;
				default:                                     /* READ_OPEN / RDGAP_EXTEND: move left */
					e->pos = row + 1; e->chr = (uint8_t)MASK2DNA[refm]; e->qchr = '-'; e->type = H2O_EDIT_READ_GAP; e->pad = 0; e->snp = H2O_MAX; ned++;
					col--; ct = cur == 3 ? 0 : 1; score -= cur == 3 ? rdgapo : rdgape; gaps++; readGaps++;
					break;
				}
			}
			if(ok) {                                         /* :1770-1850 */
				int overlapped = 0;
				for(uint32_t k = 0; k < ncells && !overlapped; k++) {
					int64_t diagi = (int64_t)cells[2 * k + 1] - (int64_t)cells[2 * k] + triml;
					if(diagi >= 0 && diagi >= o->corel && diagi <= o->corer) overlapped = 1;
				}
				if(!overlapped) ok = 0;
			}
			if(ok) {
				const int readc = seq[row], refm = 1 << rf[col];
				const int m = (refm >= 16 || readc > 3) ? -1 : ((refm >> readc) & 1);
				if(m != 1) {
					h2o_edit* e = &o->edits[ned < H2O_MAX_EDITS ? ned : H2O_MAX_EDITS - 1];
					e->pos = row; e->chr = (uint8_t)MASK2DNA[refm]; e->qchr = (uint8_t)"ACGTN"[readc]; e->type = H2O_EDIT_MM; e->pad = 0; e->snp = H2O_MAX; ned++;
					const int q = (qual ? qual[row] : 'I') - 33;
					score -= (readc > 3 || refm > 15) ? sc->nPen : mmpen_q(sc, q);
				}
				if(m == -1) ns++;
				if(ns > nceil) ok = 0;
			}
			if(ok) {
				if(ned > H2O_MAX_EDITS) { o->overflow = 1; ned = H2O_MAX_EDITS; }
				for(uint32_t a = 0; a < ned / 2; a++) { h2o_edit t = o->edits[a]; o->edits[a] = o->edits[ned - 1 - a]; o->edits[ned - 1 - a] = t; }
				o->found = 1; o->score = score; o->nedits = ned; o->off = (int64_t)col + rfi; o->gaps = gaps; *ns_out = ns;
				(void)origCol; (void)fail; (void)refGaps; (void)readGaps;
			}
			*rnd = TOP == 0xffu ? reseed + 1 : reseed;      /* aligner_sw.cpp:840 (8-bit branch: rnd.init(reseed + 1)) / :906 (16-bit branch: rnd.init(reseed)) */
		}
		free(stack); free(cells); free(cand);
	}
	free(rf); free(H); free(E); free(F); free(M);
	return o->found;
}

/* ------------------------------------------------------------------ graph SA walk (a14 on a graph index) */
/* getGenomeCoords hi_aligner.h:5774-5855 on a GRAPH index: GroupWalk2S::init/advanceElement (group_walk.h:1430-1545)
 * driving GWState::init (:464-885) and GWState::advance (:1035-1336).  Elements are NODES of [node_top, node_bot); a
 * range is walked left as a group, split by preceding character (mapLFRange masks, gfm.h:3636) and at '$' rows, and
 * merged when several rows lead into one node — the merged-away duplicates are "resolved" with their own element index
 * as the offset (group_walk.h:1171, 1246), which is reproduced here because it reaches the alignment through
 * joinedToTextOff.  tryOffset (gfm.h:2719) samples by NODE: (node & offMask) == node -> offs[node >> offRate]. */
#define GW_MAXELT 64
#define GW_MAXST 96
#define GW_MAXIE 64
typedef struct { uint32_t first, second; } gw_pair;
typedef struct {
	uint32_t top, bot, node_top, node_bot, step, mapi, nmap, nie;
	uint32_t map[GW_MAXELT];
	gw_pair ie[GW_MAXIE];
} gw_state;
typedef struct {
	const h2o_gfm* g;
	uint32_t topf, botf, sa_node_top, nelt;
	uint32_t offs[GW_MAXELT];
	gw_pair fmap[GW_MAXELT];
	gw_state st[GW_MAXST];
	uint32_t nst;
	uint32_t nsteps;
	int overflow;
} gw_ctx;

static uint32_t gw_try_offset(const h2o_gfm* g, uint32_t row, uint32_t node) {
	for(uint32_t i = 0; i < g->nZ; i++) if(row == g->zOffs[i]) return 0;
	if((node & g->p.offMask) == node) return g->offs[node >> g->p.offRate];
	return H2O_MAX;
}
/* mapGLF1(row, l, &node_range) without a required character (gfm.h:4029-4095) */
static void gw_map_glf1_nochar(const h2o_gfm* g, uint32_t row, uint32_t* otop, uint32_t* obot, uint32_t* ontop, uint32_t* onbot) {
	for(uint32_t i = 0; i < g->nZ; i++) if(row == g->zOffs[i]) { *otop = *obot = H2O_MAX; *ontop = *onbot = 0; return; }
	const int c = h2o_rowL(g, row);
	uint32_t t = h2o_rank(g, row, c);
	if(g->p.linear) { *otop = *ontop = t; *obot = *onbot = t + 1; return; }
	uint32_t node_top = h2o_rank_M(g, t + 1) - 1, F_loc, M_occ;
	uint32_t ft = node_to_Frow(g, t + 1, node_top, &F_loc, &M_occ);
	uint32_t node_bot = node_top + 1;
	uint32_t fb = (node_bot + 1 > M_occ) ? h2o_select_F(g, F_loc, node_bot + 1 - M_occ) : F_loc;
	*otop = ft; *obot = fb; *ontop = node_top; *onbot = node_bot;
}
static void gw_init(gw_ctx* x, uint32_t range);
static gw_state* gw_new_state(gw_ctx* x) {
	if(x->nst >= GW_MAXST) { x->overflow = 1; return &x->st[GW_MAXST - 1]; }
	gw_state* s = &x->st[x->nst++];
	memset(s, 0, sizeof *s);
	return s;
}
/* GWState::init (group_walk.h:506-885); top/bot/node range/iedges/step/map already set */
static void gw_init(gw_ctx* x, uint32_t range) {
	const h2o_gfm* g = x->g;
	gw_state* s = &x->st[range];
	uint32_t trimBegin = 0, trimEnd = 0;
	int empty = 1;
	uint32_t num_iedges = 0, e = 0;
	for(uint32_t i = s->mapi; i < s->nmap; i++) {
		int resolved = x->offs[s->map[i]] != H2O_MAX;
		if(!resolved) {
			while(e < s->nie) {
				if(i <= s->ie[e].first) break;
				num_iedges += s->ie[e].second;
				e++;
			}
			uint32_t bwrow = s->top + i + num_iedges, node = s->node_top + i;
			uint32_t toff = gw_try_offset(g, bwrow, node);
			if(toff != H2O_MAX) {
				toff += s->step;
				x->offs[s->map[i + s->mapi]] = toff;        /* setOff(i, ...) indexes map_[i + mapi_] (:1013) */
			}
		}
		if(x->offs[s->map[i]] != H2O_MAX) {
			if(empty) trimBegin++; else trimEnd++;
		} else {
			trimEnd = 0;
			empty = 0;
			x->fmap[s->map[i]].first = range;
			x->fmap[s->map[i]].second = i;
		}
	}
	s->mapi += trimBegin;
	if(trimBegin > 0) {
		s->top += trimBegin;
		uint32_t k = 0;
		for(; k < s->nie; k++) {
			if(s->ie[k].first >= trimBegin) break;
			s->top += s->ie[k].second;
		}
		if(k > 0) { memmove(s->ie, s->ie + k, sizeof(gw_pair) * (s->nie - k)); s->nie -= k; }
		for(k = 0; k < s->nie; k++) s->ie[k].first -= trimBegin;
	}
	s->node_top += trimBegin;
	if(trimEnd > 0) {
		s->nmap -= trimEnd;
		s->bot -= trimEnd;
		uint32_t node_range = s->node_bot - s->node_top;
		while(s->nie > 0) {
			if(s->ie[s->nie - 1].first < (node_range - trimEnd)) break;
			s->bot -= s->ie[s->nie - 1].second;
			s->nie--;
		}
	}
	s->node_bot -= trimEnd;
	if(empty) return;
	/* '$' rows strictly inside (top, bot): split (:741-868) */
	uint32_t zin[8], nz = 0;
	for(uint32_t i = 0; i < g->nZ; i++) if(g->zOffs[i] > s->top && g->zOffs[i] < s->bot && nz < 8) zin[nz++] = g->zOffs[i];
	if(nz > 0) {
		uint32_t g2n[GW_MAXELT * 4], ng = 0;
		uint32_t n = 0, ee = 0;
		for(uint32_t r = 0; r < s->bot - s->top; r++) {
			if(ng < GW_MAXELT * 4) g2n[ng++] = n; else x->overflow = 1;
			if(ee < s->nie) {
				if(n == s->ie[ee].first) {
					for(uint32_t a = 0; a < s->ie[ee].second; a++) { if(ng < GW_MAXELT * 4) g2n[ng++] = n; else x->overflow = 1; r++; }
					ee++;
				}
			}
			n++;
		}
		for(uint32_t i = 0; i < nz; i++) {
			s = &x->st[range];
			uint32_t new_top = zin[i] + 1;
			if(i + 1 < nz && new_top == zin[i + 1]) continue;
			if(new_top - s->top == ng) break;
			uint32_t new_node_top = g2n[new_top - s->top] + s->node_top;
			uint32_t new_bot = (i + 1 < nz) ? zin[i + 1] : s->bot;
			uint32_t new_node_bot = s->node_bot;
			if(new_bot - s->top < ng) {
				new_node_bot = s->node_top + g2n[new_bot - s->top];
				if(new_bot - s->top > 0 && g2n[new_bot - s->top] == g2n[new_bot - s->top - 1]) new_node_bot++;
			}
			if(new_top >= new_bot) continue;
			gw_pair tie[GW_MAXIE]; uint32_t ntie = 0;
			for(uint32_t j = new_top - s->top; j + 1 < new_bot - s->top;) {
				uint32_t nn = g2n[j], j2 = j + 1;
				while(j2 < new_bot - s->top) { if(nn != g2n[j2]) break; j2++; }
				if(j + 1 < j2 && ntie < GW_MAXIE) { tie[ntie].first = nn - (new_node_top - s->node_top); tie[ntie].second = j2 - j - 1; ntie++; }
				j = j2;
			}
			gw_state* ns = gw_new_state(x);
			s = &x->st[range];
			ns->nmap = new_node_bot - new_node_top; ns->mapi = 0;
			for(uint32_t j = new_node_top; j < new_node_bot; j++) ns->map[j - new_node_top] = s->map[j - s->node_top + s->mapi];
			ns->top = new_top; ns->bot = new_bot; ns->node_top = new_node_top; ns->node_bot = new_node_bot;
			ns->nie = ntie; memcpy(ns->ie, tie, sizeof(gw_pair) * ntie);
			ns->step = s->step;
			gw_init(x, x->nst - 1);
		}
		s = &x->st[range];
		s->bot = zin[0];
		s->node_bot = g2n[s->bot - s->top - 1] + s->node_top + 1;
		s->nmap = s->node_bot - s->node_top + s->mapi;
		uint32_t width = s->node_bot - s->node_top;
		for(uint32_t k = 0; k < s->nie; k++) {
			if(s->ie[k].first >= s->node_bot - s->node_top) { s->nie = k; break; }
			width += s->ie[k].second;
		}
		if(width != s->bot - s->top) {
			s->ie[s->nie - 1].second -= 1;
			if(s->ie[s->nie - 1].second == 0) s->nie--;
		}
	}
}
/* narrow a freshly mapped (node-merged) element list: group_walk.h:1143-1185 / :1218-1262 */
static void gw_merge_dups(gw_ctx* x, uint32_t curtop, const uint8_t* mask, uint32_t nmask, int c, uint32_t* map, uint32_t* nmap) {
	const h2o_gfm* g = x->g;
	uint32_t j1 = 0, j2 = 0;
	for(uint32_t k = 0; k < nmask; k++) if(mask[k]) { j1 = k; break; }
	for(uint32_t j = 0; j + 1 < *nmap; j++) {
		for(uint32_t k = j1 + 1; k < nmask; k++) if(mask[k]) { j2 = k; break; }
		uint32_t t, b, nt, nb, nie = 0;
		h2o_map_glf(g, curtop + j1, curtop + j2 + 1, c, 5, &t, &b, &nt, &nb, NULL, 0, &nie);
		if(nb - nt == 1) { x->offs[map[j]] = map[j]; map[j] = H2O_MAX; }
		j1 = j2; j2 = 0;
	}
	uint32_t w = 0;
	for(uint32_t j = 0; j < *nmap; j++) if(map[j] != H2O_MAX) map[w++] = map[j];
	*nmap = w;
}
/* GWState::advance (group_walk.h:1035-1336) */
static void gw_advance(gw_ctx* x, uint32_t range) {
	const h2o_gfm* g = x->g;
	gw_state* s = &x->st[range];
	x->nsteps++;
	if(s->bot - s->top > 1) {
		int first = 1;
		uint32_t newtop = 0, newbot = 0, new_node_top = 0, new_node_bot = 0;
		uint32_t gmap[GW_MAXELT], ngmap = 0;
		gw_pair backup[GW_MAXIE]; uint32_t nbackup = 0;
		uint32_t curtop = s->top, curbot = s->bot, cur_node_top = s->node_top, cur_node_bot = s->node_bot;
		for(uint32_t e = 0; e < s->nie + 1; e++) {
			s = &x->st[range];
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
			uint32_t n = curbot - curtop, in[4] = {0, 0, 0, 0};
			uint8_t mask[4][GW_MAXELT * 2];
			if(n > GW_MAXELT * 2) { x->overflow = 1; n = GW_MAXELT * 2; }
			memset(mask, 0, sizeof mask);
			for(uint32_t k = 0; k < n; k++) { int c = h2o_rowL(g, curtop + k); mask[c][k] = 1; in[c]++; }   /* mapLFRange */
			for(int c = 0; c < 4; c++) {
				if(in[c] == 0) continue;
				s = &x->st[range];
				uint32_t t, b, nt, nb;
				gw_pair tie[GW_MAXIE]; uint32_t ntie = 0, tmpie[2 * GW_MAXIE];
				h2o_map_glf(g, curtop, curbot, c, cur_node_bot - cur_node_top, &t, &b, &nt, &nb, tmpie, GW_MAXIE, &ntie);
				if(ntie > GW_MAXIE) { ntie = GW_MAXIE; x->overflow = 1; }
				for(uint32_t k = 0; k < ntie; k++) { tie[k].first = tmpie[2 * k]; tie[k].second = tmpie[2 * k + 1]; }
				if(first) {
					first = 0;
					newtop = t; newbot = b; new_node_top = nt; new_node_bot = nb;
					nbackup = ntie; memcpy(backup, tie, sizeof(gw_pair) * ntie);
					for(uint32_t j = 0; j < n; j++) if(mask[c][j]) gmap[ngmap++] = s->map[j + s->mapi + (cur_node_top - s->node_top)];
					if(new_node_bot - new_node_top < ngmap) gw_merge_dups(x, curtop, mask[c], n, c, gmap, &ngmap);
				} else {
					gw_state* ns = gw_new_state(x);
					s = &x->st[range];
					ns->mapi = 0; ns->nmap = 0;
					for(uint32_t j = 0; j < n; j++) if(mask[c][j]) ns->map[ns->nmap++] = s->map[j + s->mapi + (cur_node_top - s->node_top)];
					if(nb - nt < ns->nmap) gw_merge_dups(x, curtop, mask[c], n, c, ns->map, &ns->nmap);
					ns->top = t; ns->bot = b; ns->node_top = nt; ns->node_bot = nb;
					ns->nie = ntie; memcpy(ns->ie, tie, sizeof(gw_pair) * ntie);
					ns->step = s->step + 1;
					gw_init(x, x->nst - 1);
				}
			}
		}
		s = &x->st[range];
		s->mapi = 0;
		s->top = newtop; s->bot = newbot; s->node_top = new_node_top; s->node_bot = new_node_bot;
		s->nie = nbackup; memcpy(s->ie, backup, sizeof(gw_pair) * nbackup);
		if(ngmap > 0) { memcpy(s->map, gmap, 4 * ngmap); s->nmap = ngmap; }
	} else {
		uint32_t t, b, nt, nb;
		gw_map_glf1_nochar(g, s->top, &t, &b, &nt, &nb);
		s->top = t; s->bot = t + 1; s->node_top = nt; s->node_bot = nb;
		if(s->mapi > 0) { s->map[0] = s->map[s->mapi]; s->mapi = 0; }
		s->nmap = 1;
		(void)b;
	}
	s->step++;
	gw_init(x, range);
}

int h2o_genome_coords_graph(const h2o_index* ix, uint32_t top, uint32_t bot, uint32_t node_top, uint32_t node_bot,
                            const uint32_t* iedges, uint32_t niedges, uint32_t maxelt, uint32_t rdlen, int rejectStraddle,
                            h2o_coord* coords, uint32_t* ncoords, int* straddled, uint32_t* nsteps)
{
	static gw_ctx ctx;                                      /* ~60 KB; the oracle is single-threaded test code */
	gw_ctx* x = &ctx;
	const h2o_gfm* g = &ix->g;
	*straddled = 0;
	uint32_t nelt = node_bot - node_top;
	if(nelt > maxelt) nelt = maxelt;
	if(nelt > GW_MAXELT) return -1;
	x->g = g; x->topf = top; x->botf = bot; x->sa_node_top = node_top; x->nelt = nelt; x->nst = 0; x->nsteps = 0; x->overflow = 0;
	for(uint32_t i = 0; i < nelt; i++) { x->offs[i] = H2O_MAX; x->fmap[i].first = x->fmap[i].second = H2O_MAX; }
	gw_state* s = gw_new_state(x);                          /* GroupWalk2S::init :1430-1470 */
	s->nmap = nelt; s->mapi = 0;
	for(uint32_t i = 0; i < nelt; i++) s->map[i] = i;
	s->top = top; s->bot = bot; s->node_top = node_top; s->node_bot = node_top + nelt;
	s->nie = niedges < GW_MAXIE ? niedges : GW_MAXIE;
	for(uint32_t k = 0; k < s->nie; k++) { s->ie[k].first = iedges[2 * k]; s->ie[k].second = iedges[2 * k + 1]; }
	s->step = 0;
	gw_init(x, 0);
	uint32_t n = *ncoords;
	for(uint32_t elt = 0; elt < nelt; elt++) {
		uint32_t guard = 0;
		while(x->offs[elt] == H2O_MAX) {                     /* advanceElement :1491-1545 */
			gw_advance(x, x->fmap[elt].first);
			if(++guard > 100000 || x->overflow) return -1;
		}
		uint32_t tidx = 0, toff = 0, tlen = 0;
		int st2 = 0;
		h2o_joined_to_text(g, rdlen, x->offs[elt], &tidx, &toff, &tlen, rejectStraddle, &st2);
		*straddled |= st2;
		if(tidx == H2O_MAX) { *ncoords = n; if(nsteps) *nsteps += x->nsteps; return 0; }
		coords[n].tidx = st2 ? H2O_MAX : tidx;
		coords[n].toff = toff;
		coords[n].joinedOff = x->offs[elt];
		n++;
	}
	*ncoords = n;
	if(nsteps) *nsteps += x->nsteps;
	return 1;
}
