// h2g_emul.cpp — TEST-ONLY host instantiation of the per-item device functions (hisat2_amd/csrc/h2g_core.h).
//
// The build container has no GPU, so the `__host__ __device__` item functions that the HIP kernels wrap are
// also compiled here with g++ and driven item-by-item, to check the kernel logic against the golden vectors in
// `-m "not gpu"` runs.  This library lives under tests/, is never shipped with or loaded by hisat2_amd, and
// is not a fallback: libh2g.so has no host execution path.
#include <vector>
#include <stdlib.h>
#include <string.h>
// the host instantiation runs with the large-workspace capacities (one workspace, no second pass on the host)
#ifndef H2G_MEMPROF   // tools/memprof profiles the default device workspace layout
#include "../../hisat2_amd/csrc/h2g_go_big.h"
#endif
#include "../../hisat2_amd/csrc/h2g_core.h"
#include "../../hisat2_amd/csrc/h2g_host_index.h"
#include "../../hisat2_amd/csrc/h2g_align.h"
#include "../../hisat2_amd/csrc/h2g_fast.h"
#include "../../hisat2_amd/csrc/h2g_graph.h"
#include "../../hisat2_amd/csrc/h2g_sw.h"
#include "../../hisat2_amd/csrc/h2g_graph_staged.h"
#include "../../hisat2_amd/csrc/h2g_local_pack.h"
#include "../../hisat2_amd/csrc/h2g_splice_host.h"
#include "../../hisat2_amd/csrc/h2g_splice_db_host.h"

using namespace h2g;

struct Emu {
	HostIndex host;
	DGfm dg;
	DRef dr;
	std::vector<uint8_t> codes;
	std::vector<uint32_t> offs;
	std::vector<char> quals;
	bool has_quals = false;
	LocalPack lp;
	DLocalSet dls;
	DAlts dalts;
	std::vector<uint64_t> altbuf;                         // the ALTs followed by the haplotype table (pack_alts)
	std::vector<uint32_t> alt_bk;
	uint32_t bowtie2_dp = 0;
	bool has_params = false;
	h2g_align_params params;
	std::vector<uint8_t> sw;
	HostSpliceDB hssdb; DSpliceDB dssdb; uint32_t rdid_base = 0;
	std::vector<h2g_splice_site> alt_sites;
	DReads reads() const {
		DReads r;
		r.codes = codes.data(); r.offs = offs.data(); r.quals = has_quals ? quals.data() : nullptr;
		r.n = (uint32_t)offs.size() - 1;
		return r;
	}
};

extern "C" {

void h2gemu_set_splice_sites(Emu* e, const h2g_splice_site* sites, size_t n, uint32_t window);

int h2gemu_load(const char* base, Emu** out) {
	Emu* e = new Emu();
	int rc = load_host_index(base, true, e->host);
	if(rc) { delete e; return rc; }
	const HostGfm& g = e->host.g;
	DGfm& d = e->dg;
	d.sides = g.sides.data(); d.ftab = g.ftab.data(); d.eftab = g.eftab.data(); d.offs = g.offs.data();
	d.rstarts = g.rstarts.data(); d.plen = g.plen.data(); d.zoffs = g.zOffs.data();
	for(int i = 0; i < 5; i++) d.fchr[i] = g.fchr[i];
	d.len = g.p.len; d.gbwtLen = g.p.gbwtLen; d.ftabLim = g.p.linear ? g.p.len : g.p.gbwtLen;
	d.sideGbwtLen = g.p.sideGbwtLen; d.sideGbwtSz = g.p.sideGbwtSz; d.lineRate = g.p.lineRate; d.offRate = g.p.offRate;
	d.offMask = g.p.offMask; d.ftabChars = g.p.ftabChars; d.nFrag = g.nFrag; d.nPat = g.nPat;
	d.nZ = (uint32_t)g.zOffs.size(); d.zoff = g.zOffs.empty() ? H2G_MAX : g.zOffs[0]; d.minK = e->host.minK;
	d.linear = g.p.linear;
	const HostRef& r = e->host.r;
	e->dr.buf = r.buf.data(); e->dr.rec_start = r.rec_start.data(); e->dr.rec_len = r.rec_len.data();
	e->dr.rec_bufoff = r.rec_bufoff.data(); e->dr.refRecOffs = r.refRecOffs.data(); e->dr.refLens = r.refLens.data();
	e->dr.nrefs = r.nrefs;
	pack_alts(e->host.alts, e->host.hap_left, e->host.hap_right, e->host.hap_maxright, e->host.hap_first, e->host.hap_ids, e->altbuf);
	e->dalts.a = reinterpret_cast<const DAlt*>(e->altbuf.data()); e->dalts.n = g.p.linear ? 0 : (uint32_t)e->host.alts.size();
	e->dalts.maxAltsTried = 16;
	if(!g.p.linear) {
		splice_sites_of_alts(reinterpret_cast<const uint32_t*>(e->host.alts.data()), e->host.alts.size(), sizeof(HostAlt) / 4, g.rstarts.data(), g.nFrag, g.p.len, e->alt_sites);
		for(const HostAlt& a : e->host.alts) if(a.type == 5) e->dalts.has_splice = 1;
	}
	alt_buckets(e->dalts.a, e->dalts.n, e->alt_bk);
	if(!e->alt_bk.empty()) { e->dalts.bucket = e->alt_bk.data(); e->dalts.nbucket = (uint32_t)e->alt_bk.size(); }
	pack_local(e->host, e->lp);
	e->dls = e->lp.view(e->lp.desc.data(), e->lp.sides.data(), e->lp.words.data(), e->lp.first.data(), e->lp.zoffs.data());
	if(!e->alt_sites.empty()) h2gemu_set_splice_sites(e, nullptr, 0, 0);   // a --ss index: its splice sites are known sites of the database
	*out = e;
	return 0;
}

void h2gemu_set_bowtie2_dp(Emu* e, uint32_t v) { e->bowtie2_dp = v; }
// full option block (same struct as the C ABI); defaults: h2gemu_default_params
void h2gemu_default_params(Emu* e, h2g_align_params* p) { align_params_defaults(p, e->dg.linear); }
void h2gemu_set_params(Emu* e, const h2g_align_params* p) { e->params = *p; e->has_params = true; }

void h2gemu_set_reads(Emu* e, const uint8_t* codes, const uint32_t* offs, const char* quals, size_t n) {
	e->codes.assign(codes, codes + offs[n]);
	e->codes.resize((size_t)offs[n] + 8, 0);   // word-wise readers (fg_pack_read) may touch up to 3 bytes past the last read
	e->offs.assign(offs, offs + n + 1);
	e->has_quals = quals != nullptr;
	if(quals) e->quals.assign(quals, quals + offs[n]);
}

void h2gemu_rank(Emu* e, const uint32_t* rows, const uint8_t* cs, size_t n, uint32_t* out) {
	const bool graph = !e->dg.linear;
	for(size_t i = 0; i < n; i++) out[i] = graph ? rank128(e->dg, rows[i], cs[i]) : rank64(e->dg, rows[i], cs[i]);
}

void h2gemu_graph_lf(Emu* e, const h2g_glf_query* q, size_t n, uint32_t k, h2g_glf_result* res, h2g_iedges* ie) {
	for(size_t i = 0; i < n; i++) {
		GRange r;
		IEdges x;
		x.n = 0;
		bool ok = q[i].single ? map_glf1(e->dg, q[i].top, q[i].c, &r) : map_glf(e->dg, q[i].top, q[i].bot, q[i].c, k, &r, &x);
		res[i].ok = ok; res[i].top = r.top; res[i].bot = r.bot; res[i].node_top = r.node_top; res[i].node_bot = r.node_bot;
		if(ie) ie[i] = x;
	}
}

void h2gemu_sa_resolve_graph(Emu* e, const h2g_gsa_query* q, const h2g_iedges* ie, size_t n, uint32_t cap, h2g_coord* coords, h2g_sa_result* res) {
	GwCtx* x = new GwCtx();
	for(size_t i = 0; i < n; i++)
		genome_coords_graph_item(e->dg, x, q[i].top, q[i].bot, q[i].node_top, q[i].node_bot, ie ? &ie[i] : nullptr, q[i].maxelt, q[i].len,
		                         q[i].rejectStraddle != 0, coords + i * cap, cap, &res[i]);
	delete x;
}

void h2gemu_adjust_with_alt(Emu* e, const h2g_adjust_query* q, size_t n, uint32_t cap, h2g_ghit* hits, uint32_t* nhits) {
	DReads rd = e->reads();
	AwaWS* W = new AwaWS();
	for(size_t i = 0; i < n; i++) {
		SeqView sv = seq_view(rd, q[i].read, q[i].fw != 0);
		uint32_t nh = 0, ovf = 0;
		adjust_with_alt(e->dg, e->dr, e->dalts, sv, q[i].rdoff, q[i].len, q[i].tidx, q[i].toff, q[i].joinedOff, hits + i * cap, &nh, cap, W, &ovf);
		nhits[i] = ovf ? H2G_MAX : nh;
	}
	delete W;
}

// mapGLF / mapGLF1 on the LOCAL graph index covering (tidx, toff): q = {single, tidx, toff, top, bot, c}
void h2gemu_local_graph_lf(Emu* e, const uint32_t* q, size_t n, uint32_t k, h2g_glf_result* res, h2g_iedges* ie) {
	for(size_t i = 0; i < n; i++) {
		const uint32_t* a = q + 6 * i;
		const uint32_t lidx = local_index_of(e->dls, a[1], a[2]);
		LGfm x = lgfm_of(e->dls, e->dls.desc[lidx]);
		GRange r;
		IEdges t;
		t.n = 0;
		bool ok = a[0] ? map_glf1(x, a[3], (int)a[5], &r) : map_glf(x, a[3], a[4], (int)a[5], k, &r, &t);
		res[i].ok = ok; res[i].top = r.top; res[i].bot = r.bot; res[i].node_top = r.node_top; res[i].node_bot = r.node_bot;
		ie[i] = t;
	}
}

void h2gemu_fm_search_graph(Emu* e, const h2g_fm_query* q, size_t n, uint32_t khits, uint32_t kseeds, h2g_fm_hit* out, h2g_iedges* ie) {
	DReads rd = e->reads();
	for(size_t i = 0; i < n; i++) {
		SeqView sv = seq_view(rd, q[i].read, q[i].fw != 0);
		partial_search_graph_item(e->dg, sv, q[i].offset, q[i].pseudogeneStop != 0, q[i].anchorStop != 0, khits, kseeds, &out[i],
		                          ie ? &ie[i] : nullptr);
	}
}

void h2gemu_fm_search(Emu* e, const h2g_fm_query* q, size_t n, uint32_t khits, h2g_fm_hit* out) {
	DReads rd = e->reads();
	for(size_t i = 0; i < n; i++) {
		SeqView sv = seq_view(rd, q[i].read, q[i].fw != 0);
		partial_search_item(e->dg, sv, q[i].offset, q[i].pseudogeneStop != 0, q[i].anchorStop != 0, khits, &out[i]);
	}
}

uint32_t h2gemu_local_index_of(Emu* e, uint32_t tidx, uint32_t toff) { return local_index_of(e->dls, tidx, toff); }

// globalGFMSearch / localGFMSearch as h2g_ext_search runs them (linear index)
void h2gemu_ext_search(Emu* e, const h2g_ext_search_query* q, size_t n, h2g_ext_search_hit* out) {
	DReads rd = e->reads();
	h2g_align_params hp;
	align_params_defaults(&hp, e->dg.linear);
	const AlnParams P = aln_params_from(hp, true, e->dg.linear);
	for(size_t i = 0; i < n; i++) {
		const SeqView sv = seq_view(rd, q[i].read, q[i].fw != 0);
		uint32_t hitlen = 0, top = H2G_MAX, bot = H2G_MAX, nr[2] = {0, 0}, nelt = 0;
		bool us = q[i].uniqueStop != 0;
		if(q[i].lidx == H2G_MAX) { GIdx gx; gx.g = &e->dg; nelt = gfm_search(gx, sv, q[i].rdoff, &hitlen, &top, &bot, &us, e->dg.minK, H2G_MAX, P.kseeds, false, nr); }
		else { LIdx lx; lx.ls = &e->dls; lx.d = &e->dls.desc[q[i].lidx]; nelt = lx.d->len == 0 ? 0 : gfm_search(lx, sv, q[i].rdoff, &hitlen, &top, &bot, &us, P.minK_local, q[i].maxHitLen, P.kseeds, true, nr); }
		out[i].nelt = nelt; out[i].hitlen = hitlen; out[i].top = top; out[i].bot = bot; out[i].uniqueStop = us; out[i].nrank = nr[0]; out[i].nside = nr[1]; out[i].staged = 0;
	}
}

void h2gemu_sw_align(Emu* e, const h2g_sw_query* q, size_t n, h2g_sw_result* out) {
	// both matrix layouts through the same gather/backtrace: odd problems use the anti-diagonal-major layout of the
	// wavefront kernel (filled here cell by cell), even ones the row-major layout of the in-go() path
	DReads rd = e->reads();
	SwParams P;
	std::vector<uint8_t> scratch(sw_scratch_bytes(H2G_SW_MAX_ROWS, true));
	SwLaneState* ls = new SwLaneState();
	memset(ls, 0, sizeof *ls);
	for(size_t p = 0; p < n; p++) {
		SeqView sv = seq_view(rd, q[p].read, q[p].fw != 0);
		uint32_t rnd = q[p].rnd;
		SwOut* o = nullptr;
		if((p & 1) == 0) sw_align_single(e->dr, P, sv, q[p].tidx, q[p].refoff, q[p].minsc, &rnd, scratch.data(), &o);
		else {
			const uint32_t nrow = sv.len;
			const SwRect rect = sw_frame(q[p].refoff, nrow, e->dr.refLens[q[p].tidx]);
			const uint32_t ncol = (uint32_t)(rect.refr - rect.refl + 1);
			SwMats m;
			m.nrow = nrow; m.ncol = ncol; m.nd = nrow + ncol - 1; m.layout = 1; m.wide = sw_wide_for(q[p].minsc);
			std::vector<uint8_t> H(m.bytes()), E(H.size()), F(H.size()), rf(ncol);
			m.H = H.data(); m.E = E.data(); m.F = F.data(); m.rf = rf.data();
			RefCursor rc;
			rc.init(&e->dr, q[p].tidx);
			for(uint32_t j = 0; j < ncol; j++) rf[j] = (uint8_t)rc.get(rect.refl + (int64_t)j);
			sw_fill<false>(m, P, sv, 0, 1);
			std::vector<uint16_t> direct((size_t)nrow * ncol, 0xabcd);
			o = sw_finish(m, P, sv, rect, q[p].minsc, &rnd, ls, direct.data());
		}
		memcpy(&out[p], o, sizeof(SwOut));
	}
	delete ls;
}

// The reference's own SwAligner known-answer cases (tests/golden/sw_kat.json, aligner_sw.cpp:1470-2727) through the product's DP: read `read` of
// the batch over columns refl..refr of a plain reference string (N outside it), core diagonals corel..corer, a scoring of the case's own.
// scoring = mmpMax mmpMin nPen rdGapConst rdGapLinear rfGapConst rfGapLinear; layout 0 = row-major cells, 1 = the wave kernel's anti-diagonal-major.
void h2gemu_sw_align_window(Emu* e, uint32_t read, const uint8_t* ref, uint32_t reflen, int64_t refl, int64_t refr, int64_t corel, int64_t corer,
                            const int* scoring, int gapbar, int64_t minsc, int nceil, uint32_t layout, uint32_t rnd, h2g_sw_result* out) {
	DReads rd = e->reads();
	SeqView sv = seq_view(rd, read, true);
	SwParams P;
	P.sc.mmpMax = scoring[0]; P.sc.mmpMin = scoring[1]; P.sc.nPen = scoring[2];
	P.sc.rdGapConst = scoring[3]; P.sc.rdGapLinear = scoring[4]; P.sc.rfGapConst = scoring[5]; P.sc.rfGapLinear = scoring[6];
	P.gapbar = gapbar;
	SwRect rect;
	rect.refl = rect.refl_pretrim = refl; rect.refr = rect.refr_pretrim = refr; rect.triml = rect.trimr = 0; rect.corel = corel; rect.corer = corer;
	const uint32_t nrow = sv.len, ncol = (uint32_t)(refr - refl + 1);
	SwMats m;
	m.nrow = nrow; m.ncol = ncol; m.nd = nrow + ncol - 1; m.layout = layout; m.wide = sw_wide_for(minsc);
	std::vector<uint8_t> H(m.bytes()), E(H.size()), F(H.size()), rf(ncol);
	m.H = H.data(); m.E = E.data(); m.F = F.data(); m.rf = rf.data();
	for(uint32_t j = 0; j < ncol; j++) { const int64_t p = refl + (int64_t)j; rf[j] = (p < 0 || p >= (int64_t)reflen) ? 4 : ref[p]; }
	sw_fill<false>(m, P, sv, 0, 1);
	SwLaneState* ls = new SwLaneState();
	memset(ls, 0, sizeof *ls);
	std::vector<uint16_t> direct((size_t)nrow * ncol);
	const SwOut* o = sw_finish(m, P, sv, rect, minsc, &rnd, ls, direct.data(), nceil);
	memcpy(out, o, sizeof *out);
	delete ls;
}

void h2gemu_sa_resolve(Emu* e, const h2g_sa_query* q, size_t n, uint32_t cap, h2g_coord* coords, h2g_sa_result* res) {
	for(size_t i = 0; i < n; i++)
		genome_coords_item(e->dg, q[i].top, q[i].bot, q[i].maxelt, q[i].len, q[i].rejectStraddle != 0, coords + i * cap, cap, &res[i]);
}

void h2gemu_extend(Emu* e, h2g_ghit* hits, const h2g_ext_args* args, size_t n, h2g_ext_result* res) {
	DReads rd = e->reads();
	DScoring sc;
	for(size_t i = 0; i < n; i++) {
		SeqView sv = seq_view(rd, hits[i].read, hits[i].fw != 0);
		uint32_t le = 0, re = 0;
		bool ext;
		if(e->dg.linear) ext = extend_item(e->dr, sc, sv, &hits[i], args[i].mm, args[i].max_leftext, args[i].max_rightext, &le, &re);
		else {
			static AwaWS W;
			ext = extend_item_alts(e->dr, e->dalts, sc, sv, &hits[i], args[i].mm, args[i].max_leftext, args[i].max_rightext, &le, &re, &W);
		}
		res[i].extended = ext; res[i].leftext = le; res[i].rightext = re;
	}
}

void h2gemu_seed_extend(Emu* e, uint32_t pseudogeneStop, uint32_t khits, h2g_seed_result* out) {
	DReads rd = e->reads();
	DScoring sc;
	h2g_ghit scratch;
	for(size_t i = 0; i < (size_t)rd.n * 2; i++) {
		SeqView sv = seq_view(rd, (uint32_t)(i >> 1), (i & 1) == 0);
		partial_search_item(e->dg, sv, 0, pseudogeneStop != 0, true, khits, &out[i].hit);
		resolve_extend_item(e->dg, e->dr, sc, sv, &out[i], &scratch);
	}
}


#define EMU_REC_STRIDE 32   // records per read in the output arrays (tests/sam_util.py AL_MAX_RESULTS), whatever the workspace holds
// HI_Aligner::go + selection for every read of the batch (names: '\0'-free bytes + offsets), one lane of the machine at a time
static void emu_ctx(Emu* e, uint32_t no_spliced, AlnParams* P, AlnCtx* C) {
	h2g_align_params hp;
	if(e->has_params) hp = e->params; else { align_params_defaults(&hp, e->dg.linear); hp.bowtie2_dp = e->bowtie2_dp; }
	*P = aln_params_from(hp, no_spliced != 0, e->dg.linear);
	if(!no_spliced) {   // spliced alignment: the splice-site probability tables of SpliceSiteDB::probscore
		static std::vector<float> d_, a1_, a2_;
		if(d_.empty()) splice_tables(d_, a1_, a2_);
		P->sc.donor_sum = d_.data(); P->sc.acc_sum1 = a1_.data(); P->sc.acc_sum2 = a2_.data();
	}
	C->g = &e->dg; C->ref = &e->dr; C->ls = &e->dls; C->P = P;
	C->ssdb = no_spliced ? nullptr : &e->dssdb; C->rdid_base = e->rdid_base;
	ctx_ext_opts(*C, *P);
	e->sw.resize(sw_scratch_bytes(H2G_SW_MAX_ROWS, true));
	C->sw = e->sw.data();
	static GraphWS gws_;
	static GraphSlot gsl_;
	static int64_t sc_[2 * H2G_COMBINE_MAXLEN];
	C->sc = sc_;
	e->dalts.maxAltsTried = hp.max_alts_tried ? hp.max_alts_tried : 16;
	e->dalts.has_splice = (e->dalts.has_splice & 1u) | (hp.use_haplotype && !e->dg.linear ? 2u : 0u);
	C->alts = &e->dalts; C->gws = e->dg.linear ? nullptr : &gws_; C->gsl = e->dg.linear ? nullptr : &gsl_; C->graph = !e->dg.linear;
}

// GenomeHit::combineWith (hit_combine) on n pairs of partial alignments of the resident reads: a[i] is combined with b[i] in place; ok[i] = its return value.
// minsc[i] as the reference passes it (scoreMin.f(read length)).
void h2gemu_combine(Emu* e, uint32_t no_spliced, h2g_ghit* a, const h2g_ghit* b, const int64_t* minsc, size_t n, uint32_t* ok) {
	AlnParams P; AlnCtx C;
	emu_ctx(e, no_spliced, &P, &C);
	DReads rd = e->reads();
	static int64_t t1[H2G_COMBINE_MAXLEN], t2[H2G_COMBINE_MAXLEN];
	for(size_t i = 0; i < n; i++) {
		SeqView sv = seq_view(rd, a[i].read, a[i].fw != 0);
		ok[i] = hit_combine(e->dr, P.sc, sv, &a[i], &b[i], minsc[i], P.minIntronLen, P.no_spliced != 0, ScVec{t1, 1}, ScVec{t2, 1}, C.alts) ? 1u : 0u;
	}
}

void h2gemu_set_rdid_base(Emu* e, uint32_t base) { e->rdid_base = base; }
void h2gemu_set_splice_sites(Emu* e, const h2g_splice_site* sites, size_t n, uint32_t window) {
	std::vector<h2g_splice_site> all(e->alt_sites);
	if(n) all.insert(all.end(), sites, sites + n);
	build_splice_db(all.data(), all.size(), e->host.g.nPat, e->hssdb);
	e->dssdb = DSpliceDB();
	if(e->hssdb.fw.empty()) return;
	e->dssdb.fw = e->hssdb.fw.data(); e->dssdb.bw = e->hssdb.bw.data();
	e->dssdb.fw_first = e->hssdb.fw_first.data(); e->dssdb.bw_first = e->hssdb.bw_first.data();
	e->dssdb.n = (uint32_t)e->hssdb.fw.size(); e->dssdb.window = window;
}

void h2gemu_align(Emu* e, uint32_t no_spliced, const char* names, const uint32_t* name_offs, ReadOut* outs, AlnRec* recs) {
	AlnParams P; AlnCtx C;
	emu_ctx(e, no_spliced, &P, &C);
	AlignWS* ws = new AlignWS();
	Mach M;
	M.ws = ws; M.rd[0] = e->reads(); M.rd[1] = M.rd[0];
	MachOut O; O.rout = outs; O.aln = nullptr; O.aln_slots = 0; O.pout = nullptr; O.paln[0] = O.paln[1] = nullptr; O.pair_slots = 0;
	for(uint32_t i = 0; i < M.rd[0].n; i++) {
		M.name[0] = names + name_offs[i]; M.namelen[0] = name_offs[i + 1] - name_offs[i];
		M.name[1] = nullptr; M.namelen[1] = 0;
		if(getenv("H2GEMU_POISON")) memset((void*)ws, atoi(getenv("H2GEMU_POISON")), sizeof *ws);   // stale-state hunting: nothing may be read before it is written
		mach_run_single(C, M, i, false, O);
		for(uint32_t k = 0; k < ws->m[0].nres; k++) if(k < EMU_REC_STRIDE) recs[(size_t)i * EMU_REC_STRIDE + k] = ws->m[0].res[k];
	}
	delete ws;
}


// go() with the result rows of the C ABI (h2g_align_fetch's layout: `slots` h2g_alnres per read, the selected alignments in order) and the long-edit
// area (MachOut::ledits): the path a record with more than H2G_MAX_EDITS edits takes out of the large-workspace units.  *ledits_used = edits written.
void h2gemu_align_abi(Emu* e, uint32_t no_spliced, const char* names, const uint32_t* name_offs, ReadOut* outs, h2g_alnres* rows, uint32_t slots,
                      h2g_edit* ledits, uint32_t ledits_cap, uint32_t* ledits_used) {
	AlnParams P; AlnCtx C;
	emu_ctx(e, no_spliced, &P, &C);
	AlignWS* ws = new AlignWS();
	Mach M;
	M.ws = ws; M.rd[0] = e->reads(); M.rd[1] = M.rd[0];
	uint32_t cursor = 0;
	MachOut O; O.rout = outs; O.aln = rows; O.aln_slots = slots; O.pout = nullptr; O.paln[0] = O.paln[1] = nullptr; O.pair_slots = 0;
	O.ledits = ledits; O.ledits_cursor = &cursor; O.ledits_cap = ledits_cap;
	for(uint32_t i = 0; i < M.rd[0].n; i++) {
		M.name[0] = names + name_offs[i]; M.namelen[0] = name_offs[i + 1] - name_offs[i];
		M.name[1] = nullptr; M.namelen[1] = 0;
		mach_run_single(C, M, i, false, O);
	}
	*ledits_used = cursor;
	delete ws;
}
uint32_t h2gemu_ghit_edits() { return H2G_GHIT_EDITS; }

// paired go(): mate 2 passed separately; names1/names2 as in the FASTA files
void h2gemu_align_pairs(Emu* e, uint32_t no_spliced, const uint8_t* codes2, const uint32_t* offs2, const char* names1,
                        const uint32_t* noffs1, const char* names2, const uint32_t* noffs2, PairOut* outs, AlnRec* recs1, AlnRec* recs2) {
	AlnParams P; AlnCtx C;
	emu_ctx(e, no_spliced, &P, &C);
	AlignWS* ws = new AlignWS();
	Mach M;
	M.ws = ws; M.rd[0] = e->reads(); M.rd[1] = M.rd[0];
	M.rd[1].codes = codes2; M.rd[1].offs = offs2; M.rd[1].quals = nullptr;
	MachOut O; O.rout = nullptr; O.aln = nullptr; O.pout = outs; O.paln[0] = O.paln[1] = nullptr; O.pair_slots = EMU_REC_STRIDE;
	for(uint32_t i = 0; i < M.rd[0].n; i++) {
		M.name[0] = names1 + noffs1[i]; M.namelen[0] = noffs1[i + 1] - noffs1[i];
		M.name[1] = names2 + noffs2[i]; M.namelen[1] = noffs2[i + 1] - noffs2[i];
		if(getenv("H2GEMU_POISON")) memset((void*)ws, atoi(getenv("H2GEMU_POISON")), sizeof *ws);
		mach_run_single(C, M, i, true, O);
		for(uint32_t k = 0; k < ws->m[0].nres; k++) if(k < EMU_REC_STRIDE) recs1[(size_t)i * EMU_REC_STRIDE + k] = ws->m[0].res[k];
		for(uint32_t k = 0; k < ws->m[1].nres; k++) if(k < EMU_REC_STRIDE) recs2[(size_t)i * EMU_REC_STRIDE + k] = ws->m[1].res[k];
	}
	delete ws;
}


static bool rec_equal(const h2g_alnres& a, const AlnRec& b);
// paired go() with `slots` record rows per mate and an overflow area of `ovf_cap` records (MachOut::ovf): every pair's records, read the way
// the dense fetch reads them (PairOut::pad -> its block, mate 2 behind mate 1; else its rows), must equal the workspace's lists.
// stats[0] pairs kept in the area, [1] pairs with a wrong record, [2] pairs flagged (area full), [3] records the area holds at the end
void h2gemu_pairs_overflow_check(Emu* e, const uint8_t* codes2, const uint32_t* offs2, const char* names1, const uint32_t* noffs1,
                                 const char* names2, const uint32_t* noffs2, uint32_t slots, uint32_t ovf_cap, uint64_t* stats) {
	AlnParams P; AlnCtx C;
	emu_ctx(e, 1, &P, &C);
	AlignWS* ws = new AlignWS();
	Mach M;
	M.ws = ws; M.rd[0] = e->reads(); M.rd[1] = M.rd[0];
	M.rd[1].codes = codes2; M.rd[1].offs = offs2; M.rd[1].quals = nullptr;
	const uint32_t n = M.rd[0].n;
	std::vector<PairOut> outs(n);
	std::vector<h2g_alnres> r1((size_t)n * slots), r2((size_t)n * slots), area(ovf_cap ? ovf_cap : 1);
	uint32_t cursor = 0;
	MachOut O; O.rout = nullptr; O.aln = nullptr; O.aln_slots = 0; O.pout = outs.data(); O.paln[0] = r1.data(); O.paln[1] = r2.data(); O.pair_slots = slots;
	O.ovf = area.data(); O.ovf_cursor = &cursor; O.ovf_cap = ovf_cap;
	stats[0] = stats[1] = stats[2] = stats[3] = 0;
	for(uint32_t i = 0; i < n; i++) {
		M.name[0] = names1 + noffs1[i]; M.namelen[0] = noffs1[i + 1] - noffs1[i];
		M.name[1] = names2 + noffs2[i]; M.namelen[1] = noffs2[i + 1] - noffs2[i];
		mach_run_single(C, M, i, true, O);
		const PairOut& o = outs[i];
		const bool big = o.nres[0] > slots || o.nres[1] > slots;
		bool bad = (o.pad != 0 || (o.overflow & 4)) != big || (o.pad != 0 && (o.overflow & 4));
		stats[0] += o.pad != 0; stats[2] += (o.overflow & 4) != 0;
		for(int m = 0; m < 2 && !bad; m++) {
			const h2g_alnres* a = o.pad ? area.data() + (o.pad - 1) + (m ? o.nres[0] : 0) : (m ? r2.data() : r1.data()) + (size_t)i * slots;
			const uint32_t c = o.pad ? o.nres[m] : (o.nres[m] < slots ? o.nres[m] : slots);
			if(o.nres[m] != ws->m[m].nres) bad = true;
			for(uint32_t k = 0; k < c && !bad; k++) bad = !rec_equal(a[k], ws->m[m].res[k]);
		}
		stats[1] += bad;
	}
	stats[3] = cursor;
	delete ws;
}


// ---- the fast path (h2g_fast.h) against the general machine: every read / pair of the batch through both; for the ones the
// fast path completes, its PairOut / ReadOut and records must equal the machine's.  stats[0] completed, [1] mismatching,
// [2 + why] bails by reason; bad_ids (cap entries) = the first mismatching read ids.  codes2 == nullptr: unpaired.
static bool rec_equal(const h2g_alnres& a, const AlnRec& b) {
	if(a.fw != b.fw || a.tidx != b.tidx || a.toff != b.toff || a.len != b.len || a.trim5 != b.trim5 || a.trim3 != b.trim3 || a.nedits != b.nedits ||
	   a.splicescore != b.splicescore || a.score != b.score) return false;
	for(uint32_t k = 0; k < a.nedits; k++) {
		const h2g_edit &x = a.edits[k], &y = b.edits[k];
		if(x.pos != y.pos || x.chr != y.chr || x.qchr != y.qchr || x.type != y.type || x.pad != y.pad || x.snp != y.snp) return false;
	}
	return true;
}
void h2gemu_fast_check(Emu* e, const uint8_t* codes2, const uint32_t* offs2, const char* names1, const uint32_t* noffs1, const char* names2,
                       const uint32_t* noffs2, uint64_t* stats, uint32_t* bad_ids, uint32_t cap, uint8_t* done_flags) {
	AlnParams P; AlnCtx C;
	emu_ctx(e, 1, &P, &C);
	const bool paired = codes2 != nullptr;
	AlignWS* ws = new AlignWS();
	Mach M;
	M.ws = ws; M.rd[0] = e->reads(); M.rd[1] = M.rd[0];
	if(paired) { M.rd[1].codes = codes2; M.rd[1].offs = offs2; M.rd[1].quals = nullptr; }
	const uint32_t n = M.rd[0].n, slots = 16;
	std::vector<PairOut> mp(1), fp(n);
	std::vector<ReadOut> mr(1), fr(n);
	std::vector<h2g_alnres> f1((size_t)n * slots), f2((size_t)n * slots), m1(slots), m2(slots);
	FCtx F;
	F.g = &e->dg; F.ref = &e->dr; F.ls = &e->dls; F.P = &P;
	F.rd[0] = M.rd[0]; F.rd[1] = M.rd[1];
	uint32_t pk[2][H2G_PK_WORDS];
	F.pk[0] = pk[0]; F.pk[1] = pk[1]; F.pk_stride = 1;
	static int64_t sc_[2 * H2G_COMBINE_MAXLEN];
	F.sc = sc_; F.sc_stride = 1;
	for(int k = 0; k < 2 + (int)FB_COUNT; k++) stats[k] = 0;
	if((e->dg.linear != 0) == (FG_GRAPH != 0)) { stats[1] = ~0ull; delete ws; return; }      // this library's fast path is built for the other kind of index
#if FG_GRAPH
	static GraphWS fgws_;
	F.alts = &e->dalts; F.gws = &fgws_;
#endif
	F.O.rout = fr.data(); F.O.aln = f1.data(); F.O.aln_slots = slots; F.O.pout = fp.data(); F.O.paln[0] = f1.data(); F.O.paln[1] = f2.data(); F.O.pair_slots = slots;
	uint32_t words[FW_TOTAL];
	FWords W; W.hot = words; W.hot_stride = 1; W.cold = words + FW_HOT;
	for(int k = 0; k < 2 + (int)FB_COUNT; k++) stats[k] = 0;
	uint32_t nbad = 0;
	for(uint32_t i = 0; i < n; i++) {
		F.name[0] = names1 + noffs1[i]; F.namelen[0] = noffs1[i + 1] - noffs1[i];
		F.name[1] = paired ? names2 + noffs2[i] : nullptr; F.namelen[1] = paired ? noffs2[i + 1] - noffs2[i] : 0;
		memset(words, 0xa5, sizeof words);
		bool ok = fg_pack_read(F.rd[0], i, pk[0], 1);
		if(paired) ok = fg_pack_read(F.rd[1], i, pk[1], 1) && ok;
		FState S;
		memset(&S, 0xa5, sizeof S);
		const bool done = fast_run_single(F, S, W, i, paired, ok);
		if(done_flags) done_flags[i] = done ? 1 : 0;
		if(!done) { stats[2 + (S.bail < FB_COUNT ? S.bail : FB_OTHER)]++; continue; }
		stats[0]++;
		// the machine on the same read
		M.name[0] = F.name[0]; M.namelen[0] = F.namelen[0]; M.name[1] = F.name[1]; M.namelen[1] = F.namelen[1];
		MachOut O; O.rout = nullptr; O.aln = nullptr; O.aln_slots = 0; O.pout = nullptr; O.paln[0] = O.paln[1] = nullptr; O.pair_slots = 0;
		bool same = true;
		if(paired) {
			std::vector<PairOut> tmp(n ? 1 : 1);
			// the machine writes pout[i]: give it a window that starts at -i
			PairOut one; O.pout = &one - i; O.paln[0] = m1.data() - (size_t)i * slots; O.paln[1] = m2.data() - (size_t)i * slots; O.pair_slots = slots;
			mach_run_single(C, M, i, true, O);
			const PairOut& f = fp[i];
			same = one.nres[0] == f.nres[0] && one.nres[1] == f.nres[1] && one.npairs == f.npairs && one.overflow == f.overflow && one.nrank == f.nrank &&
			       one.nsteps == f.nsteps && one.depth == f.depth && one.nside == f.nside && one.rnd_state == f.rnd_state &&
			       memcmp(one.pair_i, f.pair_i, sizeof one.pair_i) == 0 && memcmp(one.pair_j, f.pair_j, sizeof one.pair_j) == 0;
			for(uint32_t k = 0; same && k < f.nres[0]; k++) same = rec_equal(f1[(size_t)i * slots + k], ws->m[0].res[k]);
			for(uint32_t k = 0; same && k < f.nres[1]; k++) same = rec_equal(f2[(size_t)i * slots + k], ws->m[1].res[k]);
		} else {
			ReadOut one; O.rout = &one - i; O.aln = m1.data() - (size_t)i * slots; O.aln_slots = slots;
			mach_run_single(C, M, i, false, O);
			const ReadOut& f = fr[i];
			same = one.nres == f.nres && one.nselect == f.nselect && one.overflow == f.overflow && one.nrank == f.nrank && one.nsteps == f.nsteps &&
			       one.depth == f.depth && one.nside == f.nside && one.best == f.best && one.secbest == f.secbest && one.best_h2 == f.best_h2 &&
			       one.secbest_h2 == f.secbest_h2;
			for(uint32_t k = 0; same && k < f.nselect; k++) same = one.select[k] == f.select[k];
			for(uint32_t k = 0; same && k < f.nselect; k++) {
				const h2g_alnres &a = f1[(size_t)i * slots + k], &b = m1[k];
				same = a.fw == b.fw && a.tidx == b.tidx && a.toff == b.toff && a.len == b.len && a.trim5 == b.trim5 && a.trim3 == b.trim3 && a.nedits == b.nedits &&
				       a.splicescore == b.splicescore && a.score == b.score && memcmp(a.edits, b.edits, a.nedits * sizeof(h2g_edit)) == 0;
			}
		}
		if(!same) {
			stats[1]++; if(nbad < cap) bad_ids[nbad++] = i;
			if(getenv("H2GEMU_FAST_VERBOSE") && paired) {
				const PairOut& f = fp[i];
				PairOut one; MachOut O2; O2.rout = nullptr; O2.aln = nullptr; O2.aln_slots = 0; O2.pout = &one - i; O2.paln[0] = m1.data() - (size_t)i * slots; O2.paln[1] = m2.data() - (size_t)i * slots; O2.pair_slots = slots;
				mach_run_single(C, M, i, true, O2);
				fprintf(stderr, "pair %u: fast nres %u/%u npairs %u nrank %u nsteps %u depth %u nside %u rnd %08x | machine nres %u/%u npairs %u nrank %u nsteps %u depth %u nside %u rnd %08x\n", i,
				        f.nres[0], f.nres[1], f.npairs, f.nrank, f.nsteps, f.depth, f.nside, f.rnd_state, one.nres[0], one.nres[1], one.npairs, one.nrank, one.nsteps, one.depth, one.nside, one.rnd_state);
				for(int m = 0; m < 2; m++) {
					for(uint32_t k = 0; k < f.nres[m]; k++) { const h2g_alnres& a = (m ? f2 : f1)[(size_t)i * slots + k]; fprintf(stderr, "   fast m%d r%u fw %u toff %u len %u trim %u/%u ned %u score %lld\n", m, k, a.fw, a.toff, a.len, a.trim5, a.trim3, a.nedits, (long long)a.score); }
					for(uint32_t k = 0; k < one.nres[m]; k++) { const AlnRec& a = ws->m[m].res[k]; fprintf(stderr, "   mach m%d r%u fw %u toff %u len %u trim %u/%u ned %u score %lld\n", m, k, a.fw, a.toff, a.len, a.trim5, a.trim3, a.nedits, (long long)a.score); }
				}
			}
		}
	}
	delete ws;
}

// development: trips (primitive requests) of the general machine per read, by primitive — the length of a hard read's latency chain.
// out[nids * 16]: [op] = requests of that primitive, [15] = control phases (mach_step calls)
void h2gemu_pair_trips(Emu* e, const uint8_t* codes2, const uint32_t* offs2, const char* names1, const uint32_t* noffs1, const char* names2,
                       const uint32_t* noffs2, const uint32_t* ids, size_t nids, uint32_t* out) {
	AlnParams P; AlnCtx C;
	emu_ctx(e, 1, &P, &C);
	const bool paired = codes2 != nullptr;
	AlignWS* ws = new AlignWS();
	Mach M;
	M.ws = ws; M.rd[0] = e->reads(); M.rd[1] = M.rd[0];
	if(paired) { M.rd[1].codes = codes2; M.rd[1].offs = offs2; M.rd[1].quals = nullptr; }
	std::vector<h2g_alnres> m1(64), m2(64);
	for(size_t k = 0; k < nids; k++) {
		const uint32_t i = ids[k];
		uint32_t* o = out + k * 16;
		for(int j = 0; j < 16; j++) o[j] = 0;
		M.name[0] = names1 + noffs1[i]; M.namelen[0] = noffs1[i + 1] - noffs1[i];
		M.name[1] = paired ? names2 + noffs2[i] : nullptr; M.namelen[1] = paired ? noffs2[i + 1] - noffs2[i] : 0;
		MachOut O; O.rout = nullptr; O.aln = nullptr; O.aln_slots = 0; O.pout = nullptr; O.paln[0] = O.paln[1] = nullptr; O.pair_slots = 0;
		PairOut onep; ReadOut oner;
		if(paired) { O.pout = &onep - i; O.paln[0] = m1.data() - (size_t)i * 64; O.paln[1] = m2.data() - (size_t)i * 64; O.pair_slots = 64; }
		else { O.rout = &oner - i; O.aln = m1.data() - (size_t)i * 64; O.aln_slots = 64; }
		M.out = &O; M.paired_input = paired;
		mach_begin(M, i, paired);
		while(M.L.pc != PC_FINISHED || M.L.op != OP_NONE) {
			mach_step(C, M);
			o[15]++;
			if(M.L.op != OP_NONE) { if(M.L.op < 15) o[M.L.op]++; mach_exec(C, M, M.L.op); }
		}
		M.L.pc = PC_IDLE;
	}
	delete ws;
}

// glf1_top_fused (one LF step of one row from sides held in registers) against map_glf1_nochar, and gw_walk_single against gw_resolve,
// on `n` seeded rows of the global graph index and of the graph local indexes.  Returns the number of differing rows.
uint64_t h2gemu_glf_fused_check(Emu* e, uint32_t n, uint64_t seed) {
	if(e->dg.linear) return ~0ull;
	uint64_t bad = 0, x = seed * 0x9e3779b97f4a7c15ull + 1;
	auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
	static GraphWS gws;
	for(uint32_t i = 0; i < n; i++) {
		const uint32_t row = (uint32_t)(rnd() % e->dg.gbwtLen);
		if(is_zoff(e->dg, row)) continue;
		GRange r;
		map_glf1_nochar(e->dg, row, &r);
		uint32_t t = 0, nd = 0;
		glf1_top_fused(e->dg, row, &t, &nd);
		if(t != r.top || nd != r.node_top) { bad++; continue; }
		for(int c = 0; c < 4; c++) {                                  // map_glf1_fused == map_glf1, whatever character is asked for
			GRange a, b;
			const bool oa = map_glf1(e->dg, row, c, &a), ob = map_glf1_fused(e->dg, row, c, &b);
			if(oa != ob || a.top != b.top || a.bot != b.bot || a.node_top != b.node_top || a.node_bot != b.node_bot) { bad++; break; }
		}
		// a walk from (row r.top, node r.node_top)
		uint32_t nelt = 0;
		const bool ok = gw_resolve(e->dg, &gws.gw, r.top, r.top + 1, r.node_top, r.node_top + 1, nullptr, 1, &nelt);
		uint32_t wr = r.top, wn = r.node_top, steps = 0, off = 0;
		bool found = false;
		for(int chunk = 0; chunk < 100000 && !found; chunk++) found = gw_walk_single(e->dg, &wr, &wn, &steps, 7, &off);
		if(!ok || !found || off != gws.gw.offs[0] || steps != gws.gw.nsteps) bad++;
	}
	for(uint32_t i = 0; i < n; i++) {
		const uint32_t li = (uint32_t)(rnd() % e->dls.n);
		const DLocalDesc& d = e->dls.desc[li];
		if(d.len == 0 || local_is_linear(d)) continue;
		const LGfm lx = lgfm_of(e->dls, d);
		const uint32_t row = (uint32_t)(rnd() % d.gbwtLen);
		if(is_zoff(lx, row)) continue;
		GRange r;
		map_glf1_nochar(lx, row, &r);
		uint32_t t = 0, nd = 0;
		glf1_top_fused(lx, row, &t, &nd);
		if(t != r.top || nd != r.node_top) { bad++; continue; }
		for(int c = 0; c < 4; c++) {
			GRange a, b;
			const bool oa = map_glf1(lx, row, c, &a), ob = map_glf1_fused(lx, row, c, &b);
			if(oa != ob || a.top != b.top || a.bot != b.bot || a.node_top != b.node_top || a.node_bot != b.node_bot) { bad++; break; }
		}
		uint32_t nelt = 0;
		const bool ok = gw_resolve(lx, &gws.gw, r.top, r.top + 1, r.node_top, r.node_top + 1, nullptr, 1, &nelt);
		uint32_t wr = r.top, wn = r.node_top, steps = 0, off = 0;
		bool found = false;
		for(int chunk = 0; chunk < 100000 && !found; chunk++) found = gw_walk_single(lx, &wr, &wn, &steps, 5, &off);
		if(!ok || !found || off != gws.gw.offs[0] || steps != gws.gw.nsteps) bad++;
	}
	return bad;
}

// the staged LF step (h2g_graph_staged.h) against glf1_top_fused: random rows of the global and of the local graph indexes, and walks of `steps` steps
// with four rows in flight (the stages of all four issued stage by stage, as k_glf_chain does); returns the number of differences
uint64_t h2gemu_glf_staged_check(Emu* e, uint32_t n, uint32_t steps, uint64_t seed) {
	if(e->dg.linear) return ~0ull;
	uint64_t bad = 0, x = seed * 0x9e3779b97f4a7c15ull + 1;
	auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
	for(uint32_t i = 0; i < n; i++) {
		uint32_t row[4], ref[4];
		for(int k = 0; k < 4; k++) { do row[k] = (uint32_t)(rnd() % e->dg.gbwtLen); while(is_zoff(e->dg, row[k])); ref[k] = row[k]; }
		for(uint32_t s = 0; s < steps; s++) {
			GlfStage st[4];
			uint32_t t[4], nd[4];
			for(int k = 0; k < 4; k++) glf_stage_a(e->dg, row[k], st[k]);
			for(int k = 0; k < 4; k++) glf_stage_b(e->dg, st[k]);
			for(int k = 0; k < 4; k++) glf_stage_c(e->dg, st[k]);
			for(int k = 0; k < 4; k++) glf_stage_d(e->dg, st[k], &t[k], &nd[k]);
			for(int k = 0; k < 4; k++) {
				uint32_t rt = 0, rn = 0;
				glf1_top_fused(e->dg, ref[k], &rt, &rn);
				if(rt != t[k] || rn != nd[k]) bad++;
				row[k] = ref[k] = (rt < e->dg.gbwtLen && !is_zoff(e->dg, rt)) ? rt : (uint32_t)(rnd() % e->dg.gbwtLen);
				while(is_zoff(e->dg, row[k])) row[k] = ref[k] = (uint32_t)(rnd() % e->dg.gbwtLen);
			}
		}
	}
	// the searches' step (a required character) in stages, four rows at a time, every character
	for(uint32_t i = 0; i < n / 4; i++) {
		uint32_t row[4];
		for(int k = 0; k < 4; k++) row[k] = (uint32_t)(rnd() % e->dg.gbwtLen);      // ('$' rows included: the step must fail on them)
		for(int c = 0; c < 4; c++) {
			GlfSearchStage q[4];
			GRange a[4];
			bool oa[4];
			for(int k = 0; k < 4; k++) glf_stage_a(e->dg, row[k], q[k].s);
			for(int k = 0; k < 4; k++) glf_search_stage_b(e->dg, row[k], c, q[k]);
			for(int k = 0; k < 4; k++) glf_search_stage_c(e->dg, q[k]);
			for(int k = 0; k < 4; k++) oa[k] = glf_search_stage_d(e->dg, q[k], &a[k]);
			for(int k = 0; k < 4; k++) {
				GRange b;
				const bool ob = map_glf1_fused(e->dg, row[k], c, &b);
				if(oa[k] != ob || a[k].top != b.top || a[k].bot != b.bot || a[k].node_top != b.node_top || a[k].node_bot != b.node_bot) bad++;
			}
		}
	}
	// four coordinate walks advanced together (gw_walk_multi, in chunks of 5 steps) against four gw_walk_single
	for(uint32_t i = 0; i < n / 8; i++) {
		uint32_t row[4], node[4], steps[4] = {0, 0, 0, 0}, off[4] = {0, 0, 0, 0};
		uint32_t r1[4], n1[4];
		for(int k = 0; k < 4; k++) {
			uint32_t r0;
			do r0 = (uint32_t)(rnd() % e->dg.gbwtLen); while(is_zoff(e->dg, r0));
			glf1_top_fused(e->dg, r0, &row[k], &node[k]);             // a (row, node) pair that belongs together
			if(row[k] >= e->dg.gbwtLen) { k--; continue; }
			r1[k] = row[k]; n1[k] = node[k];
		}
		uint32_t done = 0;
		for(int chunk = 0; chunk < 100000 && done != 15u; chunk++) done = gw_walk_multi<4>(e->dg, row, node, steps, 5, off, done);
		for(int k = 0; k < 4; k++) {
			uint32_t s1 = 0, o1 = 0;
			bool found = false;
			for(int chunk = 0; chunk < 100000 && !found; chunk++) found = gw_walk_single(e->dg, &r1[k], &n1[k], &s1, 7, &o1);
			if(done != 15u || !found || o1 != off[k] || s1 != steps[k]) bad++;
		}
	}
	for(uint32_t i = 0; i < n; i++) {
		const uint32_t li = (uint32_t)(rnd() % e->dls.n);
		const DLocalDesc& d = e->dls.desc[li];
		if(d.len == 0 || local_is_linear(d)) continue;
		const LGfm lx = lgfm_of(e->dls, d);
		const uint32_t row = (uint32_t)(rnd() % d.gbwtLen);
		if(is_zoff(lx, row)) continue;
		GlfStage st;
		uint32_t t = 0, nd = 0, rt = 0, rn = 0;
		glf_stage_a(lx, row, st); glf_stage_b(lx, st); glf_stage_c(lx, st); glf_stage_d(lx, st, &t, &nd);
		glf1_top_fused(lx, row, &rt, &rn);
		if(rt != t || rn != nd) bad++;
	}
	return bad;
}

// load_local_pack (straight from the files, threaded) against pack_local(load_host_index(..)): 0 = byte-identical
int h2gemu_local_pack_check(Emu* e, const char* base) {
	LocalPack a;
	const int rc = load_local_pack(base, e->host.g.nPat, a, 3);
	if(e->host.local.empty()) return rc == -1 ? 0 : 100;          // an index without local files
	if(rc != 0) return 1;
	const LocalPack& b = e->lp;
	if(a.desc.size() != b.desc.size() || memcmp(a.desc.data(), b.desc.data(), a.desc.size() * sizeof(DLocalDesc)) != 0) return 2;
	if(a.sides != b.sides) return 3;
	if(a.words != b.words) return 4;
	if(a.first != b.first) return 5;
	if(a.zoffs != b.zoffs) return 6;
	if(a.ftabChars != b.ftabChars || a.offRate != b.offRate) return 7;
	return 0;
}

// What the device loader does since round 6 — plan_local_pack + local_fill_sides / local_fill_words: any byte range of the two packed arrays straight out of the files, nothing
// assembled on the host — against load_local_pack's arrays, in chunks of several (odd) sizes; and the global index's three big arrays as views into the mapped files (BigViews)
// against the copies a plain load_host_index makes.  0 = byte-identical.
int h2gemu_local_fill_check(Emu* e, const char* base) {
	LocalPack a;
	if(!e->host.local.empty()) {
		if(load_local_pack(base, e->host.g.nPat, a, 3) != 0) return 1;
		LocalPack b; LocalPlan pl;
		if(plan_local_pack(base, e->host.g.nPat, b, pl) != 0) return 2;
		if(a.desc.size() != b.desc.size() || memcmp(a.desc.data(), b.desc.data(), a.desc.size() * sizeof(DLocalDesc)) != 0) return 3;
		if(a.first != b.first || a.zoffs != b.zoffs || a.ftabChars != b.ftabChars || a.offRate != b.offRate) return 4;
		if(a.sides.size() != pl.nsides_tot + 256 || a.words.size() != pl.nwords_tot + 64) return 5;
		const size_t chunks[4] = {(size_t)1 << 20, 999983, 4096 + 7, (size_t)16 << 20};
		std::vector<uint8_t> tmp((size_t)16 << 20);
		for(size_t ch : chunks) {
			for(size_t off = 0; off < a.sides.size(); off += ch) {
				const size_t len = a.sides.size() - off < ch ? a.sides.size() - off : ch;
				local_fill_sides(pl, tmp.data(), off, len);
				if(memcmp(tmp.data(), a.sides.data() + off, len) != 0) return 6;
			}
			const size_t wb = a.words.size() * 2;
			for(size_t off = 0; off < wb; off += ch) {
				const size_t len = wb - off < ch ? wb - off : ch;
				local_fill_words(pl, tmp.data(), off, len);
				if(memcmp(tmp.data(), (const uint8_t*)a.words.data() + off, len) != 0) return 7;
			}
		}
	}
	HostIndex h2;
	BigViews bv;
	if(load_host_index(base, false, h2, false, &bv) != 0) return 8;
	const HostIndex& h1 = e->host;
	if(bv.sides_n != h1.g.sides.size() || memcmp(bv.sides, h1.g.sides.data(), bv.sides_n) != 0) return 9;
	if(bv.offs_n != h1.g.offs.size() * 4 || memcmp(bv.offs, h1.g.offs.data(), bv.offs_n) != 0) return 10;
	if(bv.buf_n + 16 != h1.r.buf.size() || memcmp(bv.buf, h1.r.buf.data(), bv.buf_n) != 0) return 11;
	if(!h2.g.sides.empty() || !h2.g.offs.empty() || !h2.r.buf.empty()) return 12;          // (nothing was copied)
	if(h2.g.ftab != h1.g.ftab || h2.names != h1.names || h2.r.rec_start != h1.r.rec_start || h2.g.rstarts != h1.g.rstarts) return 13;
	return 0;
}

#ifdef H2G_MEMPROF
void mp_report(unsigned nreads, const char** opnames, int nops);
void h2gemu_memprof_report(unsigned nreads) {
	static const char* ops[] = {"(none)", "PSEARCH", "GCOORDS", "EXTEND", "LSEARCH", "LCOORDS", "GSEARCH", "COMBINE", "ADJUST", "ADJMEMBER", "SW", "FINISH"};
	mp_report(nreads, ops, 12);
}
#endif

}

// ---- the splice-site database kept up to date wave by wave (merge_splice_db) against the one built from scratch over the same sites
// in order of first appearance (build_splice_db): `sites` arrive in `nwaves` slices; within a later slice a site may repeat an earlier
// one with a smaller read id (the caller passes the updated entry, as h2g_cli.cpp does).  Returns the number of differing entries.
extern "C" uint64_t h2gemu_splice_db_merge_check(const h2g_splice_site* sites, size_t n, uint32_t nPat, uint32_t nwaves) {
	using namespace h2g;
	std::vector<h2g_splice_site> seen;                                          // the caller's list: first appearance order, smallest id per temporary site
	HostSpliceDB inc;
	uint64_t bad = 0;
	for(uint32_t w = 0; w < nwaves; w++) {
		const size_t a = n * w / nwaves, b = n * (w + 1) / nwaves;
		std::vector<h2g_splice_site> delta;
		for(size_t i = a; i < b; i++) {
			const h2g_splice_site& x = sites[i];
			size_t at = seen.size();
			for(size_t k = 0; k < seen.size(); k++) if(seen[k].tidx == x.tidx && seen[k].left == x.left && seen[k].right == x.right && seen[k].dir == x.dir) { at = k; break; }
			if(at == seen.size()) { seen.push_back(x); delta.push_back(x); }
			else if(!seen[at].fromfile && x.readid < seen[at].readid) { seen[at].readid = x.readid; delta.push_back(seen[at]); }
		}
		if(w == 0) build_splice_db(seen.data(), seen.size(), nPat, inc);                 // (h2g_index_set_splice_sites: the whole list)
		else merge_splice_db(inc, delta.data(), delta.size(), nPat);
		HostSpliceDB full;
		build_splice_db(seen.data(), seen.size(), nPat, full);
		if(inc.fw.size() != full.fw.size() || inc.bw.size() != full.bw.size() || inc.fw_first != full.fw_first || inc.bw_first != full.bw_first) { bad++; continue; }
		for(size_t k = 0; k < full.fw.size(); k++) {
			bad += memcmp(&inc.fw[k], &full.fw[k], sizeof(DSpliceSite)) != 0;
			bad += memcmp(&inc.bw[k], &full.bw[k], sizeof(DSpliceSite)) != 0;
		}
	}
	return bad;
}

// ---- the 32-base chunk accessors of the word-wise comparison loops (SeqView::chunk32, RefCursor::chunk32) against the base-by-base
// accessors they stand in for: every view position of every read on both strands, and `nref` seeded reference positions per contig.
// Returns the number of differing bases.
extern "C" uint64_t h2gemu_chunk_check(Emu* e, uint32_t nref, uint64_t seed) {
	using namespace h2g;
	uint64_t bad = 0;
	DReads rd = e->reads();
	for(uint32_t i = 0; i < rd.n; i++) {
		uint32_t pk[H2G_PK_WORDS];
		if(!fg_pack_read(rd, i, pk, 1)) continue;
		for(int fw = 0; fw < 2; fw++) {
			SeqView plain = seq_view(rd, i, fw != 0);
			SeqView packed = plain;
			packed.pk = pk; packed.pk_stride = 1; packed.pk_nomask = true;
			for(uint32_t p = 0; p < plain.len; p++) {
				const uint64_t c = packed.chunk32(p);
				for(uint32_t j = 0; j < 32 && p + j < plain.len; j++) bad += (uint32_t)((c >> (2 * j)) & 3) != (uint32_t)plain.at(p + j);
			}
		}
	}
	uint64_t x = seed * 0x9e3779b97f4a7c15ull + 1;
	for(uint32_t t = 0; t < e->dr.nrefs; t++) {
		RefCursor rc, rc2;
		rc.init(&e->dr, t); rc2.init(&e->dr, t);
		const uint32_t len = e->dr.refLens[t];
		for(uint32_t k = 0; k < nref && len > 0; k++) {
			x ^= x << 13; x ^= x >> 7; x ^= x << 17;
			const int64_t pos = (int64_t)(x % len);
			const uint32_t n = 1 + (uint32_t)((x >> 40) % 32);
			if(!rc.covers(pos, n)) continue;                                   // (touches an ambiguous stretch or the contig's end: the loops take over)
			const uint64_t c = rc.chunk32(pos);
			for(uint32_t j = 0; j < n; j++) bad += (int)((c >> (2 * j)) & 3) != rc2.get(pos + j);
		}
	}
	return bad;
}
