/*
 * ref_probe — TEST INFRASTRUCTURE.  Drives the *real* reference classes
 * (compiled from /root/reference where it lies; nothing copied) to emit golden
 * vectors for the hot-path primitives of SURVEY.md §8(a):
 *
 *   rank     GFM::mapLF(SideLocus, c)            gfm.h:3712  (a5)  + rowL gfm.h:3615
 *   ftab     GFM::ftabLoHi                       gfm.h:2670  (a10)
 *   offset   GFM::tryOffset / getOffset walk     gfm.h:2719, group_walk.h (a14)
 *   j2t      GFM::joinedToTextOff                gfm.h:5527  (a15)
 *   stretch  BitPairReference::getStretch        reference.cpp:486 (a17)
 *   psearch  HI_Aligner::partialSearch           hi_aligner.h:6361 (a11)
 *   coords   HI_Aligner::getGenomeCoords         hi_aligner.h:5774 (a14)
 *   extend   GenomeHit::extend                   hi_aligner.h:2031 (a18/a19)
 *
 * Built by oracle/Makefile.ref into oracle/_ref/ref_probe (git-ignored).  Output is
 * line-oriented text on stdout; tests/gen_golden.py turns it into tests/golden/.
 * Never linked into, called by, or shipped with the product library.
 */
#include <iostream>
#include <fstream>
#include <string>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>

#include "alphabet.h"
#include "assert_helpers.h"
#include "endian_swap.h"
#include "hgfm.h"
#include "rfm.h"
#include "reference.h"
#include "read.h"
#include "scoring.h"
#include "aln_sink.h"
#include "hi_aligner.h"
#include "spliced_aligner.h"
#include "splice_site.h"
#include "aligner_sw.h"
#include "tp.h"
#include "gp.h"

using namespace std;

typedef uint32_t index_t;
typedef uint16_t local_index_t;

// Globals the reference expects its main program to define (hisat2.cpp).
MemoryTally gMemTally;
bool gMate1fw = true, gMate2fw = false, gColor = false;
int gTrim3 = 0, gTrim5 = 0;
extern void initializeCntLut();
extern void initializeCntBit();

static uint64_t splitmix64(uint64_t& s) {
	uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

struct Probe {
	ALTDB<index_t>* altdb;
	RepeatDB<index_t>* repeatdb;
	HGFM<index_t, local_index_t>* gfm;
	BitPairReference* ref;
	Probe(const string& base) {
		altdb = new ALTDB<index_t>();
		repeatdb = new RepeatDB<index_t>();
		gfm = new HGFM<index_t, local_index_t>(
			base, altdb, NULL, NULL, -1, true, -1, 0, false, false, false,
			true, true, true, true, false, false, false, false, false, false);
		gfm->loadIntoMemory(-1, true, true, true, true, false);
		ref = new BitPairReference(base, NULL, false, false, NULL, NULL, false,
		                           false, false, false, false, false);
	}
};

static Scoring* makeScoring(SimpleFunc& scoreMin, SimpleFunc& nCeil,
                            SimpleFunc& canIL, SimpleFunc& noncanIL, bool nospliced)
{
	scoreMin.init(SIMPLE_FUNC_LINEAR, 0.0f, -0.2f);
	nCeil.init(SIMPLE_FUNC_LINEAR, 0.0f, std::numeric_limits<double>::max(), 2.0f, 0.1f);
	if(nospliced) {
		canIL.init(SIMPLE_FUNC_LOG, -8, 1);
		noncanIL.init(SIMPLE_FUNC_LOG, -8, 1);
	} else {
		canIL.init(SIMPLE_FUNC_LOG, -8, 1);
		noncanIL.init(SIMPLE_FUNC_LOG, -8, 1);
	}
	return new Scoring(
		DEFAULT_MATCH_BONUS, DEFAULT_MM_PENALTY_TYPE, DEFAULT_MM_PENALTY_MAX,
		DEFAULT_MM_PENALTY_MIN, DEFAULT_SC_PENALTY_MAX, DEFAULT_SC_PENALTY_MIN,
		scoreMin, nCeil, DEFAULT_N_PENALTY_TYPE, DEFAULT_N_PENALTY, DEFAULT_N_CAT_PAIR,
		DEFAULT_READ_GAP_CONST, DEFAULT_REF_GAP_CONST, DEFAULT_READ_GAP_LINEAR,
		DEFAULT_REF_GAP_LINEAR, 4, 0, 12, 1000000, &canIL, &noncanIL);
}

/** Parse a FASTA file of reads into Read objects (names = text after '>'). */
static void loadReads(const char* fn, vector<Read*>& rds) {
	ifstream in(fn);
	string line, name, seq;
	uint64_t id = 0;
	while(true) {
		bool ok = (bool)getline(in, line);
		if(!ok || (!line.empty() && line[0] == '>')) {
			if(!seq.empty()) {
				Read* r = new Read();
				r->name.install(name.c_str());
				r->patFw.installChars(seq.c_str(), seq.size());
				string q(seq.size(), 'I');
				r->qual.install(q.c_str(), q.size());
				r->rdid = id++;
				r->mate = 0;
				r->finalize();
				rds.push_back(r);
			}
			if(!ok) break;
			name = line.substr(1);
			seq.clear();
		} else {
			seq += line;
		}
	}
}

int main(int argc, char** argv) {
	if(argc < 3) {
		cerr << "usage: ref_probe <cmd> <index_base> [args]" << endl;
		return 2;
	}
	string cmd = argv[1], base = argv[2];
	if(cmd == "probscore") {
		// probscore <unused> <n> <seed>: the splice-site probability tables (splice_site.cpp:45-105) as hex floats at n seeded
		// indexes each, then SpliceSiteDB::probscore on n seeded (donor, acceptor) sequences
		init_junction_prob();
		uint64_t n = strtoull(argv[3], NULL, 10), s = strtoull(argv[4], NULL, 10);
		for(uint64_t i = 0; i < n; i++) {
			uint64_t h = splitmix64(s);
			uint32_t a = (uint32_t)(h % (1u << (donor_len << 1))), b = (uint32_t)((h >> 20) % (1u << (acceptor_len1 << 1))), c = (uint32_t)((h >> 40) % (1u << (acceptor_len2 << 1)));
			float fa = donor_prob_sum[a], fb = acceptor_prob_sum1[b], fc = acceptor_prob_sum2[c];
			uint32_t ua, ub, uc; memcpy(&ua, &fa, 4); memcpy(&ub, &fb, 4); memcpy(&uc, &fc, 4);
			int64_t dseq = (int64_t)a, aseq = ((int64_t)b << (acceptor_len2 << 1)) | c;
			float ps = SpliceSiteDB::probscore(dseq, aseq);
			uint32_t up; memcpy(&up, &ps, 4);
			printf("%u %u %u %08x %08x %08x %08x\n", a, b, c, ua, ub, uc, up);
		}
		return 0;
	}
	initializeCntLut();
	initializeCntBit();
	Probe p(base);
	const GFM<index_t>& gfm = *p.gfm;
	const GFMParams<index_t>& gh = gfm.gh();
	if(cmd == "params") {
		printf("len %u gbwtLen %u numNodes %u lineRate %d offRate %d ftabChars %d eftabLen %u linear %d sideSz %u sideGbwtSz %u sideGbwtLen %u numSides %u offsLen %u nPat %u nFrag %u\n",
		       gh._len, gh._gbwtLen, gh._numNodes, gh._lineRate, gh._offRate, gh._ftabChars,
		       gh._eftabLen, (int)gh.linearFM(), gh._sideSz, gh._sideGbwtSz, gh._sideGbwtLen,
		       gh._numSides, gh._offsLen, (unsigned)gfm.nPat(), (unsigned)gfm.nFrag());
		return 0;
	}
	if(cmd == "rank") {
		// rank <base> <n> <seed>: rows ~ U[0,gbwtLen), c = hash&3 -> mapLF(row,c), rowL(row)
		uint64_t n = strtoull(argv[3], NULL, 10), s = strtoull(argv[4], NULL, 10);
		for(uint64_t i = 0; i < n; i++) {
			uint64_t h = splitmix64(s);
			index_t row = (index_t)(h % gh._gbwtLen);
			int c = (int)((h >> 40) & 3);
			SideLocus<index_t> l;
			l.initFromRow(row, gh, gfm.gfm());
			index_t r = gfm.mapLF(l, c);
			int rl = gfm.rowL(l);
			printf("%u %d %u %d\n", row, c, r, rl);
		}
		return 0;
	}
	if(cmd == "ftab") {
		// ftab <base> <n> <seed>: random k-mers (some with N) -> ftabLoHi
		uint64_t n = strtoull(argv[3], NULL, 10), s = strtoull(argv[4], NULL, 10);
		int fc = gh._ftabChars;
		for(uint64_t i = 0; i < n; i++) {
			BTDnaString seq;
			string txt;
			uint64_t h = splitmix64(s);
			for(int j = 0; j < fc; j++) {
				int c = (int)((h >> (2 * j)) & 3);
				if(((h >> 50) & 63) == 0 && j == (int)((h >> 56) % fc)) c = 4;
				seq.append(c);
				txt += "ACGTN"[c];
			}
			index_t top = 0, bot = 0;
			bool ok = gfm.ftabLoHi(seq, 0, false, top, bot);
			printf("%s %d %u %u\n", txt.c_str(), (int)ok, top, bot);
		}
		return 0;
	}
	if(cmd == "offset") {
		// offset <base> <n> <seed>: random rows -> text offset in joined string + joinedToTextOff(len 20)
		uint64_t n = strtoull(argv[3], NULL, 10), s = strtoull(argv[4], NULL, 10);
		for(uint64_t i = 0; i < n; i++) {
			uint64_t h = splitmix64(s);
			index_t row = (index_t)(h % gh._gbwtLen);
			index_t off = gfm.getOffset(row, row);
			index_t tidx = 0, toff = 0, tlen = 0;
			bool straddled = false;
			index_t qlen = (index_t)((h >> 40) % 120) + 1;
			bool ok = false;
			if(off != (index_t)INDEX_MAX && off < gh._len)
				ok = gfm.joinedToTextOff(qlen, off, tidx, toff, tlen, false, straddled);
			printf("%u %u %u %d %u %u %u %d\n", row, off, qlen, (int)ok, tidx, toff, tlen, (int)straddled);
		}
		return 0;
	}
	if(cmd == "stretch") {
		// stretch <base> <n> <seed>: random windows -> bases 0..4
		uint64_t n = strtoull(argv[3], NULL, 10), s = strtoull(argv[4], NULL, 10);
		SStringExpandable<char> buf;
		SStringExpandable<uint32_t> destU32;
		for(uint64_t i = 0; i < n; i++) {
			uint64_t h = splitmix64(s);
			index_t tidx = (index_t)((h >> 48) % gfm.nPat());
			index_t tlen = gfm.plen()[tidx];
			index_t toff = (index_t)(h % (tlen + 40));
			if(((h >> 44) & 15) == 0) toff = (index_t)((h >> 20) % 64);
			index_t cnt = (index_t)((h >> 32) % 150) + 1;
			buf.resize(cnt + 32);
			buf.fill(4);
			int off = p.ref->getStretch(reinterpret_cast<uint32_t*>(buf.wbuf()), tidx, toff, cnt
			                            ASSERT_ONLY(, destU32));
			printf("%u %u %u ", tidx, toff, cnt);
			for(index_t j = 0; j < cnt; j++) putchar("ACGTN"[(int)buf.wbuf()[off + j]]);
			putchar('\n');
		}
		return 0;
	}
	if(cmd == "lglf") {
		// lglf <base> <n> <seed>: mapGLF / mapGLF1 on random LOCAL graph indexes (LocalGFM<local_index_t>, 128 B sides of u16 words)
		uint64_t n = strtoull(argv[3], NULL, 10), s = strtoull(argv[4], NULL, 10);
		EList<pair<local_index_t, local_index_t> > iedges;
		for(uint64_t i = 0; i < n; i++) {
			uint64_t h = splitmix64(s);
			index_t tidx = (index_t)((h >> 48) % gfm.nPat());
			index_t tlen = gfm.plen()[tidx];
			index_t toff = (index_t)((h >> 8) % tlen);
			const LocalGFM<local_index_t, index_t>* l = p.gfm->getLocalGFM(tidx, toff);
			if(l == NULL || l->gh()._len == 0) continue;
			const GFMParams<local_index_t>& lh = l->gh();
			local_index_t top = (local_index_t)(h % (lh._gbwtLen - 1));
			int c = (int)((h >> 40) & 3);
			if((h >> 63) & 1) {
				SideLocus<local_index_t> loc;
				loc.initFromRow(top, lh, l->gfm());
				if(((h >> 50) & 7) != 0) c = l->rowL(loc);
				pair<local_index_t, local_index_t> nr(0, 0);
				pair<local_index_t, local_index_t> r = l->mapGLF1(top, loc, c, &nr);
				printf("1 %u %u %u 0 %d %u %u %u %u 0\n", tidx, toff, (unsigned)top, c, (unsigned)r.first, (unsigned)r.second, (unsigned)nr.first, (unsigned)nr.second);
				continue;
			}
			uint32_t spread = (uint32_t)((h >> 44) % 300) + 2;
			if(((h >> 60) & 3) == 0) spread = (uint32_t)((h >> 44) % 6) + 2;
			uint32_t bot = (uint32_t)top + spread;
			if(bot > lh._gbwtLen) bot = lh._gbwtLen;
			if(bot <= (uint32_t)top + 1) continue;
			SideLocus<local_index_t> tl, bl;
			SideLocus<local_index_t>::initFromTopBot(top, (local_index_t)bot, lh, l->gfm(), tl, bl);
			pair<local_index_t, local_index_t> nr(0, 0);
			iedges.clear();
			pair<local_index_t, local_index_t> r = l->mapGLF(tl, bl, c, &nr, &iedges, 10);
			printf("0 %u %u %u %u %d %u %u %u %u %u", tidx, toff, (unsigned)top, bot, c, (unsigned)r.first, (unsigned)r.second, (unsigned)nr.first, (unsigned)nr.second, (unsigned)iedges.size());
			for(size_t e = 0; e < iedges.size(); e++) printf(" %u:%u", (unsigned)iedges[e].first, (unsigned)iedges[e].second);
			putchar('\n');
		}
		return 0;
	}
	if(cmd == "glf" || cmd == "glf1") {
		// glf  <base> <n> <seed>: random ranges [top, top+spread) and c -> mapGLF (graph LF on a range, gfm.h:3759)
		// glf1 <base> <n> <seed>: random row and c -> mapGLF1 (gfm.h:3957)
		uint64_t n = strtoull(argv[3], NULL, 10), s = strtoull(argv[4], NULL, 10);
		EList<pair<index_t, index_t> > iedges;
		for(uint64_t i = 0; i < n; i++) {
			uint64_t h = splitmix64(s);
			index_t top = (index_t)(h % (gh._gbwtLen - 1));
			int c = (int)((h >> 40) & 3);
			if(cmd == "glf1") {
				SideLocus<index_t> l;
				l.initFromRow(top, gh, gfm.gfm());
				if(((h >> 50) & 7) != 0) c = gfm.rowL(l);
				pair<index_t, index_t> nr(0, 0);
				pair<index_t, index_t> r = gfm.mapGLF1(top, l, c, &nr);
				printf("%u %d %u %u %u %u\n", top, c, r.first, r.second, nr.first, nr.second);
				continue;
			}
			index_t spread = (index_t)((h >> 44) % 400) + 2;
			if(((h >> 60) & 3) == 0) spread = (index_t)((h >> 44) % 6) + 2;
			index_t bot = top + spread;
			if(bot > gh._gbwtLen) bot = gh._gbwtLen;
			if(bot <= top + 1) continue;
			SideLocus<index_t> tl, bl;
			SideLocus<index_t>::initFromTopBot(top, bot, gh, gfm.gfm(), tl, bl);
			pair<index_t, index_t> nr(0, 0);
			iedges.clear();
			pair<index_t, index_t> r = gfm.mapGLF(tl, bl, c, &nr, &iedges, 10);
			printf("%u %u %d %u %u %u %u %u", top, bot, c, r.first, r.second, nr.first, nr.second, (unsigned)iedges.size());
			for(size_t e = 0; e < iedges.size(); e++) printf(" %u:%u", iedges[e].first, iedges[e].second);
			putchar('\n');
		}
		return 0;
	}
	if(cmd == "extsearch") {
		// extsearch <base> <reads.fa> <queries.txt>: globalGFMSearch (hi_aligner.h:6606) / localGFMSearch (:6751) as hybridSearch_recur calls
		// them.  Query line: read fw rdoff kind(0 global | 1 local) tidx toff maxHitLen uniqueStop; output line: nelt hitlen top bot uniqueStop
		vector<Read*> rds;
		loadReads(argv[3], rds);
		SimpleFunc scoreMin, nCeil, canIL, noncanIL;
		Scoring* sc = makeScoring(scoreMin, nCeil, canIL, noncanIL, true);
		bool linear = gh.linearFM();
		int khits = linear ? 5 : 10;
		ReportingParams rp(khits, std::max(5, khits * 2), 0, 0, true, true, true, false, false, 0, false, false);
		HI_Aligner<index_t, local_index_t> al(gfm, true, 0);
		RandomSource rnd;
		rnd.init(0);
		ifstream qf(argv[4]);
		unsigned ri, fwv, rdoff, kind, tidx, toff, maxHitLen, us;
		while(qf >> ri >> fwv >> rdoff >> kind >> tidx >> toff >> maxHitLen >> us) {
			Read& rd = *rds[ri];
			bool uniqueStop = us != 0;
			index_t hitlen = 0;
			if(kind == 0) {
				index_t top = (index_t)INDEX_MAX, bot = (index_t)INDEX_MAX, ntop = top, nbot = bot;
				EList<pair<index_t, index_t> > ie;
				index_t nelt = al.globalGFMSearch(gfm, rd, *sc, rp, fwv != 0, rdoff, hitlen, top, bot, ntop, nbot, ie, rnd, uniqueStop);
				printf("%u %u %u %u %d\n", nelt, hitlen, top, bot, (int)uniqueStop);
			} else {
				const LocalGFM<local_index_t, index_t>* l = p.gfm->getLocalGFM(tidx, toff);
				local_index_t top = (local_index_t)INDEX_MAX, bot = (local_index_t)INDEX_MAX, ntop = top, nbot = bot;
				EList<pair<local_index_t, local_index_t> > lie;
				index_t nelt = l == NULL ? 0 : al.localGFMSearch(*l, rd, *sc, rp, fwv != 0, rdoff, hitlen, top, bot, ntop, nbot, lie, rnd, uniqueStop, 8, (local_index_t)maxHitLen);
				printf("%u %u %u %u %d\n", nelt, hitlen, (unsigned)top, (unsigned)bot, (int)uniqueStop);
			}
		}
		return 0;
	}
	if(cmd == "combine") {
		// combine <base> <reads.fa> <nospliced:0|1>: GenomeHit::combineWith (hi_aligner.h:1420-2025) on two partial alignments of a read whose
		// names carry them: ">id|fw|tidx|rdoffA|lenA|toffA|rdoffB|lenB|toffB".  Both hits are GenomeHit::init'ed exact anchors (the generator keeps
		// them free of mismatches); what lies between them on the read — mismatches, an insertion, a deletion, an intron — is combineWith's to place.
		vector<Read*> rds;
		loadReads(argv[3], rds);
		bool nospliced = (argc > 4) ? atoi(argv[4]) != 0 : true;
		SimpleFunc scoreMin, nCeil, canIL, noncanIL;
		Scoring* sc = makeScoring(scoreMin, nCeil, canIL, noncanIL, nospliced);
		RandomSource rnd;
		EList<string> refnames;
		SpliceSiteDB ssdb(*p.ref, refnames, false, false, false);
		SwAligner swa;
		SwMetrics swm;
		SharedTempVars<index_t> sharedVars;
		init_junction_prob();
		for(size_t ri = 0; ri < rds.size(); ri++) {
			Read& rd = *rds[ri];
			unsigned id, fw, tidx, roA, lenA, toA, roB, lenB, toB;
			if(sscanf(rd.name.toZBuf(), "%u|%u|%u|%u|%u|%u|%u|%u|%u", &id, &fw, &tidx, &roA, &lenA, &toA, &roB, &lenB, &toB) != 9) continue;
			index_t rdlen = (index_t)rd.length();
			TAlScore minsc = (TAlScore)scoreMin.f<double>((double)rdlen);
			if(minsc > 0) minsc = 0;
			// joined offsets of the two anchors (rstarts: joined start, text id, text offset of every fragment)
			auto joined = [&](index_t t, index_t off) -> index_t {
				const index_t* rs = gfm.rstarts();
				for(index_t f = 0; f < gfm.nFrag(); f++) {
					if(rs[f * 3 + 1] != t || rs[f * 3 + 2] > off) continue;
					index_t flen = (f + 1 < gfm.nFrag() ? rs[(f + 1) * 3] : gh._len) - rs[f * 3];
					if(off - rs[f * 3 + 2] < flen) return rs[f * 3] + (off - rs[f * 3 + 2]);
				}
				return 0;
			};
			GenomeHit<index_t> a, b;
			a.init(fw != 0, roA, lenA, 0, 0, tidx, toA, joined(tidx, toA), sharedVars);
			b.init(fw != 0, roB, lenB, 0, 0, tidx, toB, joined(tidx, toB), sharedVars);
			rnd.init((uint32_t)(id * 13 + 1));
			bool ok = a.combineWith(b, rd, gfm, *p.ref, *p.altdb, *p.repeatdb, ssdb, swa, swm, *sc, minsc, rnd, (index_t)8,
			                        (index_t)20, (index_t)500000, (index_t)7, (index_t)14, (index_t)16, NULL, nospliced);
			printf("%u %d %lld -> %d %u %u %u %lld %u", id, (int)fw, (long long)minsc, (int)ok, a.rdoff(), a.len(), a.refoff(), (long long)a.score(), (unsigned)a.edits().size());
			for(size_t e = 0; e < a.edits().size(); e++) {
				const Edit& ed = a.edits()[e];
				if(ed.type == EDIT_TYPE_SPL) printf(" %u:S:%u:%d:%d", ed.pos, ed.splLen, (int)ed.splDir, (int)ed.knownSpl);
				else printf(" %u:%c>%c:%d", ed.pos, (char)ed.chr, (char)ed.qchr, (int)ed.type);
			}
			putchar('\n');
		}
		return 0;
	}
	if(cmd == "psearch" || cmd == "lsearch" || cmd == "coords" || cmd == "extend" || cmd == "sw" || cmd == "adjust") {
		// <cmd> <base> <reads.fa> <nospliced:0|1>
		vector<Read*> rds;
		loadReads(argv[3], rds);
		bool nospliced = (argc > 4) ? atoi(argv[4]) != 0 : true;
		SimpleFunc scoreMin, nCeil, canIL, noncanIL;
		Scoring* sc = makeScoring(scoreMin, nCeil, canIL, noncanIL, nospliced);
		bool linear = gh.linearFM();
		int khits = linear ? 5 : 10;
		ReportingParams rp(khits, std::max(5, khits * 2), 0, 0, true, true, true,
		                   false, false, 0, false, false);
		HI_Aligner<index_t, local_index_t> al(gfm, true, 0);
		RandomSource rnd;
		rnd.init(0);
		WalkMetrics wlm;
		PerReadMetrics prm;
		HIMetrics him;
		TranscriptomePolicy tpol(20, 500000, 7, 14, nospliced, false, false, false, false);
		GraphPolicy gpol(16, false, false, false);
		EList<string> refnames;
		SpliceSiteDB ssdb(*p.ref, refnames, false, false, false);
		SwAligner swa;
		SwMetrics swm;
		SharedTempVars<index_t> sharedVars;
		for(size_t ri = 0; ri < rds.size(); ri++) {
			Read& rd = *rds[ri];
			index_t rdlen = (index_t)rd.length();
			TAlScore minsc = (TAlScore)scoreMin.f<double>((double)rdlen);
			if(minsc > 0) minsc = 0;
			if(cmd == "sw" && argc > 5) minsc = (TAlScore)atoll(argv[5]);   // sw <base> <reads> <nospliced> <minsc>: below -254 SwAligner::align takes its 16-bit path (aligner_sw.cpp:496)
			for(int fwi = 0; fwi < 2; fwi++) {
				bool fw = fwi == 0;
				ReadBWTHit<index_t> hit;
				hit.init(fw, rdlen);
				size_t mineFw = 0, mineRc = 0;
				bool pseudogeneStop = linear && !nospliced, anchorStop = true;
				al.partialSearch(gfm, rd, *sc, rp, fw, 0, mineFw, mineRc, hit, rnd,
				                 pseudogeneStop, anchorStop);
				BWTHit<index_t>& ph = hit.getPartialHit(hit.offsetSize() - 1);
				if(cmd == "psearch") {
					printf("%llu %d %u %u %u %u %u %u %u %u %d %u %u %d %d",
					       (unsigned long long)rd.rdid, (int)fw, ph._top, ph._bot, ph._node_top,
					       ph._node_bot, ph._bwoff, ph._len, ph._hit_type, hit._cur,
					       (int)hit._done, hit._numPartialSearch, hit._numUniqueSearch,
					       (int)pseudogeneStop, (int)anchorStop);
					if(!linear) {   // graph index: in-edge list of the hit
						printf(" %u", (unsigned)ph._node_iedge_count.size());
						for(size_t e = 0; e < ph._node_iedge_count.size(); e++) printf(" %u:%u", ph._node_iedge_count[e].first, ph._node_iedge_count[e].second);
					}
					putchar('\n');
					continue;
				}
				if(cmd == "lsearch") {
					// lsearch <base> <reads.fa> <nospliced> <tidx> <toff> <extoff> <fw>: localGFMSearch (hi_aligner.h:6751) leftwards from
					// read offset extoff on the local index covering (tidx, toff), then getGenomeCoords_local (:5861)
					if(fwi != 0) continue;
					index_t tidx = (index_t)atoi(argv[5]), toff = (index_t)atoi(argv[6]), extoff = (index_t)atoi(argv[7]);
					bool lfw = atoi(argv[8]) != 0;
					const LocalGFM<local_index_t, index_t>* l = p.gfm->getLocalGFM(tidx, toff);
					index_t extlen = 0;
					local_index_t top = (local_index_t)INDEX_MAX, bot = (local_index_t)INDEX_MAX, ntop = top, nbot = bot;
					EList<pair<local_index_t, local_index_t> > lie;
					bool uniqueStop = true;
					index_t nelt = al.localGFMSearch(*l, rd, *sc, rp, lfw, extoff, extlen, top, bot, ntop, nbot, lie, rnd, uniqueStop, 8);
					printf("%llu nelt %u extlen %u top %u bot %u node %u %u unique %d nie %u", (unsigned long long)rd.rdid, nelt, extlen, (unsigned)top, (unsigned)bot,
					       (unsigned)ntop, (unsigned)nbot, (int)uniqueStop, (unsigned)lie.size());
					for(size_t e = 0; e < lie.size(); e++) printf(" %u:%u", (unsigned)lie[e].first, (unsigned)lie[e].second);
					if(nelt > 0 && nelt <= 5) {
						EList<Coord> lco;
						bool st = false;
						al.getGenomeCoords_local(*l, *p.altdb, *p.ref, rnd, top, bot, ntop, nbot, lie, lfw, extoff + 1 - extlen, extlen, lco, wlm, prm, him, true, st);
						printf(" | %u", (unsigned)lco.size());
						for(size_t k = 0; k < lco.size(); k++) printf(" %lld:%lld:%llu", (long long)(int32_t)lco[k].ref(), (long long)lco[k].off(), (unsigned long long)lco[k].joinedOff());
					}
					putchar('\n');
					continue;
				}
				if(ph.empty() || ph._bot - ph._top > 16) continue;
				EList<Coord> coords;
				bool straddled = false;
				index_t rdoff = hit._len - ph._bwoff - ph._len;
				al.getGenomeCoords(gfm, *p.altdb, *p.ref, rnd, ph._top, ph._bot, ph._node_top,
				                   ph._node_bot, ph._node_iedge_count, fw, ph._bot - ph._top,
				                   rdoff, ph._len, coords, wlm, prm, him, false, straddled);
				if(cmd == "coords") {
					printf("%llu %d %u %u %u %u %d %u", (unsigned long long)rd.rdid, (int)fw,
					       ph._top, ph._bot, rdoff, ph._len, (int)straddled, (unsigned)coords.size());
					for(size_t k = 0; k < coords.size(); k++)
						printf(" %lld:%lld:%llu", (long long)(int32_t)coords[k].ref(), (long long)coords[k].off(),
						       (unsigned long long)coords[k].joinedOff());
					if(!linear) {   // graph index: the node range and in-edge list getGenomeCoords was called with
						printf(" | %u %u %u", ph._node_top, ph._node_bot, (unsigned)ph._node_iedge_count.size());
						for(size_t e = 0; e < ph._node_iedge_count.size(); e++) printf(" %u:%u", ph._node_iedge_count[e].first, ph._node_iedge_count[e].second);
					}
					putchar('\n');
					continue;
				}
				if(cmd == "adjust") {
					// GenomeHit::adjustWithALT (static, hi_aligner.h:2239-2390) as getAnchorHits calls it (:5175): one anchor
					// coordinate -> the GenomeHits it yields once offsets are corrected for indel ALTs and edits rewritten
					for(size_t k = 0; k < coords.size(); k++) {
						if(coords[k].ref() == (TRefId)std::numeric_limits<index_t>::max()) continue;
						EList<GenomeHit<index_t> > ghs;
						bool found = GenomeHit<index_t>::adjustWithALT(rdoff, ph._len, coords[k], sharedVars, ghs, rd, gfm, *p.altdb, *p.ref, gpol);
						printf("%llu %d %u %u %u %u %u -> %d %u", (unsigned long long)rd.rdid, (int)fw, rdoff, ph._len, (unsigned)coords[k].ref(),
						       (unsigned)coords[k].off(), (unsigned)coords[k].joinedOff(), (int)found, (unsigned)ghs.size());
						for(size_t g = 0; g < ghs.size(); g++) {
							printf(" | %u %u %u %u %u", ghs[g].rdoff(), ghs[g].len(), ghs[g].refoff(), ghs[g]._joinedOff, (unsigned)ghs[g].edits().size());
							for(size_t e = 0; e < ghs[g].edits().size(); e++) {
								const Edit& ed = ghs[g].edits()[e];
								printf(" %u:%c>%c:%d:%lld", ed.pos, (char)ed.chr, (char)ed.qchr, (int)ed.type,
								       ed.snpID == (uint32_t)INDEX_MAX ? -1LL : (long long)ed.snpID);
							}
						}
						putchar('\n');
					}
					continue;
				}
				if(cmd == "sw") {
					// the SwAligner call site of hybridSearch (spliced_aligner.h:209-262): frame the DP rectangle around the
					// seed hit, fill (8-bit end-to-end SSE), gather, one nextAlignment.  Output: the rectangle, whether an
					// alignment was found, its score, reference offset, edits (read coordinates of the aligned strand,
					// i.e. before invertEdits) and the next PRNG draw (nextAlignment reseeds rnd).
					// sw <base> <reads> <nospliced> <minsc> <shift>: with a shift, every coordinate is also tried `shift` bases to the right
					// (an unrelated placement: deep scores, many equal choices in the backtrace); those lines carry k + 100
					const size_t swshift = argc > 6 ? (size_t)atoll(argv[6]) : 0;
					for(size_t kk = 0; kk < coords.size() * (swshift ? 2 : 1); kk++) {
						const size_t k = kk % coords.size();
						const bool shifted = kk >= coords.size();
						if(coords[k].ref() == (TRefId)std::numeric_limits<index_t>::max()) continue;
						swa.initRead(rd.patFw, rd.patRc, rd.qual, rd.qualRev, 0, rd.length(), *sc);
						DynProgFramer dpframe(false);
						size_t tlen = p.ref->approxLen(coords[k].ref());
						size_t readGaps = 10, refGaps = 10, nceil = 0, maxhalf = 10;
						index_t hit_refoff = (index_t)coords[k].off() + (shifted ? (index_t)swshift : 0);
						if(shifted && (size_t)hit_refoff + rd.length() + 64 > tlen) continue;
						index_t refoff = hit_refoff > rdoff ? hit_refoff - rdoff : 0;
						DPRect rect;
						dpframe.frameSeedExtensionRect(refoff, rd.length(), tlen, readGaps, refGaps, nceil, maxhalf, rect);
						size_t cminlen = 2000, cpow2 = 4, nwindow = 10, nsInLeftShift = 0;
						swa.initRef(fw, coords[k].ref(), rect, *p.ref, tlen, *sc, minsc, true, cminlen, cpow2, false, true,
						            nwindow, nsInLeftShift);
						rnd.init((uint32_t)(rd.rdid * 7 + k + (shifted ? 100 : 0) + 1));
						TAlScore bestCell = std::numeric_limits<TAlScore>::min();
						bool found = swa.align(rnd, bestCell);
						printf("%llu %d %u %u %u %lld | %lld %lld %lld %lld %lld %lld | %d %lld",
						       (unsigned long long)rd.rdid, (int)fw, (unsigned)(k + (shifted ? 100 : 0)), (unsigned)coords[k].ref(), refoff, (long long)minsc,
						       (long long)rect.refl, (long long)rect.refr, (long long)rect.refl_pretrim, (long long)rect.refr_pretrim,
						       (long long)rect.corel, (long long)rect.corer, (int)found,
						       bestCell == std::numeric_limits<TAlScore>::min() ? -99999LL : (long long)bestCell);
						bool found2 = false;
						SwResult res;
						LinkedEList<EList<Edit> > rawEdits;
						if(found) {
							res.reset();
							res.alres.init_raw_edits(&rawEdits);
							found2 = swa.nextAlignment(res, minsc, rnd);
						}
						printf(" %d", (int)found2);
						if(found2) {
							const Coord& co = res.alres.refcoord();
							printf(" %lld %lld %u", (long long)res.alres.score().score(), (long long)co.off(), (unsigned)res.alres.ned().size());
							for(size_t e = 0; e < res.alres.ned().size(); e++) {
								const Edit& ed = res.alres.ned()[e];
								printf(" %u:%c>%c:%d", ed.pos, (char)ed.chr, (char)ed.qchr, (int)ed.type);
							}
						}
						printf(" r%u\n", rnd.nextU32());
					}
					continue;
				}
				// extend: for each coordinate, a GenomeHit extended with mm = 0, 1, 2, 3
				for(size_t k = 0; k < coords.size(); k++) {
					if(coords[k].ref() == (TRefId)std::numeric_limits<index_t>::max()) continue;
					for(index_t mm = 0; mm < 4; mm++) {
						GenomeHit<index_t> gh_;
						gh_.init(fw, rdoff, ph._len, 0, 0, (index_t)coords[k].ref(),
						         (index_t)coords[k].off(), (index_t)coords[k].joinedOff(), sharedVars);
						index_t leftext = (index_t)INDEX_MAX, rightext = (index_t)INDEX_MAX;
						bool ext = gh_.extend(rd, gfm, *p.ref, *p.altdb, *p.repeatdb, ssdb, swa, swm, prm,
						                      *sc, minsc, rnd, (index_t)8, tpol, gpol, leftext, rightext, mm);
						printf("%llu %d %u %u %u %u %u %u -> %d %u %u %u %u %u %u %lld %u",
						       (unsigned long long)rd.rdid, (int)fw, rdoff, ph._len,
						       (unsigned)coords[k].ref(), (unsigned)coords[k].off(),
						       (unsigned)coords[k].joinedOff(), mm,
						       (int)ext, gh_.rdoff(), gh_.len(), gh_.refoff(), gh_._joinedOff,
						       leftext, rightext, (long long)gh_.score(), (unsigned)gh_.edits().size());
						for(size_t e = 0; e < gh_.edits().size(); e++) {
							const Edit& ed = gh_.edits()[e];
							if(linear) printf(" %u:%c>%c", ed.pos, (char)ed.chr, (char)ed.qchr);
							else printf(" %u:%c>%c:%d:%lld", ed.pos, (char)ed.chr, (char)ed.qchr, (int)ed.type,
							            ed.snpID == (uint32_t)INDEX_MAX ? -1LL : (long long)ed.snpID);   // graph: + type and snpID
						}
						putchar('\n');
					}
				}
			}
		}
		return 0;
	}
	cerr << "unknown command " << cmd << endl;
	return 2;
}
