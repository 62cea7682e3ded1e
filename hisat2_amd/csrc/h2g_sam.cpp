// h2g_sam.cpp — host-side SAM emission for device-produced alignments (include/h2g_sam.h).
// Plain C++ (no HIP): compiled by the host compiler and linked into libh2g.so.  Each function cites the reference code
// whose output it reproduces byte for byte (hisat2 2.2.3, default print options hisat2.cpp:371-409, --mapq-v 2).
#include <stdint.h>
#include <string.h>
#include <ctype.h>
#include <limits.h>
#include <math.h>
#include <string>
#include <vector>
#include <map>
#include <array>
#include <algorithm>
#include <thread>
#include "../../include/h2g_sam.h"
#include "h2g_host_index.h"
#include "h2g_splice_db_host.h"

using namespace h2g;

struct Met {            // ReportingMetrics (aln_sink.h:51-160), the counters printAlSumm reads
	uint64_t nread = 0, npaired = 0, nunpaired = 0, nconcord_0 = 0, nconcord_uni1 = 0, nconcord_uni2 = 0, ndiscord = 0;
	uint64_t nunp_0_0 = 0, nunp_0_uni1 = 0, nunp_0_uni2 = 0, nunp_0 = 0, nunp_uni1 = 0, nunp_uni2 = 0;
	std::vector<h2g_splice_site> novel;                  // splice sites of the lines written (h2g_sam_collect_novel_sites), in read order
	void add(const Met& o) {
		novel.insert(novel.end(), o.novel.begin(), o.novel.end());
		nread += o.nread; npaired += o.npaired; nunpaired += o.nunpaired; nconcord_0 += o.nconcord_0; nconcord_uni1 += o.nconcord_uni1;
		nconcord_uni2 += o.nconcord_uni2; ndiscord += o.ndiscord; nunp_0_0 += o.nunp_0_0; nunp_0_uni1 += o.nunp_0_uni1; nunp_0_uni2 += o.nunp_0_uni2;
		nunp_0 += o.nunp_0; nunp_uni1 += o.nunp_uni1; nunp_uni2 += o.nunp_uni2;
	}
};
struct h2g_sam {
	mutable Met met;                                      // summed over every format call (single caller thread)
	std::vector<std::string> refnames;
	std::vector<uint32_t>    reflens;
	std::vector<HostAlt>     alts;
	std::vector<std::string> altnames;
	int threads = 1;
	bool no_unal = false;                                 // --no-unal: SamConfig::omitUnalignedReads
	bool secondary = false;                               // --secondary: selectByScore keeps lower-scoring alignments too
	uint32_t smType = 2;                                  // --score-min (MAPQ's scMin), default L,0,-0.2
	double smConst = 0.0, smCoeff = (double)(-0.2f);
	std::vector<h2g_splice_site> alt_sites;               // the splice-site ALTs of a --ss index (SpliceSiteDB::read(gfm, alts)): always in the database
	h2g::HostSpliceDB ssdb;                               // h2g_sam_set_splice_sites: TLEN of concordant pairs leaves known introns out
	uint32_t ssdb_window = 0;
	int rna_strandness = 0;                               // --rna-strandness: 0 unknown, 1 F, 2 R, 3 FR, 4 RF (read.h:30)
	bool collect_novel = false;                           // h2g_sam_collect_novel_sites
	std::string rg_id, rg_fields, rg_optflag;             // --rg-id / --rg: "\tID:x", "\tSM:y...", "RG:Z:x" (hisat2.cpp:1389-1407)
	bool new_summary = false;                             // --new-summary (aln_sink.h:1659)
	bool no_sq = false, omit_sec_seq = false;             // --no-sq (hisat2.cpp:4130), --omit-sec-seq (aln_sink.h:3190)
	bool report_discordant = true, report_mixed = true;   // --no-discordant / --no-mixed clear them (ReportingParams::discord / mixed aln_sink.h:272)
	bool tlen_adjust = true;                              // --no-templatelen-adjustment clears it (aln_sink.h:2070-2076)
	const h2g_edit* long_edits = nullptr; size_t n_long_edits = 0;   // h2g_sam_set_long_edits: the caller's buffer (not owned), of the batch being formatted
	// what SpliceSiteDB keeps per site for --novel-splicesite-outfile (splice_site.cpp:243-276): the number of lines written across it
	// and the smallest edit distance among them; file / index sites enter with 0 / 0 (SpliceSite::init splice_site.h:237)
	struct SiteStat { uint64_t numreads = 0; uint32_t editdist = 0; };
	mutable std::map<std::array<uint32_t, 4>, SiteStat> site_stats;   // (text, left, right, dir), the order SpliceSitePos sorts in
	void count_sites(const std::vector<h2g_splice_site>& v) const {
		for(const h2g_splice_site& x : v) {
			auto r = site_stats.emplace(std::array<uint32_t, 4>{x.tidx, x.left, x.right, (uint32_t)x.dir}, SiteStat{1, x.editdist});
			if(!r.second) { r.first->second.numreads++; if(x.editdist < r.first->second.editdist) r.first->second.editdist = x.editdist; }
		}
	}
	void register_sites(const h2g_splice_site* v, size_t n) const {       // SpliceSiteDB::read: from the index or a file
		for(size_t i = 0; i < n; i++) if(v[i].fromfile) site_stats.emplace(std::array<uint32_t, 4>{v[i].tidx, v[i].left, v[i].right, (uint32_t)v[i].dir}, SiteStat{0, 0});
	}
	uint64_t first_read_id = 0;                           // Read::rdid of read 0 of the next format call
};

namespace {

enum { EDIT_READ_GAP = 1, EDIT_REF_GAP = 2, EDIT_MM = 3, EDIT_SPL = 5 };        // edit.h:36-42
// splice edits keep splLen / splDir in chr, qchr, pad (include/h2g.h)
inline uint32_t spl_len(const h2g_edit& e) { return (uint32_t)e.chr | ((uint32_t)e.qchr << 8) | (((uint32_t)e.pad & 15u) << 16); }
inline uint32_t spl_dir(const h2g_edit& e) { return ((uint32_t)e.pad >> 4) & 7u; }
enum { ALT_SGL = 1, ALT_INS = 2, ALT_DEL = 3 };                                 // alt.h:31-39
enum { PAIR_CONCORD_M1 = 1, PAIR_CONCORD_M2, PAIR_DISCORD_M1, PAIR_DISCORD_M2, PAIR_UNP_M1, PAIR_UNP_M2, PAIR_UNPAIRED };   // aligner_result.h:404-412

struct Score {        // AlnScore (aligner_result.h:44-330): score, then hisat2_score (fewer trimmed bases wins)
	bool valid = false;
	int64_t score = 0, h2 = 0;
	bool gt(const Score& o) const { if(!o.valid) return valid; if(!valid) return false; return score > o.score || (score == o.score && h2 > o.h2); }
	bool eq(const Score& o) const { return valid && o.valid && score == o.score && h2 == o.h2; }
};
// AlnScore::calculate_hisat2_score aligner_result.h:322-350: score, repeat (never), transcript (1 = spliced: near splice sites,
// reportHit hi_aligner.h:6100-6143), splice score (mean intron length of the short-anchored splices / 100), trimmed bases
// A record with more edits than its H2G_MAX_EDITS inline entries (a long deletion: one edit per base, edit.h) keeps them in the long-edit area
// of its batch (h2g_align_fetch_long_edits -> h2g_sam_set_long_edits): edits[0].pos is their offset there.  Every reader goes through ED().
thread_local const h2g_edit* tl_long_edits = nullptr;
thread_local size_t tl_long_edits_n = 0;
inline const h2g_edit* ED(const h2g_alnres& r) {
	if(r.nedits <= H2G_MAX_EDITS) return r.edits;
	static const h2g_edit none[1] = {};
	if(!tl_long_edits || (size_t)r.edits[0].pos + r.nedits > tl_long_edits_n) return none;      // (format_* refuses such a batch before any line is written)
	return tl_long_edits + r.edits[0].pos;
}
int64_t hisat2_score(int64_t sc, uint32_t trim, int transcript = 0, uint32_t splicescore = 0) {
	if(sc > INT32_MAX) sc = INT32_MAX; else if(sc < INT32_MIN) sc = INT32_MIN;
	const int64_t t = trim > 0xFFFF ? 0 : 0xFFFF - (int64_t)trim;
	int64_t spl = (int64_t)splicescore / 100;
	spl = spl > 255 ? 0 : 255 - spl;
	return (int64_t)(((uint64_t)sc << 32) | ((uint64_t)transcript << 24) | ((uint64_t)spl << 16) | (uint64_t)t);
}
Score score_of(const h2g_alnres& r) {
	Score s; s.valid = true; s.score = r.score;
	int transcript = 0;                  // 2: every splice is a database site, 1: spliced (GenomeHit::spliced hi_aligner.h:1086)
	bool all_known = true;
	for(uint32_t i = 0; i < r.nedits; i++) if(ED(r)[i].type == EDIT_SPL) { transcript = 1; all_known = all_known && (ED(r)[i].pad >> 7) != 0; }
	if(transcript && all_known) transcript = 2;
	s.h2 = hisat2_score(r.score, r.trim5 + r.trim3, transcript, r.splicescore);
	return s;
}
Score add(const Score& a, const Score& b) { Score s; s.valid = a.valid; s.score = a.score + b.score; s.h2 = a.h2 + b.h2; return s; }

struct Flags {        // AlnFlags (aligner_result.h:414-640); the filters are "passed" bits
	int  pairing = PAIR_UNPAIRED;
	bool primary = true, oppAligned = false, oppFw = true, nfilt = true, lenfilt = true;
	bool partOfPair() const { return pairing < PAIR_UNPAIRED; }
	bool concordant() const { return pairing == PAIR_CONCORD_M1 || pairing == PAIR_CONCORD_M2; }
	bool discordant() const { return pairing == PAIR_DISCORD_M1 || pairing == PAIR_DISCORD_M2; }
	bool unpairedMate() const { return pairing == PAIR_UNP_M1 || pairing == PAIR_UNP_M2; }
	bool readMate1() const { return pairing == PAIR_CONCORD_M1 || pairing == PAIR_DISCORD_M1 || pairing == PAIR_UNP_M1; }
};
struct Summ {         // AlnSetSumm (aligner_result.cpp:1167-1260)
	bool paired = false;
	Score best[2], secbest[2], bestPaired, secbestPaired;
	uint64_t numAlns[2] = {0, 0}, numAlnsPaired = 0;
	int64_t orefid = -1, orefoff = -1;
};
struct Rd { const char* name; uint32_t namelen; const uint8_t* codes; uint32_t len; const char* qual; };

void put(std::string& o, int64_t v) {     // itoa10
	char b[24];
	int k = 24;
	uint64_t u = v < 0 ? (uint64_t)(-(v + 1)) + 1 : (uint64_t)v;
	do { b[--k] = (char)('0' + u % 10); u /= 10; } while(u);
	if(v < 0) b[--k] = '-';
	o.append(b + k, 24 - k);
}

// Scoring::nFilter / length filter as the worker applies them (hisat2.cpp:3404-3440): which YF:Z flag an unaligned read gets
void read_filters(const Rd& r, bool* lenfilt, bool* nfilt) {
	*lenfilt = r.len >= 2;                     // rdlens <= multiseedMms(0) || < 2  => filtered
	uint32_t ns = 0;
	for(uint32_t i = 0; i < r.len; i++) ns += r.codes[i] > 3;
	*nfilt = ns <= (uint32_t)(0.0 + (double)0.15f * (double)r.len);
}

// BowtieMapq2::mapq unique.h:187-403 (end-to-end branch; canMax = false, exhausted = false)
int mapq_v2(const h2g_sam& S, const Summ& s, bool mate1, uint32_t rdlen, uint32_t ordlen) {
	const int m = mate1 ? 0 : 1;
	const Score& bst = s.paired ? s.bestPaired : s.best[m];
	const Score& sec = s.paired ? s.secbestPaired : s.secbest[m];
	const bool hasSecbest = sec.valid;
	const bool equalSecbest = hasSecbest && bst.eq(sec);
	if(!hasSecbest || !equalSecbest) return 60;
	auto scmin = [&](uint32_t len) {                     // scoreMin_.f<TAlScore>((float)rdlen) simple_func.h:88
		const double x = (double)(float)len;
		const double X = S.smType == 2 ? x : S.smType == 3 ? sqrt(x) : S.smType == 4 ? log(x) : 0.0;
		return (int64_t)(S.smConst + S.smCoeff * X);
	};
	int64_t scPer = 0;                                    // monotone scoring: perfect score 0
	int64_t scMin = scmin(rdlen);
	if(s.paired) scMin += scmin(ordlen);
	const int64_t diff = scPer - scMin;
	const int64_t best = bst.score, bestOver = best - scMin;
	const int64_t secb = sec.score;
	int64_t bestdiff = llabs(llabs(best) - llabs(secb));
	const double d = (double)diff;
	int ret;
	if(bestdiff >= d * (double)0.9f)      ret = bestOver == diff ? 39 : 33;
	else if(bestdiff >= d * (double)0.8f) ret = bestOver == diff ? 38 : 27;
	else if(bestdiff >= d * (double)0.7f) ret = bestOver == diff ? 37 : 26;
	else if(bestdiff >= d * (double)0.6f) ret = bestOver == diff ? 36 : 22;
	else if(bestdiff >= d * (double)0.5f) ret = bestOver == diff ? 35 : bestOver >= d * (double)0.84f ? 25 : bestOver >= d * (double)0.68f ? 16 : 5;
	else if(bestdiff >= d * (double)0.4f) ret = bestOver == diff ? 34 : bestOver >= d * (double)0.84f ? 21 : bestOver >= d * (double)0.68f ? 14 : 4;
	else if(bestdiff >= d * (double)0.3f) ret = bestOver == diff ? 32 : bestOver >= d * (double)0.88f ? 18 : bestOver >= d * (double)0.67f ? 15 : 3;
	else if(bestdiff >= d * (double)0.2f) ret = bestOver == diff ? 31 : bestOver >= d * (double)0.88f ? 17 : bestOver >= d * (double)0.67f ? 11 : 0;
	else if(bestdiff >= d * (double)0.1f) ret = bestOver == diff ? 30 : bestOver >= d * (double)0.88f ? 12 : bestOver >= d * (double)0.67f ? 7 : 0;
	else if(bestdiff > 0)                 ret = bestOver >= d * (double)0.67f ? 6 : 2;
	else                                  ret = bestOver >= d * (double)0.67f ? 1 : 0;
	return ret;
}

struct Ed { uint32_t pos; char chr, qchr; uint8_t type; uint32_t snp; uint32_t skip; };
// Edit::invertPoss(edits, sz, false) edit.cpp:69-96
void invert(std::vector<Ed>& e, uint32_t sz) {
	std::reverse(e.begin(), e.end());
	for(auto& x : e) x.pos = (x.type == EDIT_READ_GAP || x.type == EDIT_SPL) ? sz - x.pos : sz - x.pos - 1;
}
struct Stacked { std::string ref, rel, read; std::vector<uint8_t> snp; std::vector<uint32_t> skip; uint32_t trimLS = 0, trimRS = 0; };
// AlnRes::initStacked aligner_result.h:1856 + StackedAln::init aligner_result.cpp:660-728 + leftAlign(false) :746-791
void stack_alignment(const h2g_alnres& r, const std::string& seq /* aligned strand, ASCII */, Stacked& st) {
	static thread_local std::vector<Ed> ed;
	ed.resize(r.nedits);
	for(uint32_t i = 0; i < r.nedits; i++) { ed[i].pos = ED(r)[i].pos; ed[i].chr = (char)ED(r)[i].chr; ed[i].qchr = (char)ED(r)[i].qchr; ed[i].type = ED(r)[i].type; ed[i].snp = ED(r)[i].snp; ed[i].skip = ED(r)[i].type == EDIT_SPL ? spl_len(ED(r)[i]) : 0; }
	// h2g_alnres trims are those of the GenomeHit (left / right of the aligned strand) == trimLS / trimRS after the swap
	st.trimLS = r.trim5; st.trimRS = r.trim3;
	const uint32_t len_trimmed = (uint32_t)seq.size() - st.trimLS - st.trimRS;
	if(!r.fw) invert(ed, len_trimmed);
	st.ref.clear(); st.rel.clear(); st.read.clear(); st.snp.clear(); st.skip.clear();
	size_t rdoff = st.trimLS;
	auto match_to = [&](size_t pos) { while(rdoff < pos) { const char c = seq[rdoff++]; st.ref.push_back(c); st.rel.push_back('='); st.snp.push_back(0); st.read.push_back(c); } };
	for(const Ed& e : ed) {
		match_to(e.pos + st.trimLS);
		const uint8_t sn = e.snp != 0xffffffffu;
		if(e.type == EDIT_SPL)           { st.ref.push_back('N'); st.rel.push_back('N'); st.snp.push_back(0); st.read.push_back('N'); st.skip.push_back(e.skip); }
		else if(e.type == EDIT_MM)            { const char c = seq[rdoff++]; st.ref.push_back(e.chr); st.rel.push_back('X'); st.snp.push_back(sn); st.read.push_back(c); }
		else if(e.type == EDIT_REF_GAP)  { const char c = seq[rdoff++]; st.ref.push_back('-');   st.rel.push_back('I'); st.snp.push_back(sn); st.read.push_back(c); }
		else if(e.type == EDIT_READ_GAP) {                               st.ref.push_back(e.chr); st.rel.push_back('D'); st.snp.push_back(sn); st.read.push_back('-'); }
	}
	match_to(seq.size() - st.trimRS);
	const size_t ln = st.ref.size();
	for(size_t i = 0; i < ln; i++) {
		const char rel = st.rel[i];
		if(rel != '=' && rel != 'X' && rel != 'N') {
			if(st.snp[i]) continue;
			size_t glen = 1;
			for(size_t j = i + 1; j < ln; j++) { if(rel != st.rel[j]) break; glen++; }
			size_t l = i - 1, rr = l + glen;
			std::string& gp = rel == 'I' ? st.ref : st.read;
			const std::string& ngp = rel == 'I' ? st.read : st.ref;
			while(l > 0 && l < ln && ngp[l] == ngp[rr]) {
				if(st.rel[l] == 'I' || st.rel[l] == 'D') break;
				if(st.rel[l] == 'X' || st.rel[l] == 'N') break;
				std::swap(gp[l], gp[rr]);
				std::swap(st.rel[l], st.rel[rr]);
				l--; rr--;
			}
			i += glen - 1;
		}
	}
}
// StackedAln::buildCigar(false) + writeCigar aligner_result.cpp:796-843, 898
void write_cigar(const Stacked& st, std::string& o) {
	if(st.trimLS > 0) { put(o, st.trimLS); o.push_back('S'); }
	const size_t ln = st.rel.size();
	size_t nskip = 0;
	for(size_t i = 0; i < ln; i++) {
		char op = st.rel[i];
		if(op == 'X' || op == '=') op = 'M';
		size_t run = 1;
		if(op != 'N') {
			for(; i + run < ln; run++) { char op2 = st.rel[i + run]; if(op2 == 'X' || op2 == '=') op2 = 'M'; if(op2 != op) break; }
			i += run - 1;
		} else run = st.skip[nskip++];
		put(o, (int64_t)run); o.push_back(op);
	}
	if(st.trimRS > 0) { put(o, st.trimRS); o.push_back('S'); }
}
// StackedAln::buildMdz + writeMdz aligner_result.cpp:848-893, 935-1000
void write_mdz(const Stacked& st, std::string& o) {
	bool mm_last = false, rdgap_last = false, first_print = true;
	const size_t ln = st.rel.size();
	for(size_t i = 0; i < ln; i++) {
		const char op = st.rel[i];
		if(op == '=') {
			size_t run = 1, nins = 0;
			for(; i + run < ln; run++) { if(st.rel[i + run] == '=') {} else if(st.rel[i + run] == 'I' || st.rel[i + run] == 'N') nins++; else break; }
			i += run - 1;
			if(run - nins > 0) { put(o, (int64_t)(run - nins)); first_print = false; mm_last = false; rdgap_last = false; }
		} else if(op == 'X') {
			if(rdgap_last || mm_last || first_print) o.push_back('0');
			o.push_back(st.ref[i]);
			first_print = false; mm_last = true; rdgap_last = false;
		} else if(op == 'D') {
			if(mm_last || first_print) o.push_back('0');
			if(!rdgap_last) o.push_back('^');
			o.push_back(st.ref[i]);
			first_print = false; mm_last = false; rdgap_last = true;
		}
	}
	if(mm_last || rdgap_last) o.push_back('0');
}

void seq_ascii(const Rd& r, bool fw, std::string& s, std::string& q) {
	s.resize(r.len); q.resize(r.len);
	for(uint32_t i = 0; i < r.len; i++) {
		const uint32_t k = fw ? i : r.len - 1 - i;
		const uint8_t c = r.codes[k];
		s[i] = "ACGTN"[fw ? (c > 4 ? 4 : c) : (c > 3 ? 4 : 3 - c)];
		q[i] = r.qual ? r.qual[k] : 'I';
	}
}
// SamConfig::printReadName sam.h:233-256 (truncQname_ = true)
void put_read_name(std::string& o, const Rd& r, bool omitSlashMate) {
	size_t n = r.namelen;
	if(omitSlashMate && n >= 2 && r.name[n - 2] == '/' && (r.name[n - 1] == '1' || r.name[n - 1] == '2' || r.name[n - 1] == '3')) n -= 2;
	if(n > 255) n = 255;
	for(size_t i = 0; i < n; i++) { if(isspace((unsigned char)r.name[i])) return; o.push_back(r.name[i]); }
}
// SamConfig::printRefName: the name up to the first whitespace
void put_ref_name(std::string& o, const std::string& name) { for(char c : name) { if(isspace((unsigned char)c)) break; o.push_back(c); } }

// AlnRes::setFragmentLength aligner_result.h:1631-1697 with an empty splice-site database; trims extend both ends
// (getExtendedCoords :1156).  rfextent_ does not count introns (calcRefExtent :1880) while refcoord_right() does (:1256), so
// each alignment has two (start, end) pairs — (st, en) anchored at its left end, (st2, en2) at its right end — and the
// upstream mate enters with the right-anchored pair: introns inside the mates do not count towards TLEN.
int64_t fragment_length(const h2g_alnres& me, const h2g_alnres& o, bool meMate1, const h2g_sam* S = nullptr /* concordant pairs: the splice-site database */, uint64_t rdid = 0) {
	auto coords = [](const h2g_alnres& r, int64_t& st, int64_t& en, int64_t& st2, int64_t& en2) {
		int64_t ext = r.len, spl = 0;
		for(uint32_t i = 0; i < r.nedits; i++) {
			if(ED(r)[i].type == EDIT_REF_GAP) ext--; else if(ED(r)[i].type == EDIT_READ_GAP) ext++;
			else if(ED(r)[i].type == EDIT_SPL) spl += spl_len(ED(r)[i]);
		}
		st = (int64_t)r.toff - r.trim5; en = (int64_t)r.toff + ext - 1 + r.trim3;
		st2 = st + spl; en2 = en + spl;
	};
	int64_t st, en, st2, en2, ost, oen, ost2, oen2;
	coords(me, st, en, st2, en2);
	coords(o, ost, oen, ost2, oen2);
	bool imUpstream;
	if(st < ost) imUpstream = true;
	else if(st == ost) {
		if(me.fw && o.fw && meMate1) imUpstream = true;
		else if(me.fw && !o.fw) imUpstream = true;
		else imUpstream = false;
	} else imUpstream = false;
	const int64_t up = imUpstream ? std::min(st2, ost) : std::min(st, ost2);
	const int64_t dn = imUpstream ? std::max(en2, oen) : std::max(en, oen2);
	int64_t intron_len = 0;
	if(S && !S->ssdb.fw.empty() && me.tidx < S->ssdb.fw_first.size() - 1) {   // :1669-1686: the longest database intron between the mates
		const int64_t up_right = imUpstream ? std::min(en2, oen) : std::min(en, oen2);
		const int64_t dn_left = imUpstream ? std::max(st2, ost) : std::max(st, ost2);
		if(up_right + 100 < dn_left) {
			// getRightSpliceSites(ref, up_right, dn_left - up_right): sites whose left end lies in [up_right, dn_left - 1]
			const uint32_t lo = S->ssdb.fw_first[me.tidx], hi = S->ssdb.fw_first[me.tidx + 1];
			const auto* a = S->ssdb.fw.data();
			uint32_t x = lo, y = hi;
			while(x < y) { const uint32_t m = x + ((y - x) >> 1); if((int64_t)a[m].left < up_right) x = m + 1; else y = m; }
			for(; x < hi && (int64_t)a[x].left <= dn_left - 1; x++) {
				if(!a[x].fromfile && (uint64_t)a[x].readid + S->ssdb_window > rdid) continue;
				if((int64_t)a[x].left <= up || (int64_t)a[x].right >= dn) continue;
				const int64_t il = (int64_t)a[x].right - (int64_t)a[x].left - 1;
				if(intron_len < il) intron_len = il;
			}
		}
	}
	int64_t fl = 1 + dn - up - intron_len;
	return imUpstream ? fl : -fl;
}

thread_local std::vector<h2g_splice_site>* tl_novel = nullptr;     // where this formatter thread collects the sites of its lines
thread_local uint64_t tl_rdid = 0;                                 // Read::rdid of the read being formatted
// SpliceSiteDB::addSpliceSite splice_site.cpp:190-347 (minAnchorLen 15), called for every alignment line written
// (AlnSinkSam::append aln_sink.h:1570-1582): the junctions of an untrimmed alignment whose anchors on both sides are long enough
// for the mismatches they carry.  The database keeps, per site, the smallest id of the reads that added it.
void add_splice_sites(const h2g_alnres& r, uint32_t rdlen, uint64_t rdid, std::vector<h2g_splice_site>& out) {
	if(r.trim5 + r.trim3 > 0) return;
	static thread_local std::vector<Ed> ed;
	ed.resize(r.nedits);
	for(uint32_t i = 0; i < r.nedits; i++) { ed[i].pos = ED(r)[i].pos; ed[i].chr = (char)ED(r)[i].chr; ed[i].qchr = (char)ED(r)[i].qchr; ed[i].type = ED(r)[i].type; ed[i].snp = ED(r)[i].snp; ed[i].skip = ED(r)[i].type == EDIT_SPL ? spl_len(ED(r)[i]) : 0; }
	std::vector<uint32_t> dirs(r.nedits);
	for(uint32_t i = 0; i < r.nedits; i++) dirs[i] = spl_dir(ED(r)[i]);
	if(!r.fw) { invert(ed, rdlen); std::reverse(dirs.begin(), dirs.end()); }
	const uint32_t minAnchorLen = 15, SPL_UNKNOWN = 1;
	auto is_mm_gap = [](const Ed& e) { return e.type == EDIT_MM || e.type == EDIT_READ_GAP || e.type == EDIT_REF_GAP; };
	uint32_t editdist = 0;
	for(const Ed& e : ed) editdist += is_mm_gap(e);
	uint32_t refoff = r.toff, leftAnchor = 0, rightAnchor = 0, mm = 0;
	size_t eidx = 0, last = 0;
	bool inited = false;
	h2g_splice_site ssp; memset(&ssp, 0, sizeof ssp);
	auto consider = [&](size_t e_for_dir, size_t after) {
		const uint32_t extra = dirs[e_for_dir] == SPL_UNKNOWN ? 6 : 0;
		uint32_t mm2 = 0;
		for(size_t j = after + 1; j < ed.size(); j++) if(is_mm_gap(ed[j])) mm2++;
		if(leftAnchor >= minAnchorLen + mm * 2 + extra && rightAnchor >= minAnchorLen + mm2 * 2 + extra) out.push_back(ssp);
	};
	for(uint32_t i = 0; i < rdlen; i++, refoff++) {
		while(eidx < ed.size() && ed[eidx].pos == i) {
			if(ed[eidx].type == EDIT_READ_GAP) refoff++;
			else if(ed[eidx].type == EDIT_REF_GAP) refoff--;
			if(is_mm_gap(ed[eidx])) mm++;
			if(ed[eidx].type == EDIT_SPL) {
				if(inited) {
					rightAnchor = ed[eidx].pos - ed[last].pos;
					consider(eidx, eidx);
					leftAnchor = rightAnchor; rightAnchor = 0;
				} else leftAnchor = ed[eidx].pos;
				ssp.tidx = r.tidx; ssp.left = refoff - 1; ssp.right = refoff + ed[eidx].skip; ssp.dir = (uint8_t)dirs[eidx];
				ssp.readid = (uint32_t)rdid; ssp.fromfile = 0; ssp.known = 0; ssp.editdist = (uint8_t)std::min(editdist, 255u);
				inited = true;
				refoff += ed[eidx].skip;
				last = eidx;
			}
			eidx++;
		}
	}
	if(inited) {
		rightAnchor = rdlen - ed[last].pos;
		consider(last, last);
	}
}

// AlnSinkSam::appendMate aln_sink.h:3024-3260 + printAlignedOptFlags / printEmptyOptFlags sam.h:525-1100
void append_mate(const h2g_sam& S, std::string& o, const Rd& rd, const Rd* rdo, const h2g_alnres* rs, const h2g_alnres* rso,
                 const Summ& summ, const Flags& fl, uint64_t nh)
{
	if(rs == nullptr && S.no_unal) return;                 // aln_sink.h:3040
	static thread_local Stacked st;                       // scratch reused across lines (no per-line allocations)
	static thread_local std::string seq, qual;
	seq_ascii(rd, rs == nullptr || rs->fw, seq, qual);
	// an alignment whose edits are all mismatches (nearly every line) needs no stacked form: its CIGAR is one M run between the soft
	// clips and its MD:Z follows from the mismatch positions; what buildCigar / buildMdz would make of the stacked strings is written directly
	bool simple = rs != nullptr;
	if(rs) for(uint32_t i = 0; i < rs->nedits; i++) if(ED(*rs)[i].type != EDIT_MM) { simple = false; break; }
	if(rs && !simple) stack_alignment(*rs, seq, st);
	if(rs && S.collect_novel && tl_novel) add_splice_sites(*rs, rd.len, tl_rdid, *tl_novel);
	put_read_name(o, rd, fl.partOfPair());
	o.push_back('\t');
	int f = 0;
	if(fl.partOfPair()) {
		f |= 1;
		if(fl.concordant()) f |= 2;
		if(!fl.oppAligned) f |= 8;
		f |= fl.readMate1() ? 0x40 : 0x80;
		if(fl.oppAligned && rso != nullptr && !rso->fw) f |= 0x20;
	}
	if(!fl.primary) f |= 0x100;
	if(rs && !rs->fw) f |= 0x10;
	if(!rs) f |= 4;
	put(o, f); o.push_back('\t');
	if(rs) put_ref_name(o, S.refnames[rs->tidx]);
	else if(summ.orefid != -1) put_ref_name(o, S.refnames[(size_t)summ.orefid]);
	else o.push_back('*');
	o.push_back('\t');
	if(rs) put(o, (int64_t)rs->toff + 1);
	else if(summ.orefid != -1) put(o, summ.orefoff + 1);
	else o.push_back('0');
	o.push_back('\t');
	if(rs) put(o, mapq_v2(S, summ, fl.pairing == PAIR_UNPAIRED || fl.readMate1(), rd.len, rdo ? rdo->len : 0));
	else o.push_back('0');
	o.push_back('\t');
	if(rs && simple) {
		if(rs->trim5 > 0) { put(o, rs->trim5); o.push_back('S'); }
		put(o, (int64_t)(rd.len - rs->trim5 - rs->trim3)); o.push_back('M');
		if(rs->trim3 > 0) { put(o, rs->trim3); o.push_back('S'); }
	} else if(rs) write_cigar(st, o); else o.push_back('*');
	o.push_back('\t');
	if(rs && fl.partOfPair()) {                                         // RNEXT
		if(rso && rs->tidx != rso->tidx) put_ref_name(o, S.refnames[rso->tidx]); else o.push_back('=');
	} else if(summ.orefid != -1) o.push_back('=');
	else o.push_back('*');
	o.push_back('\t');
	if(rs && fl.partOfPair()) put(o, (int64_t)(rso ? rso->toff : rs->toff) + 1);   // PNEXT
	else if(summ.orefid != -1) put(o, summ.orefoff + 1);
	else o.push_back('0');
	o.push_back('\t');
	// ISIZE: setMateParams computes it when the opposite mate is known and on the same reference (or concordant)
	if(rs && rso && summ.paired && (rs->tidx == rso->tidx || fl.concordant())) put(o, fragment_length(*rs, *rso, fl.readMate1(), fl.concordant() && S.tlen_adjust ? &S : nullptr, tl_rdid));
	else o.push_back('0');
	o.push_back('\t');
	if(!fl.primary && S.omit_sec_seq) o += "*\t*\t";                     // aln_sink.h:3190, :3206
	else { o += seq; o.push_back('\t'); o += qual; o.push_back('\t'); }
	if(!rs) {                                                            // printEmptyOptFlags sam.h:1033-1100
		o += "YT:Z:";
		o += fl.concordant() ? "CP" : fl.discordant() ? "DP" : fl.unpairedMate() ? "UP" : "UU";
		if(!fl.lenfilt) o += "\tYF:Z:LN"; else if(!fl.nfilt) o += "\tYF:Z:NS";
		if(!S.rg_optflag.empty()) { o.push_back('\t'); o += S.rg_optflag; }   // sam.h:1102
		o.push_back('\n');
		return;
	}
	o += "AS:i:"; put(o, rs->score);
	{
		const Score& sco = summ.secbest[(fl.pairing == PAIR_UNPAIRED || fl.readMate1()) ? 0 : 1];      // summ.secbestMate(rd.mate < 2)
		if(sco.valid) { o += "\tZS:i:"; put(o, sco.score); }
	}
	o += "\tXN:i:0";                                                       // refNs is never set (hi_aligner.h:6170)
	const size_t nalts = S.alts.size();
	size_t num_mm = 0, num_go = 0, num_gx = 0, NM = 0;
	for(uint32_t i = 0; i < rs->nedits; i++) {                            // on the edits as stored (5'-relative)
		const h2g_edit* e = ED(*rs);
		if(e[i].type == EDIT_SPL) continue;
		if(e[i].type == EDIT_MM) { if(e[i].snp >= nalts) num_mm++; }
		else if(e[i].type == EDIT_READ_GAP) {
			if(e[i].snp >= nalts) { num_go++; num_gx++; }
			while(i + 1 < rs->nedits && e[i + 1].pos == e[i].pos && e[i + 1].type == EDIT_READ_GAP) { i++; if(e[i].snp >= nalts) num_gx++; }
		} else if(e[i].type == EDIT_REF_GAP) {
			if(e[i].snp >= nalts) { num_go++; num_gx++; }
			while(i + 1 < rs->nedits && e[i + 1].pos == e[i].pos + 1 && e[i + 1].type == EDIT_REF_GAP) { i++; if(e[i].snp >= nalts) num_gx++; }
		}
	}
	for(uint32_t i = 0; i < rs->nedits; i++) if(ED(*rs)[i].type != EDIT_SPL && ED(*rs)[i].snp >= nalts) NM++;
	o += "\tXM:i:"; put(o, (int64_t)num_mm);
	o += "\tXO:i:"; put(o, (int64_t)num_go);
	o += "\tXG:i:"; put(o, (int64_t)num_gx);
	o += "\tNM:i:"; put(o, (int64_t)NM);
	o += "\tMD:Z:";
	if(simple) {   // writeMdz over '=' runs and 'X' cells only: a run length (0 between two mismatches and at either end), then the reference base
		const uint32_t lt = rd.len - rs->trim5 - rs->trim3;
		uint32_t at = 0;                                               // the next position of the aligned strand not yet accounted for
		for(uint32_t k = 0; k < rs->nedits; k++) {
			const h2g_edit& e = ED(*rs)[rs->fw ? k : rs->nedits - 1 - k];
			const uint32_t pos = rs->fw ? e.pos : lt - 1 - e.pos;
			if(pos > at) put(o, (int64_t)(pos - at)); else o.push_back('0');
			o.push_back((char)e.chr);
			at = pos + 1;
		}
		if(lt > at) put(o, (int64_t)(lt - at)); else o.push_back('0');
	} else write_mdz(st, o);
	if(summ.paired && rso) { o += "\tYS:i:"; put(o, rso->score); }
	o += "\tYT:Z:";
	o += fl.concordant() ? "CP" : fl.discordant() ? "DP" : fl.unpairedMate() ? "UP" : "UU";
	if(!fl.lenfilt) o += "\tYF:Z:LN"; else if(!fl.nfilt) o += "\tYF:Z:NS";
	if(!S.rg_optflag.empty()) { o.push_back('\t'); o += S.rg_optflag; }       // sam.h:780
	if(S.rna_strandness != 0) {   // a stranded library: the tag follows from the mate and the strand it aligned to (sam.h:940-966)
		char strand = '+';
		const bool m1 = fl.pairing == PAIR_UNPAIRED || fl.readMate1();   // unpaired reads are ALN_RES_TYPE_UNPAIRED_MATE1 (aln_sink.h:2368)
		const int rs_ = S.rna_strandness;
		if(m1) { if(rs->fw) { if(rs_ == 2 || rs_ == 4) strand = '-'; } else if(rs_ == 1 || rs_ == 3) strand = '-'; }
		else   { if(rs->fw) { if(rs_ == 3) strand = '-'; } else if(rs_ == 4) strand = '-'; }
		o += "\tXS:A:"; o.push_back(strand);
	} else {   // XS:A: sam.h:925-940 with AlnRes::spliced_whichsense_transcript aligner_result.h:1289 (unstranded library)
		uint32_t sense = 1;
		for(uint32_t i = 0; i < rs->nedits; i++) {
			if(ED(*rs)[i].type != EDIT_SPL) continue;
			const uint32_t d = spl_dir(ED(*rs)[i]);
			if(sense == 1) sense = d;
			else if(d != 1) {
				if((sense == 2 || sense == 4) && d != 2 && d != 4) { sense = 1; break; }
				if((sense == 3 || sense == 5) && d != 3 && d != 5) { sense = 1; break; }
			}
		}
		if(sense != 1) { o += "\tXS:A:"; o.push_back((sense == 2 || sense == 4) ? '+' : '-'); }
	}
	o += "\tNH:i:"; put(o, (int64_t)nh);
	// Zs:Z (sam.h:985-1030): the known variants the alignment goes through, positions relative to the previous one
	{
		static thread_local std::vector<Ed> ed;
		ed.resize(rs->nedits);
		for(uint32_t i = 0; i < rs->nedits; i++) { ed[i].pos = ED(*rs)[i].pos; ed[i].type = ED(*rs)[i].type; ed[i].snp = ED(*rs)[i].type == EDIT_SPL ? 0xffffffffu : ED(*rs)[i].snp /* a splice edit keeps its probscore there */; ed[i].chr = ed[i].qchr = 0; }
		const uint32_t len_trimmed = rd.len - rs->trim5 - rs->trim3;
		if(!rs->fw) invert(ed, len_trimmed);
		bool snp_first = true;
		uint32_t prev = 0xffffffffu;
		for(size_t i = 0; i < ed.size(); i++) {
			if(ed[i].snp >= nalts) continue;
			const uint32_t si = ed[i].snp;
			const HostAlt& snp = S.alts[si];
			if(si == prev) continue;
			o += snp_first ? "\tZs:Z:" : ",";
			uint64_t pos = ed[i].pos;
			size_t j = i;
			while(j > 0) {
				if(ed[j - 1].snp < nalts) {
					const HostAlt& s2 = S.alts[ed[j - 1].snp];
					if(s2.type == ALT_SGL) pos -= (ed[j - 1].pos + 1);
					else if(s2.type == ALT_DEL) pos -= ed[j - 1].pos;
					else if(s2.type == ALT_INS) pos -= (ed[j - 1].pos + snp.len);
					break;
				}
				j--;
			}
			put(o, (int64_t)pos);
			o += snp.type == ALT_SGL ? "|S|" : snp.type == ALT_DEL ? "|D|" : "|I|";
			o += S.altnames[si];
			snp_first = false;
			prev = si;
		}
	}
	o.push_back('\n');
}

// selectByScore aln_sink.h:2680-2760 on a list of AlnScore keys; RandomSource random_source.h:33
struct Rng { uint32_t last; uint32_t next() { last = 1664525u * last + 1013904223u; uint32_t r = last >> 16; last = 1664525u * last + 1013904223u; return r ^ last; } };
void select_by_score(const std::vector<Score>& keys, size_t num, Rng& rnd, std::vector<size_t>& sel, bool secondary) {
	sel.clear();
	const size_t sz = keys.size();
	if(sz < 1) return;
	if(num > sz) num = sz;
	static thread_local std::vector<std::pair<Score, size_t> > buf;      // (reused: no allocation per read)
	buf.resize(sz);
	for(size_t i = 0; i < sz; i++) buf[i] = std::make_pair(keys[i], i);
	// EList::sort of pair<AlnScore, size_t> descending: (score, h2) then index
	std::sort(buf.begin(), buf.end(), [](const std::pair<Score, size_t>& a, const std::pair<Score, size_t>& b) {
		if(a.first.score != b.first.score) return a.first.score > b.first.score;
		if(a.first.h2 != b.first.h2) return a.first.h2 > b.first.h2;
		return a.second > b.second;
	});
	auto shuffle = [&](size_t begin, size_t cnt) {
		size_t left = cnt;
		for(size_t q = begin; q + 1 < begin + cnt; q++) { const uint32_t r = rnd.next() % (uint32_t)left; if(r > 0) std::swap(buf[q], buf[q + r]); left--; }
	};
	size_t streak = 0;
	for(size_t i = 1; i < sz; i++) {
		if(buf[i].first.eq(buf[i - 1].first)) { if(streak == 0) streak = 1; streak++; }
		else { if(streak > 1) shuffle(i - streak, streak); streak = 0; }
	}
	if(streak > 1) shuffle(sz - streak, streak);
	for(size_t i = 0; i < num; i++) sel.push_back(buf[i].second);
	if(!secondary) for(size_t i = 0; i + 1 < sel.size(); i++) if(!buf[i].first.eq(buf[i + 1].first)) { sel.resize(i + 1); break; }
}

// The records of one read / mate: rows of h2g_alnres (fixed-stride and dense layouts), or compact records (h2g_align_*_fetch_compact: prefixes of
// h2g_alnres of H2G_COMPACT_BYTES(nedits) bytes, back to back) found through a table of pointers built once per read.
struct Recs {
	const h2g_alnres* base = nullptr;
	const h2g_alnres* const* tab = nullptr;
	const h2g_alnres& operator[](size_t k) const { return tab ? *tab[k] : base[k]; }
};
// walks `n` compact records from `p` (at most to `end`) into tab; false = the bytes do not hold n records
inline bool compact_table(const uint8_t* p, const uint8_t* end, size_t n, std::vector<const h2g_alnres*>& tab) {
	tab.resize(n);
	for(size_t k = 0; k < n; k++) {
		if(p + 40 > end) return false;
		const h2g_alnres* r = reinterpret_cast<const h2g_alnres*>(p);
		const size_t b = H2G_COMPACT_BYTES(r->nedits);
		if(p + b > end) return false;
		tab[k] = r;
		p += b;
	}
	return true;
}
// walks every compact record in [p, end) into tab; returns their count (a truncated tail ends the walk)
inline size_t compact_walk(const uint8_t* p, const uint8_t* end, std::vector<const h2g_alnres*>& tab) {
	tab.clear();
	while(p + 40 <= end) {
		const h2g_alnres* r = reinterpret_cast<const h2g_alnres*>(p);
		const size_t b = H2G_COMPACT_BYTES(r->nedits);
		if(p + b > end) break;
		tab.push_back(r);
		p += b;
	}
	return tab.size();
}
void summ_unpaired(Summ& s, int m, const Recs& lst, size_t n) {   // the rs1u_/rs2u_ loop of AlnSetSumm::init
	for(size_t i = 0; i < n; i++) {
		const Score sc = score_of(lst[i]);
		if(sc.gt(s.best[m])) { s.secbest[m] = s.best[m]; s.best[m] = sc; }
		else if(sc.gt(s.secbest[m])) s.secbest[m] = sc;
	}
	s.numAlns[m] = n;
}

}  // namespace

extern "C" h2g_status h2g_sam_open(const char* base, h2g_sam** out) {
	if(!base || !out) return H2G_ERR_ARG;
	HostIndex ix;
	const int rc = load_host_index(base, false, ix, true);       // (light: names, lengths, fragment table, ALTs — no sides / SA sample / reference bases)
	if(rc != 0) return (h2g_status)rc;
	h2g_sam* s = new h2g_sam();
	s->refnames = ix.names;
	s->reflens.assign(ix.g.plen.begin(), ix.g.plen.end());
	s->alts = ix.alts;
	s->altnames = ix.alt_names;
	if(!ix.alts.empty()) {
		h2g::splice_sites_of_alts(reinterpret_cast<const uint32_t*>(ix.alts.data()), ix.alts.size(), sizeof(HostAlt) / 4, ix.g.rstarts.data(), ix.g.nFrag, ix.g.p.len, s->alt_sites);
		s->register_sites(s->alt_sites.data(), s->alt_sites.size());
		if(!s->alt_sites.empty()) h2g::build_splice_db(s->alt_sites.data(), s->alt_sites.size(), (uint32_t)s->refnames.size(), s->ssdb);
	}
	*out = s;
	return H2G_OK;
}
extern "C" void h2g_sam_close(h2g_sam* s) { delete s; }

extern "C" size_t h2g_sam_header(const h2g_sam* S, const char* cmdline, char* out, size_t cap) {
	if(!S) return 0;
	std::string o = "@HD\tVN:1.0\tSO:unsorted\n";
	if(!S->no_sq) for(size_t i = 0; i < S->refnames.size(); i++) {
		o += "@SQ\tSN:"; put_ref_name(o, S->refnames[i]); o += "\tLN:"; put(o, i < S->reflens.size() ? S->reflens[i] : 0); o.push_back('\n');
	}
	if(!S->rg_id.empty()) { o += "@RG"; o += S->rg_id; o += S->rg_fields; o.push_back('\n'); }   // sam.h:456-461
	o += "@PG\tID:hisat2\tPN:hisat2\tVN:2.2.3\tCL:\""; o += cmdline ? cmdline : ""; o += "\"\n";
	if(out && cap) memcpy(out, o.data(), std::min(cap, o.size()));
	return o.size();
}

namespace {
// formats reads [0, n) with `one(i, text)` on S->threads host threads (contiguous ranges, concatenated in read order)
template <class F>
h2g_status drive(const h2g_sam* S, size_t n, F one, char* out, size_t cap, size_t* used) {
	size_t T = S->threads < 1 ? 1 : (size_t)S->threads;
	if(T > n / 2048 + 1) T = n / 2048 + 1;
	// per-thread text and counters, each on cache lines of its own: a std::string's size field changes with every append, and two of them (or two threads' counters) in
	// one line made eight formatter threads as slow as one (measured: 0.176 s for 60 000 pairs on 1 thread, 0.164 s on 8)
	struct alignas(128) Part { std::string o; char pad[128 - sizeof(std::string) % 128]; };
	struct alignas(128) PMet { Met m; };
	std::vector<Part> parts_(T);
	std::vector<PMet> mets_(T);
	struct PartsView { std::vector<Part>& v; std::string& operator[](size_t t) { return v[t].o; } size_t size() const { return v.size(); } } parts{parts_};
	struct MetsView { std::vector<PMet>& v; Met& operator[](size_t t) { return v[t].m; } } mets{mets_};
	auto work = [&](size_t t) {
		const size_t b = n * t / T, e = n * (t + 1) / T;
		std::string& o = parts[t];
		o.reserve((e - b) * 760);                       // (two lines of a 101 bp pair; longer reads grow it)
		tl_novel = &mets[t].novel;
		tl_long_edits = S->long_edits; tl_long_edits_n = S->n_long_edits;
		for(size_t i = b; i < e; i++) { tl_rdid = S->first_read_id + i; one(i, o, mets[t]); }
		tl_novel = nullptr; tl_long_edits = nullptr; tl_long_edits_n = 0;
	};
	if(T == 1) work(0);
	else {
		std::vector<std::thread> th;
		for(size_t t = 1; t < T; t++) th.emplace_back(work, t);
		work(0);
		for(auto& x : th) x.join();
	}
	size_t total = 0;
	for(size_t t = 0; t < T; t++) total += parts[t].size();
	*used = total;
	if(total > cap || !out) return total <= cap && total == 0 ? H2G_OK : H2G_ERR_ARG;
	for(size_t t = 0; t < T; t++) { S->count_sites(mets[t].novel); S->met.add(mets[t]); }   // only a call that delivered its text counts
	// the parts go to the caller's buffer side by side (hundreds of MB per batch: one thread would spend a third of the call here)
	std::vector<size_t> at(T + 1, 0);
	for(size_t t = 0; t < T; t++) at[t + 1] = at[t] + parts[t].size();
	auto place = [&](size_t t) { memcpy(out + at[t], parts[t].data(), parts[t].size()); };
	if(T == 1 || total < (1u << 20)) for(size_t t = 0; t < T; t++) place(t);
	else {
		std::vector<std::thread> th;
		for(size_t t = 1; t < T; t++) th.emplace_back(place, t);
		place(0);
		for(auto& x : th) x.join();
	}
	return H2G_OK;
}
}  // namespace

extern "C" void h2g_sam_set_long_edits(h2g_sam* S, const h2g_edit* area, size_t n) { if(S) { S->long_edits = n ? area : nullptr; S->n_long_edits = area ? n : 0; } }
namespace {
// every long record of a batch must point inside the area the caller set: checked before a line is written (H2G_ERR_ARG otherwise)
bool long_records_ok(const h2g_sam* S, const h2g_alnres* a, size_t n) {
	for(size_t i = 0; i < n; i++) if(a[i].nedits > H2G_MAX_EDITS && (!S->long_edits || (size_t)a[i].edits[0].pos + a[i].nedits > S->n_long_edits)) return false;
	return true;
}
inline bool long_record_ok(const h2g_sam* S, const h2g_alnres& r) {
	return r.nedits <= H2G_MAX_EDITS || (S->long_edits && (size_t)r.edits[0].pos + r.nedits <= S->n_long_edits);
}
// the compact layout of n reads: every read's bytes [boffs[i], boffs[i + 1]) must be whole records (at least `need(i)` of them) and every long record must point inside
// the area — checked before a line is written, so that the formatter itself never meets a malformed buffer (ED() handed out a one-entry list, a failed
// table dropped the read's lines silently)
template <typename NEED>
bool compact_ok(const h2g_sam* S, const uint8_t* rec, const uint64_t* boffs, size_t n, NEED&& need) {
	for(size_t i = 0; i < n; i++) {
		if(boffs[i + 1] < boffs[i]) return false;
		const uint8_t* p = rec + boffs[i]; const uint8_t* const end = rec + boffs[i + 1];
		size_t k = 0;
		while(p < end) {
			if(p + 40 > end) return false;
			const h2g_alnres* r = reinterpret_cast<const h2g_alnres*>(p);
			const size_t b = H2G_COMPACT_BYTES(r->nedits);
			if(p + b > end || !long_record_ok(S, *r)) return false;
			p += b; k++;
		}
		if(k < need(i)) return false;
	}
	return true;
}
}  // namespace
extern "C" void h2g_sam_set_threads(h2g_sam* S, int threads) { if(S) S->threads = threads < 1 ? 1 : threads; }
// AlnSink::printAlSumm aln_sink.h:1637-1815 (old-style summary, -k mode: no repeat threshold, discordant + mixed reporting on)
extern "C" size_t h2g_sam_summary(const h2g_sam* S, char* out, size_t cap) {
	if(!S) return 0;
	const Met& m = S->met;
	std::string o;
	char b[64];
	auto pct = [&](uint64_t num, uint64_t den) { snprintf(b, sizeof b, "%.2f%%", den ? 100.0 * (double)num / (double)den : 0.0); o += b; };
	auto line = [&](const char* ind, uint64_t v, uint64_t den, const char* txt) { o += ind; o += std::to_string(v); o += " ("; pct(v, den); o += ") "; o += txt; o += "\n"; };
	const uint64_t tot_al_cand = m.nunpaired + m.npaired * 2;
	const uint64_t tot_al = (m.nconcord_uni1 + m.nconcord_uni2) * 2 + m.ndiscord * 2 + m.nunp_0_uni1 + m.nunp_0_uni2 + m.nunp_uni1 + m.nunp_uni2;
	if(S->new_summary) {                                  // --new-summary: aln_sink.h:1659-1679
		o += "HISAT2 summary stats:\n";
		auto row = [&](const char* txt, uint64_t v, uint64_t den) { o += txt; o += std::to_string(v); o += " ("; pct(v, den); o += ")\n"; };
		if(m.npaired > 0) {
			const uint64_t n0 = m.nconcord_0 - m.ndiscord;
			o += "\tTotal pairs: "; o += std::to_string(m.npaired); o += "\n";
			row("\t\tAligned concordantly or discordantly 0 time: ", n0, m.npaired);
			row("\t\tAligned concordantly 1 time: ", m.nconcord_uni1, m.npaired);
			row("\t\tAligned concordantly >1 times: ", m.nconcord_uni2, m.npaired);
			row("\t\tAligned discordantly 1 time: ", m.ndiscord, m.npaired);
			o += "\tTotal unpaired reads: "; o += std::to_string(n0 * 2); o += "\n";
			row("\t\tAligned 0 time: ", m.nunp_0_0, n0 * 2);
			row("\t\tAligned 1 time: ", m.nunp_0_uni1, n0 * 2);
			row("\t\tAligned >1 times: ", m.nunp_0_uni2, n0 * 2);
		} else {
			o += "\tTotal reads: "; o += std::to_string(m.nread); o += "\n";
			row("\t\tAligned 0 time: ", m.nunp_0, m.nunpaired);
			row("\t\tAligned 1 time: ", m.nunp_uni1, m.nunpaired);
			row("\t\tAligned >1 times: ", m.nunp_uni2, m.nunpaired);
		}
		o += "\tOverall alignment rate: "; pct(tot_al, tot_al_cand); o += "\n";
		if(out && cap) memcpy(out, o.data(), std::min(cap, o.size()));
		return o.size();
	}
	o += std::to_string(m.nread); o += m.nread ? " reads; of these:\n" : " reads\n";
	if(m.npaired > 0) {
		line("  ", m.npaired, m.nread, "were paired; of these:");
		line("    ", m.nconcord_0, m.npaired, "aligned concordantly 0 times");
		line("    ", m.nconcord_uni1, m.npaired, "aligned concordantly exactly 1 time");
		line("    ", m.nconcord_uni2, m.npaired, "aligned concordantly >1 times");
		if(S->report_discordant) {                          // aln_sink.h:1724-1734
			o += "    ----\n    "; o += std::to_string(m.nconcord_0); o += " pairs aligned concordantly 0 times; of these:\n";
			line("      ", m.ndiscord, m.nconcord_0, "aligned discordantly 1 time");
		}
		const uint64_t n0 = m.nconcord_0 - m.ndiscord;
		if(S->report_mixed) {                               // :1736-1772
			o += "    ----\n    "; o += std::to_string(n0); o += " pairs aligned 0 times concordantly or discordantly; of these:\n";
			o += "      "; o += std::to_string(n0 * 2); o += " mates make up the pairs; of these:\n";
			line("        ", m.nunp_0_0, n0 * 2, "aligned 0 times");
			line("        ", m.nunp_0_uni1, n0 * 2, "aligned exactly 1 time");
			line("        ", m.nunp_0_uni2, n0 * 2, "aligned >1 times");
		}
	}
	if(m.nunpaired > 0) {
		line("  ", m.nunpaired, m.nread, "were unpaired; of these:");
		line("    ", m.nunp_0, m.nunpaired, "aligned 0 times");
		line("    ", m.nunp_uni1, m.nunpaired, "aligned exactly 1 time");
		line("    ", m.nunp_uni2, m.nunpaired, "aligned >1 times");
	}
	pct(tot_al, tot_al_cand); o += " overall alignment rate\n";
	if(out && cap) memcpy(out, o.data(), std::min(cap, o.size()));
	return o.size();
}
// --rg-id <text> (id != NULL) and --rg <text> (field != NULL; "ID:x" sets the id like --rg-id x), hisat2.cpp:1389-1407
extern "C" void h2g_sam_add_read_group(h2g_sam* S, const char* id, const char* field) {
	if(!S) return;
	if(id) { S->rg_id = std::string("\tID:") + id; S->rg_optflag = std::string("RG:Z:") + id; }
	if(field) {
		const std::string f = field;
		if(f.compare(0, 3, "ID:") == 0) { S->rg_id = "\t" + f; S->rg_optflag = "RG:Z:" + f.substr(3); }
		else { S->rg_fields += '\t'; S->rg_fields += f; }
	}
}
// --remove-chrname (mode 1) / --add-chrname (mode 2): hisat2.cpp:3962-3976, before anything else uses the names (@SQ, RNAME, the
// splice-site files' names, --novel-splicesite-outfile)
extern "C" void h2g_sam_set_chrname_mode(h2g_sam* S, int mode) {
	if(!S) return;
	for(std::string& n : S->refnames) {
		if(mode == 1) { if(n.compare(0, 3, "chr") == 0) n = n.substr(3); }
		else if(mode == 2) { if(n.compare(0, 3, "chr") != 0) n = "chr" + n; }
	}
}
extern "C" void h2g_sam_set_new_summary(h2g_sam* S, int on) { if(S) S->new_summary = on != 0; }
extern "C" void h2g_sam_set_header_options(h2g_sam* S, int no_sq, int omit_sec_seq) { if(S) { S->no_sq = no_sq != 0; S->omit_sec_seq = omit_sec_seq != 0; } }
extern "C" void h2g_sam_set_report_policy(h2g_sam* S, int discordant, int mixed) { if(S) { S->report_discordant = discordant != 0; S->report_mixed = mixed != 0; } }
extern "C" void h2g_sam_set_no_unal(h2g_sam* S, int on) { if(S) S->no_unal = on != 0; }
extern "C" void h2g_sam_set_secondary(h2g_sam* S, int on) { if(S) S->secondary = on != 0; }
// SpliceSiteDB::read(ifstream&, known) splice_site.cpp:727-776: whitespace-separated (name, left, right, strand) records; names the
// index does not hold are skipped.  Returns the number of records; the first `cap` are written to out.
extern "C" size_t h2g_sam_read_splice_site_file(const h2g_sam* S, const char* path, int known, h2g_splice_site* out, size_t cap) {
	if(!S || !path) return 0;
	FILE* f = fopen(path, "rb");
	if(!f) return (size_t)-1;
	std::string data;
	char buf[1 << 16];
	size_t got;
	while((got = fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, got);
	fclose(f);
	std::vector<std::string> first(S->refnames.size());
	for(size_t i = 0; i < S->refnames.size(); i++) { const std::string& nm = S->refnames[i]; size_t k = 0; while(k < nm.size() && !isspace((unsigned char)nm[k])) k++; first[i] = nm.substr(0, k); }
	size_t n = 0, pos = 0;
	auto token = [&](std::string& t) { while(pos < data.size() && isspace((unsigned char)data[pos])) pos++; const size_t a = pos; while(pos < data.size() && !isspace((unsigned char)data[pos])) pos++; t.assign(data, a, pos - a); return pos > a; };
	std::string name, l, r, d;
	while(token(name) && token(l) && token(r) && token(d)) {
		uint32_t ref = 0;
		for(; ref < first.size(); ref++) if(first[ref] == name) break;
		if(ref >= first.size()) continue;
		if(n < cap && out) {
			h2g_splice_site& x = out[n];
			x.tidx = ref; x.left = (uint32_t)strtoul(l.c_str(), nullptr, 10); x.right = (uint32_t)strtoul(r.c_str(), nullptr, 10); x.readid = 0;
			x.dir = d[0] == '+' ? 2 : 3; x.fromfile = 1; x.known = known ? 1 : 0; x.editdist = 0;    // SPL_FW : SPL_RC
		}
		n++;
	}
	return n;
}
extern "C" void h2g_sam_set_splice_sites(h2g_sam* S, const h2g_splice_site* sites, size_t n, uint32_t window) {
	if(!S) return;
	std::vector<h2g_splice_site> all(S->alt_sites);      // the index's own sites first: of equal sites the first is kept
	if(n) all.insert(all.end(), sites, sites + n);
	h2g::build_splice_db(all.data(), all.size(), (uint32_t)S->refnames.size(), S->ssdb);
	S->ssdb_window = window;
	S->register_sites(sites, n);
}
extern "C" void h2g_sam_add_splice_sites(h2g_sam* S, const h2g_splice_site* delta, size_t n) {
	if(!S || !n) return;
	h2g::merge_splice_db(S->ssdb, delta, n, (uint32_t)S->refnames.size());
	S->register_sites(delta, n);
}
extern "C" void h2g_sam_set_rna_strandness(h2g_sam* S, int code) { if(S) S->rna_strandness = code; }
extern "C" void h2g_sam_set_templatelen_adjustment(h2g_sam* S, int on) { if(S) S->tlen_adjust = on != 0; }
extern "C" void h2g_sam_collect_novel_sites(h2g_sam* S, int on) { if(S) S->collect_novel = on != 0; }
extern "C" void h2g_sam_set_first_read_id(h2g_sam* S, uint64_t id) { if(S) S->first_read_id = id; }
extern "C" size_t h2g_sam_take_novel_sites(h2g_sam* S, h2g_splice_site* out, size_t cap) {
	if(!S) return 0;
	const size_t n = S->met.novel.size();
	if(!out || cap < n) return n;                          // nothing is consumed until the caller's buffer holds them all
	memcpy(out, S->met.novel.data(), n * sizeof(h2g_splice_site));
	S->met.novel.clear();
	return n;
}
// SpliceSiteDB::print splice_site.cpp:565-653 (--novel-splicesite-outfile, hisat2.cpp:4189-4197): the sites in SpliceSitePos order;
// one is listed when enough lines crossed it — the 70 % point of the sites' read-count distribution — or, with no edits, 1e-5 of the
// number of sites; of sites whose left ends are < 10 apart and whose shifts agree within 10 the one with more reads survives.
extern "C" size_t h2g_sam_novel_splice_sites_text(const h2g_sam* S, char* out, size_t cap) {
	if(!S) return 0;
	struct Row { uint32_t ref, left, right, dir; uint64_t numreads; };
	int64_t dist[100] = {0};
	for(const auto& kv : S->site_stats) dist[kv.second.numreads < 100 ? kv.second.numreads : 99]++;
	for(int i = 1; i < 100; i++) dist[i] += dist[i - 1];
	uint32_t cutoff = 0;
	for(int i = 0; i < 100; i++) { const float cmf = float(dist[i]) / dist[99]; if(cmf > 0.7) { cutoff = (uint32_t)i; break; } }
	const uint32_t cutoff2 = (uint32_t)(S->site_stats.size() / 100000);
	std::string o;
	std::vector<Row> list;                               // sites waiting for a later, better neighbour
	auto flush = [&](const Row* ss) {                    // print_impl
		size_t i = 0;
		while(i < list.size()) {
			const Row t = list[i];
			bool do_print = true;
			if(ss && t.ref == ss->ref && ss->left < t.left + 10) {
				do_print = false;
				if(std::abs(((int)ss->left - (int)t.left) - ((int)ss->right - (int)t.right)) <= 10) {
					if(t.numreads < ss->numreads) { list.erase(list.begin() + (long)i); list.push_back(*ss); }
					return;
				}
			}
			if(!do_print) { i++; continue; }
			put_ref_name(o, S->refnames[t.ref]); o.push_back('\t');   // first token of the header, as SpliceSiteDB keeps it (splice_site.cpp:135-144)
			put(o, (int64_t)t.left); o.push_back('\t'); put(o, (int64_t)t.right); o.push_back('\t');
			o.push_back(t.dir == 2 || t.dir == 4 ? '+' : t.dir == 3 || t.dir == 5 ? '-' : '.');
			o.push_back('\n');
			list.erase(list.begin() + (long)i);
		}
		if(ss) list.push_back(*ss);
	};
	for(const auto& kv : S->site_stats) {
		const Row r = {kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second.numreads};
		if(r.numreads >= cutoff || (kv.second.editdist == 0 && r.numreads >= cutoff2)) flush(&r);
	}
	flush(nullptr);
	if(out && cap) memcpy(out, o.data(), std::min(cap, o.size()));
	return o.size();
}
extern "C" void h2g_sam_set_score_min(h2g_sam* S, uint32_t type, double c, double coeff) { if(S) { S->smType = type; S->smConst = c; S->smCoeff = coeff; } }

static h2g_status format_unpaired(const h2g_sam* S, const uint8_t* codes, const uint32_t* offs, const char* quals,
                                  const char* nb, const uint32_t* noffs, size_t n, const h2g_read_result* res,
                                  const h2g_alnres* aln, const uint64_t* aln_offs, char* out, size_t cap, size_t* used, const uint8_t* compact = nullptr)
{
	if(compact) aln = reinterpret_cast<const h2g_alnres*>(compact);       // (aln_offs are byte offsets then)
	if(!S || !codes || !offs || !nb || !noffs || !res || !aln || !used) return H2G_ERR_ARG;
	auto one = [&](size_t i, std::string& o, Met& met) {
		Rd rd = {nb + noffs[i], noffs[i + 1] - noffs[i], codes + offs[i], offs[i + 1] - offs[i], quals ? quals + offs[i] : nullptr};
		Flags fl;
		read_filters(rd, &fl.lenfilt, &fl.nfilt);
		Summ summ;
		const h2g_read_result& r = res[i];
		if(r.best != INT32_MIN) { summ.best[0].valid = true; summ.best[0].score = r.best; summ.best[0].h2 = (int64_t)(((uint64_t)(int64_t)r.best << 32) | r.best_h2); }
		if(r.secbest != INT32_MIN) { summ.secbest[0].valid = true; summ.secbest[0].score = r.secbest; summ.secbest[0].h2 = (int64_t)(((uint64_t)(int64_t)r.secbest << 32) | r.secbest_h2); }
		const uint32_t nsel = r.nselect < H2G_ALN_CAP ? r.nselect : H2G_ALN_CAP;
		met.nread++; met.nunpaired++;
		if(nsel == 0) met.nunp_0++; else if(nsel == 1) met.nunp_uni1++; else met.nunp_uni2++;
		if(nsel == 0) append_mate(*S, o, rd, nullptr, nullptr, nullptr, summ, fl, 0);
		static thread_local std::vector<const h2g_alnres*> tab;
		Recs R;
		if(compact) { if(!compact_table(compact + aln_offs[i], compact + aln_offs[i + 1], nsel, tab)) return; R.tab = tab.data(); }
		else R.base = aln_offs ? aln + aln_offs[i] : aln + i * H2G_ALN_CAP;
		for(uint32_t k = 0; k < nsel; k++) {
			fl.primary = k == 0;
			append_mate(*S, o, rd, nullptr, &R[k], nullptr, summ, fl, nsel);
		}
	};
	return drive(S, n, one, out, cap, used);
}

extern "C" h2g_status h2g_sam_format_unpaired(const h2g_sam* S, const uint8_t* codes, const uint32_t* offs, const char* quals,
                                              const char* nb, const uint32_t* noffs, size_t n, const h2g_read_result* res,
                                              const h2g_alnres* aln, char* out, size_t cap, size_t* used)
{
	if(!S || !res || !aln) return H2G_ERR_ARG;
	for(size_t i = 0; i < n; i++) for(uint32_t k = 0; k < res[i].nselect && k < H2G_ALN_CAP; k++) if(!long_record_ok(S, aln[i * H2G_ALN_CAP + k])) return H2G_ERR_ARG;
	return format_unpaired(S, codes, offs, quals, nb, noffs, n, res, aln, nullptr, out, cap, used);
}
extern "C" h2g_status h2g_sam_format_unpaired_dense(const h2g_sam* S, const uint8_t* codes, const uint32_t* offs, const char* quals,
                                                    const char* nb, const uint32_t* noffs, size_t n, const h2g_read_result* res,
                                                    const h2g_alnres* aln, const uint64_t* aln_offs, char* out, size_t cap, size_t* used)
{
	if(!aln_offs || !S || !aln) return H2G_ERR_ARG;
	if(!long_records_ok(S, aln, aln_offs[n])) return H2G_ERR_ARG;     // a long record without its area (h2g_sam_set_long_edits)
	return format_unpaired(S, codes, offs, quals, nb, noffs, n, res, aln, aln_offs, out, cap, used);
}

static h2g_status format_paired(const h2g_sam* S, const uint8_t* codes1, const uint32_t* offs1, const char* quals1,
                                const char* nb1, const uint32_t* noffs1, const uint8_t* codes2, const uint32_t* offs2,
                                const char* quals2, const char* nb2, const uint32_t* noffs2, size_t n,
                                const h2g_pair_result* res, const h2g_alnres* aln1, const uint64_t* ao1, const h2g_alnres* aln2, const uint64_t* ao2,
                                uint32_t khits, char* out, size_t cap, size_t* used, const uint8_t* compact1 = nullptr, const uint8_t* compact2 = nullptr)
{
	if(compact1) { aln1 = reinterpret_cast<const h2g_alnres*>(compact1); aln2 = reinterpret_cast<const h2g_alnres*>(compact2); }   // (ao1 / ao2 are byte offsets then)
	if(!S || !codes1 || !offs1 || !nb1 || !noffs1 || !codes2 || !offs2 || !nb2 || !noffs2 || !res || !aln1 || !aln2 || !used) return H2G_ERR_ARG;
	auto one = [&](size_t i, std::string& o, Met& met) {
		static thread_local std::vector<size_t> sel, sel1, sel2;         // (reused: the formatter allocates nothing per pair)
		static thread_local std::vector<Score> keys;
		sel.clear(); sel1.clear(); sel2.clear(); keys.clear();
		met.nread++; met.npaired++;
		const h2g_pair_result& pr = res[i];
		Rd rd[2] = {{nb1 + noffs1[i], noffs1[i + 1] - noffs1[i], codes1 + offs1[i], offs1[i + 1] - offs1[i], quals1 ? quals1 + offs1[i] : nullptr},
		            {nb2 + noffs2[i], noffs2[i + 1] - noffs2[i], codes2 + offs2[i], offs2[i + 1] - offs2[i], quals2 ? quals2 + offs2[i] : nullptr}};
		Recs r1, r2;
		size_t n1, n2;
		if(compact1) {   // every record the device kept, found by walking the read's bytes
			static thread_local std::vector<const h2g_alnres*> t1, t2;
			n1 = compact_walk(compact1 + ao1[i], compact1 + ao1[i + 1], t1); n2 = compact_walk(compact2 + ao2[i], compact2 + ao2[i + 1], t2);
			r1.tab = t1.data(); r2.tab = t2.data();
		} else {
			r1.base = ao1 ? aln1 + ao1[i] : aln1 + i * H2G_PAIR_RES_CAP;
			r2.base = ao2 ? aln2 + ao2[i] : aln2 + i * H2G_PAIR_RES_CAP;
			// records available: the dense layout carries every record the device kept, the slot layout H2G_PAIR_RES_CAP per mate
			n1 = ao1 ? (size_t)(ao1[i + 1] - ao1[i]) : std::min<size_t>(pr.nres[0], H2G_PAIR_RES_CAP);
			n2 = ao2 ? (size_t)(ao2[i + 1] - ao2[i]) : std::min<size_t>(pr.nres[1], H2G_PAIR_RES_CAP);
		}
		Flags f1, f2;
		read_filters(rd[0], &f1.lenfilt, &f1.nfilt);
		read_filters(rd[1], &f2.lenfilt, &f2.nfilt);
		Rng rnd = {pr.rnd_state};
		// ReportingState::foundConcordant aln_sink.cpp:74-112: concordant pairs are kept while they tie the best pair score?
		// No — every concordant pair reported is kept (rs1_/rs2_); nconcord = their count
		// pairs naming a record beyond the H2G_PAIR_RES_CAP returned per mate (such a pair carries an overflow flag) are dropped
		// rather than indexed past the end of the fetched records
		uint8_t pi[H2G_PAIR_CAP], pj[H2G_PAIR_CAP];
		size_t np = 0;
		for(size_t k = 0; k < std::min<size_t>(pr.npairs, H2G_PAIR_CAP); k++)
			if(pr.pair_i[k] < n1 && pr.pair_j[k] < n2) { pi[np] = pr.pair_i[k]; pj[np] = pr.pair_j[k]; np++; }
		if(np > 0) {
			Summ summ;
			summ.paired = true;
			summ_unpaired(summ, 0, r1, n1); summ_unpaired(summ, 1, r2, n2);
			keys.clear();
			// ReportingState::foundConcordant (aln_sink.cpp:74-112): the count of concordant pairs to report restarts whenever a
			// pair with a strictly better score sum arrives
			size_t nconc = 0;
			int64_t bestsum = INT64_MIN;
			for(size_t k = 0; k < np; k++) {
				const Score sc = add(score_of(r1[pi[k]]), score_of(r2[pj[k]]));
				if(k == 0 || sc.score > bestsum) { bestsum = sc.score; nconc = 0; }
				nconc++;
				keys.push_back(sc);
				if(sc.gt(summ.bestPaired)) { summ.secbestPaired = summ.bestPaired; summ.bestPaired = sc; }
				else if(sc.gt(summ.secbestPaired)) summ.secbestPaired = sc;
			}
			select_by_score(keys, std::min<size_t>(khits, nconc), rnd, sel, S->secondary);
			if(sel.size() == 1) met.nconcord_uni1++; else met.nconcord_uni2++;
			for(size_t q = 0; q < sel.size(); q++) {
				const h2g_alnres* a = &r1[pi[sel[q]]];
				const h2g_alnres* b = &r2[pj[sel[q]]];
				f1.pairing = PAIR_CONCORD_M1; f2.pairing = PAIR_CONCORD_M2;
				f1.primary = f2.primary = q == 0;
				f1.oppAligned = f2.oppAligned = true;
				append_mate(*S, o, rd[0], &rd[1], a, b, summ, f1, sel.size());
				append_mate(*S, o, rd[1], &rd[0], b, a, summ, f2, sel.size());
			}
		} else if(S->report_discordant && n1 == 1 && n2 == 1) {
			// discordant: one unpaired alignment per mate (ReportingState::finish -> convertUnpairedToDiscordant)
			Summ summ;
			summ.paired = true;
			summ_unpaired(summ, 0, r1, n1); summ_unpaired(summ, 1, r2, n2);
			keys.assign(1, add(score_of(r1[0]), score_of(r2[0])));
			summ.bestPaired = keys[0];
			select_by_score(keys, 1, rnd, sel, S->secondary);
			met.nconcord_0++; met.ndiscord++;
			f1.pairing = PAIR_DISCORD_M1; f2.pairing = PAIR_DISCORD_M2;
			f1.oppAligned = f2.oppAligned = true;
			append_mate(*S, o, rd[0], &rd[1], &r1[0], &r2[0], summ, f1, 1);
			append_mate(*S, o, rd[1], &rd[0], &r2[0], &r1[0], summ, f2, 1);
		} else {
			Summ s1, s2;
			sel1.clear(); sel2.clear();
			const size_t u1 = S->report_mixed ? n1 : 0, u2 = S->report_mixed ? n2 : 0;                      // --no-mixed: ReportingState::getReport leaves the unpaired counts at 0 (aln_sink.cpp:280)
			if(u1) { keys.clear(); for(size_t k = 0; k < u1; k++) keys.push_back(score_of(r1[k])); select_by_score(keys, std::min<size_t>(khits, u1), rnd, sel1, S->secondary); }
			if(u2) { keys.clear(); for(size_t k = 0; k < u2; k++) keys.push_back(score_of(r2[k])); select_by_score(keys, std::min<size_t>(khits, u2), rnd, sel2, S->secondary); }
			summ_unpaired(s1, 0, r1, u1); summ_unpaired(s1, 1, r2, u2);
			s2 = s1;
			met.nconcord_0++;
			if(sel1.empty()) met.nunp_0_0++; else if(sel1.size() == 1) met.nunp_0_uni1++; else met.nunp_0_uni2++;
			if(sel2.empty()) met.nunp_0_0++; else if(sel2.size() == 1) met.nunp_0_uni1++; else met.nunp_0_uni2++;
			const h2g_alnres* p1 = sel1.empty() ? nullptr : &r1[sel1[0]];
			const h2g_alnres* p2 = sel2.empty() ? nullptr : &r2[sel2[0]];
			f1.pairing = PAIR_UNP_M1; f2.pairing = PAIR_UNP_M2;
			f1.oppAligned = p2 != nullptr; f2.oppAligned = p1 != nullptr;
			auto unal = [&](int m, const h2g_alnres* opp) {          // g_.reportUnaligned for the mate that did not align
				Summ se = m == 0 ? s1 : s2;
				if(opp) { se.orefid = opp->tidx; se.orefoff = opp->toff; }
				Flags& f = m == 0 ? f1 : f2;
				f.primary = true;
				append_mate(*S, o, rd[m], nullptr, nullptr, nullptr, se, f, 0);
			};
			// print order of finishRead's unpaired branch (aln_sink.h:2380-2560; reportHits interleaves the two primaries)
			if(p1 && p2) {
				f1.primary = f2.primary = true;
				append_mate(*S, o, rd[0], &rd[1], p1, p2, s1, f1, sel1.size());
				append_mate(*S, o, rd[1], &rd[0], p2, p1, s2, f2, sel2.size());
				f1.primary = f2.primary = false;
				for(size_t q = 1; q < sel1.size(); q++) append_mate(*S, o, rd[0], &rd[1], &r1[sel1[q]], p2, s1, f1, sel1.size());
				for(size_t q = 1; q < sel2.size(); q++) append_mate(*S, o, rd[1], &rd[0], &r2[sel2[q]], p1, s2, f2, sel2.size());
			} else if(p1) {
				for(size_t q = 0; q < sel1.size(); q++) { f1.primary = q == 0; append_mate(*S, o, rd[0], nullptr, &r1[sel1[q]], nullptr, s1, f1, sel1.size()); }
				unal(1, p1);
			} else if(p2) {
				for(size_t q = 0; q < sel2.size(); q++) { f2.primary = q == 0; append_mate(*S, o, rd[1], nullptr, &r2[sel2[q]], nullptr, s2, f2, sel2.size()); }
				unal(0, p2);
			} else { unal(0, nullptr); unal(1, nullptr); }
		}
	};
	return drive(S, n, one, out, cap, used);
}

extern "C" h2g_status h2g_sam_format_paired_compact(const h2g_sam* S, const uint8_t* codes1, const uint32_t* offs1, const char* quals1,
                                                    const char* nb1, const uint32_t* noffs1, const uint8_t* codes2, const uint32_t* offs2,
                                                    const char* quals2, const char* nb2, const uint32_t* noffs2, size_t n,
                                                    const h2g_pair_result* res, const uint8_t* rec1, const uint64_t* boffs1,
                                                    const uint8_t* rec2, const uint64_t* boffs2, uint32_t khits, char* out, size_t cap, size_t* used)
{
	if(!boffs1 || !boffs2 || !S || !rec1 || !rec2 || !res) return H2G_ERR_ARG;
	if(!compact_ok(S, rec1, boffs1, n, [](size_t) { return (size_t)0; }) || !compact_ok(S, rec2, boffs2, n, [](size_t) { return (size_t)0; })) return H2G_ERR_ARG;
	return format_paired(S, codes1, offs1, quals1, nb1, noffs1, codes2, offs2, quals2, nb2, noffs2, n, res, nullptr, boffs1, nullptr, boffs2, khits, out, cap, used, rec1, rec2);
}
extern "C" h2g_status h2g_sam_format_unpaired_compact(const h2g_sam* S, const uint8_t* codes, const uint32_t* offs, const char* quals,
                                                      const char* nb, const uint32_t* noffs, size_t n, const h2g_read_result* res,
                                                      const uint8_t* rec, const uint64_t* boffs, char* out, size_t cap, size_t* used)
{
	if(!boffs || !S || !rec || !res) return H2G_ERR_ARG;
	if(!compact_ok(S, rec, boffs, n, [&](size_t i) { return (size_t)(res[i].nselect < H2G_ALN_CAP ? res[i].nselect : H2G_ALN_CAP); })) return H2G_ERR_ARG;
	return format_unpaired(S, codes, offs, quals, nb, noffs, n, res, nullptr, boffs, out, cap, used, rec);
}
extern "C" h2g_status h2g_sam_format_paired(const h2g_sam* S, const uint8_t* codes1, const uint32_t* offs1, const char* quals1,
                                            const char* nb1, const uint32_t* noffs1, const uint8_t* codes2, const uint32_t* offs2,
                                            const char* quals2, const char* nb2, const uint32_t* noffs2, size_t n,
                                            const h2g_pair_result* res, const h2g_alnres* aln1, const h2g_alnres* aln2,
                                            uint32_t khits, char* out, size_t cap, size_t* used)
{
	if(!S || !res || !aln1 || !aln2) return H2G_ERR_ARG;
	for(size_t i = 0; i < n; i++) for(int m = 0; m < 2; m++) for(uint32_t k = 0; k < res[i].nres[m] && k < H2G_PAIR_RES_CAP; k++)
		if(!long_record_ok(S, (m ? aln2 : aln1)[i * H2G_PAIR_RES_CAP + k])) return H2G_ERR_ARG;
	return format_paired(S, codes1, offs1, quals1, nb1, noffs1, codes2, offs2, quals2, nb2, noffs2, n, res, aln1, nullptr, aln2, nullptr, khits, out, cap, used);
}
extern "C" h2g_status h2g_sam_format_paired_dense(const h2g_sam* S, const uint8_t* codes1, const uint32_t* offs1, const char* quals1,
                                                  const char* nb1, const uint32_t* noffs1, const uint8_t* codes2, const uint32_t* offs2,
                                                  const char* quals2, const char* nb2, const uint32_t* noffs2, size_t n,
                                                  const h2g_pair_result* res, const h2g_alnres* aln1, const uint64_t* aln_offs1,
                                                  const h2g_alnres* aln2, const uint64_t* aln_offs2, uint32_t khits, char* out, size_t cap, size_t* used)
{
	if(!aln_offs1 || !aln_offs2 || !S || !aln1 || !aln2) return H2G_ERR_ARG;
	if(!long_records_ok(S, aln1, aln_offs1[n]) || !long_records_ok(S, aln2, aln_offs2[n])) return H2G_ERR_ARG;   // a long record without its area (h2g_sam_set_long_edits)
	return format_paired(S, codes1, offs1, quals1, nb1, noffs1, codes2, offs2, quals2, nb2, noffs2, n, res, aln1, aln_offs1, aln2, aln_offs2, khits, out, cap, used);
}
