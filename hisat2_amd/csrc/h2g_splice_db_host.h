// h2g_splice_db_host.h — host side of the splice-site database: the two sorted arrays of DSpliceDB (h2g_core.h) from a list of
// sites, as SpliceSiteDB::read(ifstream&, known) (splice_site.cpp:727-776) builds its two red-black trees: a site equal to an
// earlier one in (text, left, right, dir) is dropped.
#pragma once
#include <algorithm>
#include <vector>
#include "h2g_core.h"

namespace h2g {
struct HostSpliceDB {
	std::vector<DSpliceSite> fw, bw;
	std::vector<uint32_t> fw_first, bw_first;
};
inline void build_splice_db(const h2g_splice_site* sites, size_t n, uint32_t nPat, HostSpliceDB& db) {
	struct E { uint32_t tidx; DSpliceSite s; size_t order; };
	std::vector<E> v;
	v.reserve(n);
	for(size_t i = 0; i < n; i++) {
		if(sites[i].tidx >= nPat) continue;
		E e; e.tidx = sites[i].tidx; e.order = i;
		e.s.left = sites[i].left; e.s.right = sites[i].right; e.s.readid = sites[i].readid; e.s.dir = sites[i].dir;
		e.s.fromfile = sites[i].fromfile; e.s.known = sites[i].known; e.s.pad = 0;
		v.push_back(e);
	}
	auto key_fw = [](const E& a, const E& b) {
		if(a.tidx != b.tidx) return a.tidx < b.tidx;
		if(a.s.left != b.s.left) return a.s.left < b.s.left;
		if(a.s.right != b.s.right) return a.s.right < b.s.right;
		if(a.s.dir != b.s.dir) return a.s.dir < b.s.dir;
		return a.order < b.order;                                  // the first of equal sites is the one kept
	};
	std::sort(v.begin(), v.end(), key_fw);
	std::vector<E> u;
	for(const E& e : v)
		if(u.empty() || u.back().tidx != e.tidx || u.back().s.left != e.s.left || u.back().s.right != e.s.right || u.back().s.dir != e.s.dir) u.push_back(e);
	db.fw.clear(); db.bw.clear();
	db.fw_first.assign(nPat + 1, 0); db.bw_first.assign(nPat + 1, 0);
	for(const E& e : u) { db.fw.push_back(e.s); db.fw_first[e.tidx + 1]++; }
	for(uint32_t t = 0; t < nPat; t++) db.fw_first[t + 1] += db.fw_first[t];
	std::sort(u.begin(), u.end(), [](const E& a, const E& b) {
		if(a.tidx != b.tidx) return a.tidx < b.tidx;
		if(a.s.right != b.s.right) return a.s.right < b.s.right;
		if(a.s.left != b.s.left) return a.s.left < b.s.left;
		return a.s.dir < b.s.dir;
	});
	for(const E& e : u) db.bw.push_back(e.s);
	db.bw_first = db.fw_first;
}
// The database plus `delta` (sites met since the last call, or known ones whose smallest read id went down) in O(sites + delta log delta):
// what build_splice_db(all the sites in order of first appearance) would give, without sorting everything again per wave of reads.
// A delta site equal to an existing one in (text, left, right, dir) only lowers that one's read id (SpliceSiteDB::addSpliceSite keeps
// the smallest id per site, splice_site.cpp:243-276); a site read from a file or the index keeps its id.
inline void merge_splice_db(HostSpliceDB& db, const h2g_splice_site* delta, size_t d, uint32_t nPat) {
	if(db.fw_first.size() != (size_t)nPat + 1) { db.fw.clear(); db.bw.clear(); db.fw_first.assign(nPat + 1, 0); db.bw_first.assign(nPat + 1, 0); }
	struct E { uint32_t tidx; DSpliceSite s; size_t order; };
	std::vector<E> v;
	v.reserve(d);
	for(size_t i = 0; i < d; i++) {
		if(delta[i].tidx >= nPat) continue;
		E e; e.tidx = delta[i].tidx; e.order = i;
		e.s.left = delta[i].left; e.s.right = delta[i].right; e.s.readid = delta[i].readid; e.s.dir = delta[i].dir;
		e.s.fromfile = delta[i].fromfile; e.s.known = delta[i].known; e.s.pad = 0;
		v.push_back(e);
	}
	if(v.empty()) return;
	for(int side = 0; side < 2; side++) {
		// fw: (text, left, right, dir); bw: (text, right, left, dir)
		auto k1 = [side](const DSpliceSite& x) { return side == 0 ? x.left : x.right; };
		auto k2 = [side](const DSpliceSite& x) { return side == 0 ? x.right : x.left; };
		auto less = [&](uint32_t ta, const DSpliceSite& a, uint32_t tb, const DSpliceSite& b) {
			if(ta != tb) return ta < tb;
			if(k1(a) != k1(b)) return k1(a) < k1(b);
			if(k2(a) != k2(b)) return k2(a) < k2(b);
			return a.dir < b.dir;
		};
		std::sort(v.begin(), v.end(), [&](const E& a, const E& b) {
			if(less(a.tidx, a.s, b.tidx, b.s)) return true;
			if(less(b.tidx, b.s, a.tidx, a.s)) return false;
			return a.order < b.order;
		});
		std::vector<DSpliceSite>& cur = side == 0 ? db.fw : db.bw;
		const std::vector<uint32_t>& first = side == 0 ? db.fw_first : db.bw_first;
		std::vector<DSpliceSite> out;
		std::vector<uint32_t> nfirst(nPat + 1, 0);
		out.reserve(cur.size() + v.size());
		size_t j = 0;
		for(uint32_t t = 0; t < nPat; t++) {
			size_t i = first[t];
			const size_t ie = first[t + 1];
			const size_t at0 = out.size();
			while(i < ie || (j < v.size() && v[j].tidx == t)) {
				const bool have_d = j < v.size() && v[j].tidx == t;
				if(i < ie && (!have_d || less(t, cur[i], t, v[j].s))) { out.push_back(cur[i++]); continue; }
				if(i < ie && !less(t, v[j].s, t, cur[i])) {                 // equal keys: the known site stays, its read id may go down
					DSpliceSite x = cur[i++];
					while(j < v.size() && v[j].tidx == t && !less(t, x, t, v[j].s)) { if(!x.fromfile && v[j].s.readid < x.readid) x.readid = v[j].s.readid; j++; }
					out.push_back(x);
					continue;
				}
				DSpliceSite x = v[j++].s;                                  // a new site (of equal new ones the first, with the smallest id)
				while(j < v.size() && v[j].tidx == t && !less(t, x, t, v[j].s)) { if(!x.fromfile && v[j].s.readid < x.readid) x.readid = v[j].s.readid; j++; }
				out.push_back(x);
			}
			nfirst[t + 1] = (uint32_t)(out.size() - at0);
		}
		for(uint32_t t = 0; t < nPat; t++) nfirst[t + 1] += nfirst[t];
		cur.swap(out);
		(side == 0 ? db.fw_first : db.bw_first) = nfirst;
	}
}
// SpliceSiteDB::read(gfm, alts) splice_site.cpp:653-725: the splice-site ALTs of a --ss index enter the database as known sites
// read from a file (the forward copies only: left < right); joined coordinates become (text, offset), exon flanks = left - 1 / right + 1.
// `rstarts` = GFM::rstarts() (nFrag triples), `len` = the joined length.
inline void splice_sites_of_alts(const uint32_t* alts /* {pos, type, len, pad, seq lo, seq hi} x n */, size_t n, size_t stride_words,
                                 const uint32_t* rstarts, uint32_t nFrag, uint32_t joined_len, std::vector<h2g_splice_site>& out) {
	for(size_t i = 0; i < n; i++) {
		const uint32_t* a = alts + i * stride_words;
		const uint32_t left_j = a[0], type = a[1], right_j = a[2];
		if(type != 5) continue;                                    // ALT_SPLICESITE (exons only matter under --avoid-pseudogene)
		if(left_j > right_j) continue;
		// joinedToTextOff(1, left, ..., rejectStraddle = true)
		uint32_t top = 0, bot = nFrag, elt = 0xffffffffu, tidx = 0xffffffffu, toff = 0;
		while(true) {
			const uint32_t oldelt = elt;
			elt = top + ((bot - top) >> 1);
			if(oldelt == elt) break;
			const uint32_t lower = rstarts[elt * 3], upper = (elt == nFrag - 1) ? joined_len : rstarts[(elt + 1) * 3];
			if(lower <= left_j) {
				if(upper > left_j) { if(left_j + 1 <= upper) { tidx = rstarts[elt * 3 + 1]; toff = (left_j - lower) + rstarts[elt * 3 + 2]; } break; }
				top = elt;
			} else bot = elt;
		}
		if(tidx == 0xffffffffu) continue;
		h2g_splice_site x;
		x.tidx = tidx; x.left = toff - 1; x.right = toff + (right_j - left_j) + 1; x.readid = 0;
		x.dir = (a[4] & 0xff) ? 2 : 3; x.fromfile = 1; x.known = 1; x.editdist = 0;   // SPL_FW : SPL_RC
		out.push_back(x);
	}
}
}  // namespace h2g
