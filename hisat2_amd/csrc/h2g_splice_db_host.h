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
