// h2g_host_index.h — host-side parse of the on-disk .ht2 index (format unchanged from the reference).
//
// Layout facts follow GFM::readIntoMemory (gfm.h:5917-6332), the SA sample file (.2, gfm.h:6334-6432),
// HGFM/LocalGFM::readIntoMemory (hgfm.h:2560-2640, 1105-1400; 16-bit words) and BitPairReference
// (reference.cpp:101-190; RefRecord ref_read.h:73-103).  Nothing is re-encoded: sides are uploaded as
// they are on disk, because a 64 B (linear) / 128 B (graph) side already is the HBM transaction unit and
// keeps the Occ checkpoint in the same line as the BWT payload.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <ctype.h>
#include <string>
#include <algorithm>
#include <vector>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace h2g {

struct GfmParams {  // GFMParams::init gfm.h:138-185
	uint32_t len = 0, gbwtLen = 0, numNodes = 0;
	int32_t lineRate = 0, offRate = 0, ftabChars = 0;
	uint32_t eftabLen = 0;
	bool linear = true;
	uint32_t offMask = 0, ftabLen = 0, offsLen = 0, sideSz = 0, sideGbwtSz = 0, sideGbwtLen = 0, numSides = 0;
	uint64_t gbwtTotLen = 0;
	int wsz = 4;
	void init(uint32_t len_, uint32_t gbwtLen_, uint32_t numNodes_, int32_t lineRate_, int32_t offRate_,
	          int32_t ftabChars_, uint32_t eftabLen_, int wsz_) {
		wsz = wsz_;
		const uint32_t wmax = wsz == 4 ? 0xffffffffu : 0xffffu;
		linear = (((len_ + 1) & wmax) == gbwtLen_ || gbwtLen_ == 0);
		len = len_;
		gbwtLen = gbwtLen_ == 0 ? len_ + 1 : gbwtLen_;
		numNodes = numNodes_ == 0 ? len_ + 1 : numNodes_;
		const uint32_t gbwtSz = linear ? gbwtLen / 4 + 1 : gbwtLen / 2 + 1;
		lineRate = lineRate_; offRate = offRate_; ftabChars = ftabChars_; eftabLen = eftabLen_;
		offMask = (wmax << offRate) & wmax;
		ftabLen = (1u << (ftabChars * 2)) + 1;
		offsLen = (numNodes + (1u << offRate) - 1) >> offRate;
		sideSz = 1u << lineRate;
		sideGbwtSz = sideSz - wsz * (linear ? 4 : 6);
		sideGbwtLen = linear ? sideGbwtSz << 2 : sideGbwtSz << 1;
		numSides = (gbwtSz + sideGbwtSz - 1) / sideGbwtSz;
		gbwtTotLen = (uint64_t)numSides * sideSz;
	}
};

struct HostGfm {
	GfmParams p;
	std::vector<uint32_t> plen, rstarts, zOffs, ftab, eftab, offs;   // widened to u32 for local indexes
	std::vector<uint8_t> sides;
	uint32_t fchr[5] = {0, 0, 0, 0, 0};
	uint32_t nPat = 0, nFrag = 0;
	uint32_t tidx = 0, localOffset = 0, joinedOffset = 0;           // LocalGFM only
};

struct HostRef {  // BitPairReference
	std::vector<uint32_t> rec_start;    // text offset where record i's unambiguous stretch begins
	std::vector<uint32_t> rec_len;      // its length
	std::vector<uint32_t> rec_bufoff;   // unambiguous bases preceding it in buf (cumUnambig_)
	std::vector<uint32_t> refRecOffs;   // [nrefs+1]
	std::vector<uint32_t> refLens;      // [nrefs] approxLen (excludes trailing Ns)
	std::vector<uint8_t> buf;           // .4.ht2: 2 bit/base, LSB-first
	uint32_t nrefs = 0;
};

// ALT (alt.h:41-120); same layout as the device DAlt (h2g_graph.h)
struct HostAlt { uint32_t pos, type, len, pad; uint64_t seq; };

struct HostIndex {
	std::vector<HostAlt> alts;          // ALTDB::alts() as GFM::GFM leaves it (gfm.h:728-905)
	HostGfm g;
	HostRef r;
	std::vector<HostGfm> local;
	std::vector<uint32_t> local_first;  // [nPat+1]
	std::vector<std::string> names;
	std::vector<std::string> alt_names;  // ALTDB::altnames() (.8.ht2), permuted like `alts`
	// ALTDB::haplotypes() (.7.ht2 after the ALTs, gfm.h:779-793, :907-921): [left, right] in joined coordinates, sorted by (left, right);
	// hap_ids[hap_first[h] .. hap_first[h+1]) = positions in `alts` of the SNPs the haplotype carries, in ascending order;
	// hap_maxright[h] = max right of haplotypes 0..h.  An index built without --haplotype has one haplotype per SNP (gfm.h:1645).
	std::vector<uint32_t> hap_left, hap_right, hap_maxright, hap_first, hap_ids;
	uint32_t minK = 0;
};

// A file of the index, mapped read-only: what is parsed is read, what is copied out is copied once, what a caller has no use for (the SAM
// formatter needs names, lengths and ALTs, not a gigabyte of sides) is never touched.
struct FileView {
	const uint8_t* p = nullptr; size_t n = 0;
	size_t size() const { return n; }
	const uint8_t& operator[](size_t i) const { return p[i]; }
	const uint8_t* begin() const { return p; }
	~FileView() { if(p) munmap((void*)p, n); }
	FileView() {}
	FileView(const FileView&) = delete;
	FileView& operator=(const FileView&) = delete;
	void take(FileView& o) { if(p) munmap((void*)p, n); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
};
// The three big arrays of the global index — sides, SA sample, reference bases: 2.6 GB of a human-size index — left WHERE THEY ARE in the mapped files (load_host_index with a
// BigViews*): a loader that only hands them on to a device has no use for a copy in host vectors (1 GB of memcpy and 1.6 GB of zero-fill + memcpy on the loading thread, round 6).
struct BigViews {
	FileView f1, f2, f4;                 // keep the mappings alive
	const uint8_t* sides = nullptr; size_t sides_n = 0;
	const uint8_t* offs = nullptr;  size_t offs_n = 0;     // bytes (u32 entries)
	const uint8_t* buf = nullptr;   size_t buf_n = 0;
};
class Reader {
public:
	FileView d;
	size_t pos = 0;
	bool open(const std::string& fn) {
		const int fd = ::open(fn.c_str(), O_RDONLY);
		if(fd < 0) return false;
		struct stat st;
		if(fstat(fd, &st) != 0) { ::close(fd); return false; }
		d.n = (size_t)st.st_size;
		if(d.n) {
			void* m = mmap(nullptr, d.n, PROT_READ, MAP_PRIVATE, fd, 0);
			if(m == MAP_FAILED) { ::close(fd); d.n = 0; return false; }
			d.p = (const uint8_t*)m;
		}
		::close(fd);
		return true;
	}
	bool has(size_t n) const { return pos + n <= d.size(); }
	uint32_t u32() { uint32_t v = 0; if(has(4)) memcpy(&v, &d[pos], 4); else bad = true; pos += 4; return v; }
	uint32_t u16() { uint16_t v = 0; if(has(2)) memcpy(&v, &d[pos], 2); else bad = true; pos += 2; return v; }
	uint32_t w(int wsz) { return wsz == 4 ? u32() : u16(); }
	void arr(std::vector<uint32_t>& a, int wsz, size_t n) {
		a.resize(n);
		if(!has(n * wsz)) { bad = true; return; }
		if(wsz == 4) { if(n) memcpy(a.data(), &d[pos], n * 4); pos += n * 4; }
		else for(size_t i = 0; i < n; i++) a[i] = u16();
	}
	bool bad = false;
};

inline bool read_gfm_body(Reader& b, HostGfm& g, bool light = false, const uint8_t** sides_view = nullptr) {
	const int wsz = g.p.wsz;
	g.nPat = b.w(wsz);
	b.arr(g.plen, wsz, g.nPat);
	g.nFrag = b.w(wsz);
	b.arr(g.rstarts, wsz, (size_t)g.nFrag * 3);
	if(!b.has(g.p.gbwtTotLen)) return false;
	if(sides_view) *sides_view = b.d.begin() + b.pos;
	else if(!light) g.sides.assign(b.d.begin() + b.pos, b.d.begin() + b.pos + g.p.gbwtTotLen);
	b.pos += g.p.gbwtTotLen;
	uint32_t nZ = b.w(wsz);
	b.arr(g.zOffs, wsz, nZ);
	for(int i = 0; i < 5; i++) g.fchr[i] = b.w(wsz);
	if(light) { b.pos += ((size_t)g.p.ftabLen + g.p.eftabLen) * wsz; if(b.pos > b.d.size()) b.bad = true; return !b.bad; }
	b.arr(g.ftab, wsz, g.p.ftabLen);
	b.arr(g.eftab, wsz, g.p.eftabLen);
	return !b.bad;
}

// returns 0 ok, -1 io, -2 format.  light: names, lengths, fragment table, reference records and ALTs only (no sides, SA sample, ftab or
// reference bases: the SAM formatter's view of an index)
inline int load_host_index(const std::string& base, bool load_local, HostIndex& ix, bool light = false, BigViews* bv = nullptr) {
	Reader b1, b2, b3, b4;
	if(!b1.open(base + ".1.ht2") || !b2.open(base + ".2.ht2") || !b3.open(base + ".3.ht2") || !b4.open(base + ".4.ht2")) return -1;
	if(b1.u32() != 1) return -2;
	b1.u32();  // version
	uint32_t len = b1.u32(), gbwtLen = b1.u32(), numNodes = b1.u32();
	int32_t lineRate = (int32_t)b1.u32(); b1.u32();
	int32_t offRate = (int32_t)b1.u32(), ftabChars = (int32_t)b1.u32();
	uint32_t eftabLen = b1.u32(); b1.u32();
	if(lineRate < 6 || lineRate > 8 || ftabChars < 1 || ftabChars > 14 || offRate < 0 || offRate > 16) return -2;
	ix.g.p.init(len, gbwtLen, numNodes, lineRate, offRate, ftabChars, eftabLen, 4);
	if(!read_gfm_body(b1, ix.g, light, bv ? &bv->sides : nullptr)) return -2;
	if(bv) bv->sides_n = (size_t)ix.g.p.gbwtTotLen;
	{   // reference names, '\n'-separated, '\0'-terminated
		std::string cur;
		while(b1.pos < b1.d.size()) {
			char c = (char)b1.d[b1.pos++];
			if(c == '\0') { if(!cur.empty()) ix.names.push_back(cur); break; }
			if(c == '\n') { ix.names.push_back(cur); cur.clear(); } else cur.push_back(c);
		}
	}
	// the SA sample (0.8 GB of a human-size index) and the reference bases (0.8 GB) are copied on threads of their own while this one parses on
	std::thread t_offs, t_buf;
	struct Join { std::thread &a, &b; ~Join() { if(a.joinable()) a.join(); if(b.joinable()) b.join(); } } join_{t_offs, t_buf};
	if(!light && bv) {
		b2.u32();
		if(!b2.has((size_t)ix.g.p.offsLen * 4)) return -2;
		bv->offs = &b2.d[b2.pos]; bv->offs_n = (size_t)ix.g.p.offsLen * 4;
		bv->buf = b4.d.begin(); bv->buf_n = b4.d.size();
	} else if(!light) {
		b2.u32();
		if(!b2.has((size_t)ix.g.p.offsLen * 4)) return -2;
		ix.g.offs.resize(ix.g.p.offsLen);
		t_offs = std::thread([&]() { if(ix.g.p.offsLen) memcpy(ix.g.offs.data(), &b2.d[b2.pos], (size_t)ix.g.p.offsLen * 4); });
		ix.r.buf.resize(b4.d.size() + 16, 0);
		t_buf = std::thread([&]() { if(b4.d.size()) memcpy(ix.r.buf.data(), b4.d.begin(), b4.d.size()); });
	}
	// reference records
	if(b3.u32() != 1) return -2;
	uint32_t nrecs = b3.u32();
	HostRef& r = ix.r;
	uint64_t cumsz = 0, cumlen = 0;
	for(uint32_t i = 0; i < nrecs; i++) {
		uint32_t off = b3.u32(), rlen = b3.u32();
		if(!b3.has(1)) return -2;
		bool first = b3.d[b3.pos++] != 0;
		if(first) {
			r.refRecOffs.push_back(i);
			if(r.nrefs > 0) r.refLens.push_back((uint32_t)cumlen);
			cumlen = 0;
			r.nrefs++;
		} else if(i == 0) return -2;
		cumlen += off;
		r.rec_start.push_back((uint32_t)cumlen);
		r.rec_len.push_back(rlen);
		r.rec_bufoff.push_back((uint32_t)cumsz);
		cumsz += rlen;
		cumlen += rlen;
	}
	if(b3.bad || r.nrefs == 0) return -2;
	r.refRecOffs.push_back(nrecs);
	r.refLens.push_back((uint32_t)cumlen);
	if(load_local) {
		Reader b5, b6;
		if(b5.open(base + ".5.ht2") && b6.open(base + ".6.ht2")) {
			b5.u32(); b6.u32();
			uint32_t nlocal = b5.u32();
			int32_t llr = (int32_t)b5.u32(); b5.u32();
			int32_t lor = (int32_t)b5.u32(), lfc = (int32_t)b5.u32(); b5.u32();
			ix.local.resize(nlocal);
			ix.local_first.clear();
			for(uint32_t i = 0; i < nlocal; i++) {
				HostGfm& l = ix.local[i];
				l.tidx = b5.u32(); l.localOffset = b5.u32(); l.joinedOffset = b5.u32();
				uint32_t llen = b5.u16(), lgl = b5.u16(), lnn = b5.u16(), lel = b5.u16();
				l.p.init(llen, lgl, lnn, llr, lor, lfc, lel, 2);
				while(ix.local_first.size() <= l.tidx) ix.local_first.push_back(i);
				if(llen == 0) continue;
				if(!read_gfm_body(b5, l)) return -2;
				b6.arr(l.offs, 2, l.p.offsLen);
				if(b6.bad) return -2;
			}
			while(ix.local_first.size() <= ix.g.nPat) ix.local_first.push_back(nlocal);
		}
	}
	// .7.ht2: ALTs (gfm.h:728-905).  Every deletion gets a reversed copy (pos = last deleted base, low byte of seq = 1),
	// every splice site a mirrored copy, then the list is sorted by (ALT::operator< alt.h:88-102, original index).
	{
		Reader b7;
		ix.alts.clear();
		if(b7.open(base + ".7.ht2") && b7.d.size() >= 8) {
			b7.u32();
			const uint32_t n = b7.u32();
			std::vector<std::pair<HostAlt, uint32_t> > v;
			for(uint32_t i = 0; i < n && b7.has(20); i++) {
				HostAlt a;
				a.pos = b7.u32(); a.type = b7.u32(); a.len = b7.u32(); a.pad = 0;
				memcpy(&a.seq, &b7.d[b7.pos], 8); b7.pos += 8;
				v.push_back(std::make_pair(a, (uint32_t)v.size()));
			}
			const size_t n0 = v.size();
			// .8.ht2: i32 endian, index_t count, then whitespace-separated names (gfm.h:736-758); appended copies inherit
			// the name of the deletion they mirror ("ssr" for splice sites, gfm.h:879-885)
			std::vector<std::string> nm(n0);
			{
				Reader b8;
				if(b8.open(base + ".8.ht2") && b8.d.size() >= 8) {
					size_t p = 8, k = 0;
					while(k < n0 && p < b8.d.size()) {
						while(p < b8.d.size() && isspace((unsigned char)b8.d[p])) p++;
						std::string s;
						while(p < b8.d.size() && !isspace((unsigned char)b8.d[p])) s.push_back((char)b8.d[p++]);
						if(s.empty()) break;
						nm[k++] = s;
					}
				}
			}
			for(size_t i = 0; i < n0; i++) {
				HostAlt a = v[i].first;
				if(a.type == 3) { a.pos = a.pos + a.len - 1; a.seq = (a.seq & ~0xffull) | 1; v.push_back(std::make_pair(a, (uint32_t)v.size())); nm.push_back(nm[i]); }
				else if(a.type == 5) { std::swap(a.pos, a.len); v.push_back(std::make_pair(a, (uint32_t)v.size())); nm.push_back("ssr"); }
			}
			std::sort(v.begin(), v.end(), [](const std::pair<HostAlt, uint32_t>& x, const std::pair<HostAlt, uint32_t>& y) {
				const HostAlt &a = x.first, &b = y.first;
				if(a.pos != b.pos) return a.pos < b.pos;
				if(a.type != b.type) { if(a.type == 2) return true; if(b.type == 2) return false; return a.type < b.type; }
				if(a.len != b.len) return a.len < b.len;
				if(a.seq != b.seq) return a.seq < b.seq;
				return x.second < y.second;
			});
			ix.alt_names.clear();
			for(auto& e : v) { ix.alts.push_back(e.first); ix.alt_names.push_back(nm[e.second]); }
			ix.hap_left.clear(); ix.hap_right.clear(); ix.hap_maxright.clear(); ix.hap_first.assign(1, 0); ix.hap_ids.clear();
			if(b7.has(4)) {                                   // older indexes end after the ALTs
				std::vector<uint32_t> to_alti(n0, 0xffffffffu);    // file order -> position after the sort
				for(size_t i = 0; i < v.size(); i++) if(v[i].second < n0) to_alti[v[i].second] = (uint32_t)i;
				const uint32_t nh = b7.u32();
				for(uint32_t h = 0; h < nh && b7.has(12); h++) {
					const uint32_t l = b7.u32(), r = b7.u32(), k = b7.u32();
					if(!b7.has((size_t)k * 4)) break;
					ix.hap_left.push_back(l); ix.hap_right.push_back(r);
					ix.hap_maxright.push_back(h == 0 ? r : std::max(ix.hap_maxright.back(), r));
					for(uint32_t q = 0; q < k; q++) { const uint32_t id = b7.u32(); ix.hap_ids.push_back(id < n0 ? to_alti[id] : 0xffffffffu); }
					ix.hap_first.push_back((uint32_t)ix.hap_ids.size());
				}
			}
		}
	}
	if(bv) { bv->f1.take(b1.d); bv->f2.take(b2.d); bv->f4.take(b4.d); }      // (the views point into these mappings)
	uint32_t gl = ix.g.p.len;
	ix.minK = 0;
	while(gl > 0) { gl >>= 2; ix.minK++; }   // hi_aligner.h:3979-3984
	return 0;
}

}  // namespace h2g
