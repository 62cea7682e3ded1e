// h2g_local_pack.h — host-side packing of the local (.5/.6.ht2) indexes into three flat arrays + descriptors
// (DLocalSet, h2g_align.h): sides stay byte-for-byte as on disk; ftab/eftab/offs/rstarts are stored as the
// 16-bit words they are in the file (LocalGFM, hgfm.h:35).
#pragma once
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <thread>
#include "h2g_align.h"
#include "h2g_host_index.h"

namespace h2g {

struct LocalPack {
	std::vector<DLocalDesc> desc;
	std::vector<uint8_t> sides;
	std::vector<uint16_t> words;
	std::vector<uint32_t> first, zoffs;
	uint32_t ftabChars = 6, offRate = 3;
	DLocalSet view(const DLocalDesc* d, const uint8_t* s, const uint16_t* w, const uint32_t* f, const uint32_t* z) const {
		DLocalSet v;
		v.desc = d; v.sides = s; v.words = w; v.first = f; v.zoffs = z; v.n = (uint32_t)desc.size(); v.ftabChars = ftabChars; v.offRate = offRate;
		return v;
	}
};

inline void pack_local(const HostIndex& ix, LocalPack& lp) {
	lp.desc.clear(); lp.sides.clear(); lp.words.clear(); lp.zoffs.clear();
	lp.first = ix.local_first;
	if(lp.first.empty()) lp.first.assign(ix.g.nPat + 1, 0);
	for(const HostGfm& l : ix.local) {
		DLocalDesc d;
		memset(&d, 0, sizeof d);
		d.len = l.p.len; d.gbwtLen = l.p.gbwtLen; d.eftabLen = l.p.eftabLen; d.nFrag = l.nFrag;
		d.nZ = (uint32_t)l.zOffs.size(); d.zoff = l.zOffs.empty() ? H2G_MAX : l.zOffs[0];
		d.zoffs_off = (uint32_t)lp.zoffs.size();
		lp.zoffs.insert(lp.zoffs.end(), l.zOffs.begin(), l.zOffs.end());
		d.tidx = l.tidx; d.localOffset = l.localOffset; d.joinedOffset = l.joinedOffset;
		for(int i = 0; i < 5; i++) d.fchr[i] = l.fchr[i];
		d.ftabLim = l.p.linear ? l.p.len : l.p.gbwtLen;
		if(l.p.len > 0) {
			lp.ftabChars = (uint32_t)l.p.ftabChars; lp.offRate = (uint32_t)l.p.offRate;
			while(lp.sides.size() % 128) lp.sides.push_back(0);
			d.sides_off = lp.sides.size();
			d.sides_bytes = (uint32_t)l.sides.size();
			lp.sides.insert(lp.sides.end(), l.sides.begin(), l.sides.end());
			auto put = [&](const std::vector<uint32_t>& v) { uint32_t off = (uint32_t)lp.words.size(); for(uint32_t x : v) lp.words.push_back((uint16_t)x); return off; };
			d.ftab_off = put(l.ftab); d.eftab_off = put(l.eftab); d.offs_off = put(l.offs); d.rstarts_off = put(l.rstarts);
		}
		lp.desc.push_back(d);
	}
	lp.sides.resize(lp.sides.size() + 256, 0);
	lp.words.resize(lp.words.size() + 64, 0);
	lp.zoffs.resize(lp.zoffs.size() + 4, H2G_MAX);
}

// The same LocalPack straight from the files (.5.ht2 / .6.ht2), without the per-index HostGfm objects: one sequential walk over the
// ~55 000 local headers of a human-size index fixes every source and destination offset, then the sides and the 16-bit word arrays are
// copied by `nthreads` threads (the file's u16 words ARE the packed form).  Byte-identical to pack_local(load_host_index(..)) —
// tests/test_abi.py compares the two — in a fraction of its time (the old path widened every word to 32 bits and pushed it back one by one).
// Returns 0 ok, -1 io, -2 format; `nPat` pads LocalPack::first.
struct MappedFile {
	const uint8_t* p = nullptr; size_t n = 0;
	bool open(const std::string& fn) {
		const int fd = ::open(fn.c_str(), O_RDONLY);
		if(fd < 0) return false;
		struct stat st;
		if(fstat(fd, &st) != 0) { ::close(fd); return false; }
		n = (size_t)st.st_size;
		if(n) { void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0); if(m == MAP_FAILED) { ::close(fd); n = 0; return false; } p = (const uint8_t*)m; }
		::close(fd);
		return true;
	}
	~MappedFile() { if(p) munmap((void*)p, n); }
};
// one local index's pieces: where they lie in the files and where they go in the packed arrays (word arrays in packed order: ftab, eftab, offs (file 6), rstarts)
struct LocalJob { size_t src_sides, dst_sides, nsides, src_w[4], dst_w, nw[4]; };
// The walk over the headers alone: descriptors, '$' rows, first-index table, the jobs (ascending destinations) and the sizes of the two packed arrays without their padding.
// load_local_pack copies the jobs into host vectors; the device loader (h2g_kernels.hip) streams them through its staging buffers instead and never builds the 2.1 GB of host arrays.
struct LocalPlan { MappedFile f5, f6; std::vector<LocalJob> jobs; size_t nsides_tot = 0, nwords_tot = 0; };
inline int plan_local_pack(const std::string& base, uint32_t nPat, LocalPack& lp, LocalPlan& pl) {
	MappedFile& f5 = pl.f5; MappedFile& f6 = pl.f6;
	if(!f5.open(base + ".5.ht2") || !f6.open(base + ".6.ht2")) return -1;
	auto u32at = [](const MappedFile& f, size_t at, bool* bad) { uint32_t v = 0; if(at + 4 <= f.n) memcpy(&v, f.p + at, 4); else *bad = true; return v; };
	auto u16at = [](const MappedFile& f, size_t at, bool* bad) { uint16_t v = 0; if(at + 2 <= f.n) memcpy(&v, f.p + at, 2); else *bad = true; return (uint32_t)v; };
	bool bad = false;
	size_t p5 = 4, p6 = 4;                                   // (the endianness word of each file)
	const uint32_t nlocal = u32at(f5, p5, &bad); p5 += 4;
	const int32_t llr = (int32_t)u32at(f5, p5, &bad); p5 += 8;
	const int32_t lor = (int32_t)u32at(f5, p5, &bad); p5 += 4;
	const int32_t lfc = (int32_t)u32at(f5, p5, &bad); p5 += 8;
	if(bad) return -2;
	if((size_t)nlocal > f5.n / 20) return -2;                // every local-index header is at least 20 bytes: a count beyond that is a corrupt / truncated file (never reserve() on it)
	typedef LocalJob Job;
	std::vector<Job>& jobs = pl.jobs;
	jobs.clear();
	lp.desc.clear(); lp.zoffs.clear(); lp.first.clear();
	lp.desc.reserve(nlocal); jobs.reserve(nlocal);
	size_t nsides_tot = 0, nwords_tot = 0;
	for(uint32_t i = 0; i < nlocal; i++) {
		DLocalDesc d;
		memset(&d, 0, sizeof d);
		d.tidx = u32at(f5, p5, &bad); d.localOffset = u32at(f5, p5 + 4, &bad); d.joinedOffset = u32at(f5, p5 + 8, &bad); p5 += 12;
		const uint32_t llen = u16at(f5, p5, &bad), lgl = u16at(f5, p5 + 2, &bad), lnn = u16at(f5, p5 + 4, &bad), lel = u16at(f5, p5 + 6, &bad); p5 += 8;
		if(bad) return -2;
		GfmParams P;
		P.init(llen, lgl, lnn, llr, lor, lfc, lel, 2);
		while(lp.first.size() <= d.tidx) lp.first.push_back(i);
		d.len = P.len; d.gbwtLen = P.gbwtLen; d.eftabLen = P.eftabLen;
		d.zoff = H2G_MAX; d.zoffs_off = (uint32_t)lp.zoffs.size();
		d.ftabLim = P.linear ? P.len : P.gbwtLen;
		if(llen > 0) {
			Job j;
			memset(&j, 0, sizeof j);
			const uint32_t np = u16at(f5, p5, &bad); p5 += 2 + (size_t)np * 2;                 // nPat, plen
			d.nFrag = u16at(f5, p5, &bad); p5 += 2;
			j.src_w[3] = p5; j.nw[3] = (size_t)d.nFrag * 3; p5 += j.nw[3] * 2;              // rstarts
			j.src_sides = p5; j.nsides = (size_t)P.gbwtTotLen; p5 += j.nsides;
			d.nZ = u16at(f5, p5, &bad); p5 += 2;
			for(uint32_t z = 0; z < d.nZ; z++) { const uint32_t v = u16at(f5, p5, &bad); p5 += 2; if(z == 0) d.zoff = v; lp.zoffs.push_back(v); }
			for(int c = 0; c < 5; c++) { d.fchr[c] = u16at(f5, p5, &bad); p5 += 2; }
			j.src_w[0] = p5; j.nw[0] = P.ftabLen; p5 += j.nw[0] * 2;
			j.src_w[1] = p5; j.nw[1] = P.eftabLen; p5 += j.nw[1] * 2;
			j.src_w[2] = p6; j.nw[2] = P.offsLen; p6 += j.nw[2] * 2;
			if(bad || p5 > f5.n || p6 > f6.n) return -2;
			lp.ftabChars = (uint32_t)P.ftabChars; lp.offRate = (uint32_t)P.offRate;
			nsides_tot = (nsides_tot + 127) & ~(size_t)127;
			d.sides_off = nsides_tot; d.sides_bytes = (uint32_t)j.nsides;
			j.dst_sides = nsides_tot; nsides_tot += j.nsides;
			j.dst_w = nwords_tot;
			d.ftab_off = (uint32_t)nwords_tot; d.eftab_off = d.ftab_off + (uint32_t)j.nw[0]; d.offs_off = d.eftab_off + (uint32_t)j.nw[1]; d.rstarts_off = d.offs_off + (uint32_t)j.nw[2];
			nwords_tot += j.nw[0] + j.nw[1] + j.nw[2] + j.nw[3];
			jobs.push_back(j);
		}
		lp.desc.push_back(d);
	}
	while(lp.first.size() <= nPat) lp.first.push_back(nlocal);
	if(lp.first.empty()) lp.first.assign(nPat + 1, 0);
	lp.zoffs.resize(lp.zoffs.size() + 4, H2G_MAX);
	pl.nsides_tot = nsides_tot; pl.nwords_tot = nwords_tot;
	return 0;
}
// bytes [off, off + len) of the packed SIDES array (LocalPlan) into `out`: the pieces of the jobs that overlap the range, zeros in the alignment gaps and behind the last job
inline void local_fill_sides(const LocalPlan& pl, uint8_t* out, size_t off, size_t len) {
	memset(out, 0, len);
	const std::vector<LocalJob>& J = pl.jobs;
	size_t lo = 0, hi = J.size();
	while(lo < hi) { const size_t m = (lo + hi) / 2; if(J[m].dst_sides + J[m].nsides <= off) lo = m + 1; else hi = m; }
	for(size_t k = lo; k < J.size() && J[k].dst_sides < off + len; k++) {
		const size_t a = J[k].dst_sides > off ? J[k].dst_sides : off, b = J[k].dst_sides + J[k].nsides < off + len ? J[k].dst_sides + J[k].nsides : off + len;
		if(b > a) memcpy(out + (a - off), pl.f5.p + J[k].src_sides + (a - J[k].dst_sides), b - a);
	}
}
// ... and of the packed 16-bit WORD array (byte offsets)
inline void local_fill_words(const LocalPlan& pl, uint8_t* out, size_t off, size_t len) {
	memset(out, 0, len);
	const std::vector<LocalJob>& J = pl.jobs;
	auto jend = [&](const LocalJob& j) { return (j.dst_w + j.nw[0] + j.nw[1] + j.nw[2] + j.nw[3]) * 2; };
	size_t lo = 0, hi = J.size();
	while(lo < hi) { const size_t m = (lo + hi) / 2; if(jend(J[m]) <= off) lo = m + 1; else hi = m; }
	for(size_t k = lo; k < J.size() && J[k].dst_w * 2 < off + len; k++) {
		size_t w = J[k].dst_w * 2;
		for(int a = 0; a < 4; a++) {
			const size_t nb = J[k].nw[a] * 2;
			const uint8_t* src = (a == 2 ? pl.f6.p : pl.f5.p) + J[k].src_w[a];
			const size_t x = w > off ? w : off, y = w + nb < off + len ? w + nb : off + len;
			if(y > x) memcpy(out + (x - off), src + (x - w), y - x);
			w += nb;
		}
	}
}
inline int load_local_pack(const std::string& base, uint32_t nPat, LocalPack& lp, unsigned nthreads = 8) {
	LocalPlan pl;
	const int prc = plan_local_pack(base, nPat, lp, pl);
	if(prc) return prc;
	const MappedFile& f5 = pl.f5; const MappedFile& f6 = pl.f6;
	typedef LocalJob Job;
	const std::vector<Job>& jobs = pl.jobs;
	lp.sides.assign(pl.nsides_tot + 256, 0);
	lp.words.assign(pl.nwords_tot + 64, 0);
	if(nthreads < 1) nthreads = 1;
	std::vector<std::thread> th;
	for(unsigned t = 0; t < nthreads; t++) th.emplace_back([&, t]() {
		for(size_t k = t; k < jobs.size(); k += nthreads) {
			const Job& j = jobs[k];
			memcpy(lp.sides.data() + j.dst_sides, f5.p + j.src_sides, j.nsides);
			size_t w = j.dst_w;
			for(int a = 0; a < 4; a++) {
				const MappedFile& f = a == 2 ? f6 : f5;
				if(j.nw[a]) memcpy(lp.words.data() + w, f.p + j.src_w[a], j.nw[a] * 2);
				w += j.nw[a];
			}
		}
	});
	for(std::thread& x : th) x.join();
	return 0;
}

}  // namespace h2g
