// h2g_local_pack.h — host-side packing of the local (.5/.6.ht2) indexes into three flat arrays + descriptors
// (DLocalSet, h2g_align.h): sides stay byte-for-byte as on disk; ftab/eftab/offs/rstarts are stored as the
// 16-bit words they are in the file (LocalGFM, hgfm.h:35).
#pragma once
#include <vector>
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

}  // namespace h2g
