// binding_check.cpp — TEST INFRASTRUCTURE: the reference-side binding of INTEGRATION.md §3a, compiled against the reference's own
// headers (oracle/Makefile.ref builds it as an object next to the reference's; nothing here is part of the product and nothing of
// the reference is copied: the headers are included from $(REF) where they lie).  It proves that what INTEGRATION.md tells a
// maintainer to write next to hisat2.cpp:3540 is well-formed C++ against HISAT2 2.2.3: the device's report events (h2g_alnres)
// become the AlnRes objects reportHit (hi_aligner.h:6129-6166) builds, in the order given, and go to AlnSinkWrap::report
// (aln_sink.h:1153); finishRead / MAPQ / SAM stay the reference's code.
#include <vector>
#include "aln_sink.h"
#include "aligner_result.h"
#include "edit.h"
#include "read.h"
#include "ds.h"
#include "../include/h2g.h"

typedef uint32_t h2g_bind_index_t;   // index_t of the small-index build (hisat2.cpp: typedef TIndexOffU index_t)

// Edit list of one device record (hi_aligner.h:6167-6175 leaves them 5'-relative; so does the device)
static void edits_of(const h2g_alnres& a, EList<Edit>& out) {
	out.clear();
	for(uint32_t k = 0; k < a.nedits; k++) {
		const h2g_edit& e = a.edits[k];
		out.push_back(Edit(e.pos, (int)e.chr, (int)e.qchr, (int)e.type, true, e.snp));
	}
}

// One unpaired batch: rr / al as h2g_align_fetch returns them; `reads` in read-id order starting at rdid0.
// Returns the number of reads the caller has to run through the reference's own go() (fixed-capacity list overflow).
template <typename index_t>
size_t h2g_replay_unpaired(AlnSinkWrap<index_t>& msinkwrap, const EList<Read>& reads, TReadId rdid0, bool qualitiesMatter,
                           const h2g_read_result* rr, const h2g_alnres* al, const TIndexOffU* plen,
                           LinkedEList<EList<Edit> >& rawEdits)
{
	size_t fallback = 0;
	EList<Edit> edits;
	for(size_t i = 0; i < reads.size(); i++) {
		const Read& rd = reads[i];
		msinkwrap.nextRead(&rd, NULL, rdid0 + i, qualitiesMatter);                       // hisat2.cpp:3358
		if(rr[i].overflow) { fallback++; continue; }                                      // (the caller runs its own go() for this one)
		for(uint32_t k = 0; k < rr[i].nselect; k++) {
			const h2g_alnres& a = al[i * H2G_ALN_CAP + k];
			edits_of(a, edits);
			AlnScore asc((TAlScore)a.score, 0, 0, false, 0, false, false, a.trim5, a.trim3);   // hi_aligner.h:6129
			AlnRes rs;
			rs.init(rd.length(), rd.rdid, asc, &edits, 0, edits.size(), NULL, 0, 0,
			        Coord((TRefId)a.tidx, (TRefOff)a.toff, a.fw != 0), (TRefOff)plen[a.tidx], &rawEdits,
			        -1, -1, -1, 0, -1, -1, false, 0, 0, a.trim5 > 0 || a.trim3 > 0,
			        a.fw ? a.trim5 : a.trim3, a.fw ? a.trim3 : a.trim5, false);
			msinkwrap.report(0, &rs, NULL);                                                // aln_sink.h:2565
		}
	}
	return fallback;
}

template size_t h2g_replay_unpaired<h2g_bind_index_t>(AlnSinkWrap<h2g_bind_index_t>&, const EList<Read>&, TReadId, bool,
                                                      const h2g_read_result*, const h2g_alnres*, const TIndexOffU*, LinkedEList<EList<Edit> >&);
