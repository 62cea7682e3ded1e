// capacities of the *_big go() units: the second pass over the reads whose lists overflowed the default workspace, and option
// sets beyond the default capacities (-k up to 30 = --very-sensitive, --max-seeds up to 64).  Depth 128 is the reference's
// own recursion limit (spliced_aligner.h:369).
#pragma once
// ... and the reference's edit lists are unbounded (hi_aligner.h:421 LinkedEList<EList<Edit>>; a deletion of n bases is n edits, edit.h): the working hit of
// these units holds H2G_GHIT_EDITS of them (include/h2g.h: a per-translation-unit capacity; 32 in the default units, whose flagged reads come here).  A
// record beyond the 32 inline entries of h2g_alnres leaves through the long-edit area (MachOut::ledits).  192 edits = a 101-base read at --score-min L,0,-5.7
// all in deletions; one extension may add 160 (H2G_NEW_EDITS: the mismatches --score-min L,0,-3 buys a 250-base read).
#ifndef H2G_GHIT_EDITS
#define H2G_GHIT_EDITS 192
#endif
#ifndef H2G_NEW_EDITS
#define H2G_NEW_EDITS 160
#endif
// (a slot of these units is 6.1 MB: 256 reads in flight per workgroup keep the second pass's pool near its former size)
#ifndef H2G_GO_SLOTS
#define H2G_GO_SLOTS 256
#endif
#define AL_MAX_GHITS     64
#define AL_MAX_SEARCHED  512
#define AL_MAX_RESULTS   128
#define AL_MAX_DEPTH     128
#define AL_MAX_LOCALHITS 8
#define AL_MAX_COORDS    24
#define AL_MAX_PARTIAL   64
// graph walk / ALT extension scratch (h2g_graph.h)
#define H2G_GW_MAXELT    96
#define H2G_GW_MAXST     160
#define H2G_GW_MAXROWS   64      // fixed: the row masks of the group walk are 64-bit words
#define H2G_AWA_DEPTH    24
#define H2G_AWA_CAND     8
#define H2G_OFFDIFF_CAP  64
