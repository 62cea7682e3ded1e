// capacities of the *_big go() units: the second pass over the reads whose lists overflowed the default workspace, and option
// sets beyond the default capacities (-k up to 30 = --very-sensitive, --max-seeds up to 64).  Depth 128 is the reference's
// own recursion limit (spliced_aligner.h:369).
#pragma once
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
