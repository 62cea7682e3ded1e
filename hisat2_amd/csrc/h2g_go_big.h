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
