// Wave-level time split of the fast pass's loop (h2g_k_go_fast.hip), compiled in with -DH2G_GO_PROF only (tools/build_prof_lib.sh, tools/fast_perf.py): the
// shipped kernel sees empty macros.  Shader-clock ticks per wave, added to FastArgs::counters[128..]:
//   [0] choose + pop + load   [1] the trip (primitive + control)   [2] store of the reads handed on   [17] push   [16] hand-on list of the trip   [15] new reads
//   [20 + op] slots executed  [32 + op] executions   [44] slots loaded from their slot   [46] slots stepped   [47] trips
//   [176 + site] ticks of the trips of a site's queue, [208 + site] their number
#pragma once
#ifdef H2G_GO_PROF
#define FPROF_DECL \
	unsigned long long prof[48], prof_ctl[32], prof_n[32]; \
	for(int k_ = 0; k_ < 48; k_++) prof[k_] = 0; \
	for(int k_ = 0; k_ < 32; k_++) { prof_ctl[k_] = 0; prof_n[k_] = 0; } \
	uint32_t trip_site = 0; \
	const unsigned long long prof_t0 = wall_clock64(); \
	unsigned long long tp0 = __builtin_readcyclecounter(), tp1
#define FPROF(SLOT) do { tp1 = __builtin_readcyclecounter(); prof[SLOT] += tp1 - tp0; tp0 = tp1; } while(0)
#define FPROF_SITE(Q) do { trip_site = (Q); } while(0)
#define FPROF_EXEC(OP, NACTIVE, NFRESH) do { prof[20 + (OP)] += (NACTIVE); prof[32 + (OP)]++; prof[44] += (NFRESH); } while(0)
#define FPROF_TRIP(NACTIVE) do { prof[46] += (NACTIVE); prof[47]++; } while(0)
// time-resolved: per bin of 2^FPROF_BIN_SHIFT ticks of the 100 MHz clock since the wave started: [512 + 4 bin] trips, [+1] lanes that ran, [+2] slots in flight in the workgroup (summed per trip)
#ifndef FPROF_BIN_SHIFT
#define FPROF_BIN_SHIFT (FG_GRAPH ? 18 : 15)
#endif
#define FPROF_TBIN(CNT, NACTIVE, NINFLIGHT) do { const unsigned long long na_ = (unsigned long long)(NACTIVE); if(lane == 0) { unsigned long long b_ = (wall_clock64() - prof_t0) >> FPROF_BIN_SHIFT; if(b_ > 63) b_ = 63; \
	atomicAdd((CNT) + 512 + 4 * b_, 1ull); atomicAdd((CNT) + 513 + 4 * b_, na_); atomicAdd((CNT) + 514 + 4 * b_, (unsigned long long)(NINFLIGHT)); } } while(0)
#define FPROF_CTL() do { const unsigned long long t_ = __builtin_readcyclecounter(); prof_ctl[trip_site & 31] += t_ - tp0; prof_n[trip_site & 31]++; } while(0)
#define FPROF_FLUSH(CNT) do { \
	if(lane == 0) for(int k_ = 0; k_ < 48; k_++) if(prof[k_]) atomicAdd((CNT) + 128 + k_, prof[k_]); \
	if(lane == 0) for(int k_ = 0; k_ < 32; k_++) if(prof_n[k_]) { atomicAdd((CNT) + 176 + k_, prof_ctl[k_]); atomicAdd((CNT) + 208 + k_, prof_n[k_]); } } while(0)
#else
#define FPROF_DECL do {} while(0)
#define FPROF(SLOT) do {} while(0)
#define FPROF_SITE(Q) do {} while(0)
#define FPROF_EXEC(OP, NACTIVE, NFRESH) do {} while(0)
#define FPROF_TRIP(NACTIVE) do {} while(0)
#define FPROF_CTL() do {} while(0)
#define FPROF_TBIN(CNT, NACTIVE, NINFLIGHT) do {} while(0)
#define FPROF_FLUSH(CNT) do {} while(0)
#endif
