// Memory-access profile of the go() machine (development tool).  The host instantiation (tests/emul/h2g_emul.cpp) is compiled
// with -fsanitize=kernel-address and out-of-line checks, so every load / store calls __asan_{load,store}N_noabort: those are
// defined here and classify the address by workspace region and machine phase.  Nothing here is part of the product or the tests.
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <unordered_set>
#include "../../hisat2_amd/csrc/h2g_core.h"
#include "../../hisat2_amd/csrc/h2g_align.h"
using namespace h2g;

enum { R_GV, R_FH, R_M, R_SCALARS, R_TMP, R_GHITS, R_AMCO, R_FRAME_HDR, R_FRAME_HIT, R_FRAME_COORDS, R_FRAME_LOCAL, R_PARTIAL, R_SEARCHED, R_RES, R_LANE, R_OTHER, R_N };
static const char* RN[R_N] = {"gv", "fh", "m[]", "scalars", "tmp/tmp2", "ghits", "am_co", "frame.hdr", "frame.hit", "frame.coords", "frame.local_hits", "marr.partial", "marr.searched", "marr.res", "Mach/Lane", "other(index,read,out,stack)"};
#define NPH 20
extern "C" {
const void* g_mp_ws = nullptr; const void* g_mp_mach = nullptr; size_t g_mp_mach_sz = 0;
int g_mp_phase = 0;              // 0 begin, 1 control, 2.. = 2 + op
int g_mp_on = 0;
}
static unsigned long long cnt[NPH][R_N][2], bytes[NPH][R_N][2], ulines[NPH][R_N], trips[NPH];
static std::unordered_set<uint64_t> lines;
static int region(uintptr_t a) {
	const uintptr_t w = (uintptr_t)g_mp_ws;
	if(a >= (uintptr_t)g_mp_mach && a < (uintptr_t)g_mp_mach + g_mp_mach_sz) return R_LANE;
	if(a < w || a >= w + sizeof(AlignWS)) return R_OTHER;
	const size_t o = a - w;
	if(o < offsetof(AlignWS, fh)) return R_GV;
	if(o < offsetof(AlignWS, m)) return R_FH;
	if(o < offsetof(AlignWS, nghits)) return R_M;
	if(o < offsetof(AlignWS, tmp)) return R_SCALARS;
	if(o < offsetof(AlignWS, ghits)) return R_TMP;
	if(o < offsetof(AlignWS, am_co)) return R_GHITS;
	if(o < offsetof(AlignWS, stack)) return R_AMCO;
	if(o < offsetof(AlignWS, marr)) {
		const size_t f = (o - offsetof(AlignWS, stack)) % sizeof(Frame);
		if(f < sizeof(h2g_ghit)) return R_FRAME_HIT;
		if(f < offsetof(Frame, coords)) return R_FRAME_HDR;
		if(f < offsetof(Frame, local_hits)) return R_FRAME_COORDS;
		return R_FRAME_LOCAL;
	}
	const size_t q = (o - offsetof(AlignWS, marr)) % sizeof(MateArr);
	if(q < offsetof(MateArr, searched)) return R_PARTIAL;
	if(q < offsetof(MateArr, res)) return R_SEARCHED;
	return R_RES;
}
static inline void acc(uintptr_t a, size_t n, int st) {
	if(!g_mp_on) return;
	const int r = region(a), p = g_mp_phase < NPH ? g_mp_phase : NPH - 1;
	cnt[p][r][st]++; bytes[p][r][st] += n;
	if(r != R_LANE) { const uint64_t l = a >> 7; if(lines.insert(l).second) ulines[p][r]++; }
}
extern "C" {
void mp_trip() { lines.clear(); trips[g_mp_phase < NPH ? g_mp_phase : NPH - 1]++; }
#define HOOK(N) void __asan_load##N##_noabort(uintptr_t a) { acc(a, N, 0); } void __asan_store##N##_noabort(uintptr_t a) { acc(a, N, 1); } \
                void __asan_load##N(uintptr_t a) { acc(a, N, 0); } void __asan_store##N(uintptr_t a) { acc(a, N, 1); }
HOOK(1) HOOK(2) HOOK(4) HOOK(8) HOOK(16)
void __asan_loadN_noabort(uintptr_t a, size_t n) { acc(a, n, 0); }
void __asan_storeN_noabort(uintptr_t a, size_t n) { acc(a, n, 1); }
void __asan_loadN(uintptr_t a, size_t n) { acc(a, n, 0); }
void __asan_storeN(uintptr_t a, size_t n) { acc(a, n, 1); }
void __asan_handle_no_return() {}
void __asan_init() {}
void __asan_version_mismatch_check_v8() {}
void __asan_register_globals(void*, size_t) {}
void __asan_unregister_globals(void*, size_t) {}
// pc trace (H2G_MACH_PCTRACE): 16-bit stream, 0x8000|op ends a control phase, 0xffff ends a read
static FILE* g_pcf = nullptr;
void mach_pctrace(unsigned pc) { if(!g_pcf) g_pcf = fopen(getenv("H2G_PCTRACE_OUT") ? getenv("H2G_PCTRACE_OUT") : "/tmp/pctrace.bin", "wb"); uint16_t v = (uint16_t)pc; fwrite(&v, 2, 1, g_pcf); }
void mach_pctrace_close() { if(g_pcf) { fclose(g_pcf); g_pcf = nullptr; } }
void mp_report(unsigned nreads, const char** opnames, int nops) {
	unsigned long long tl = 0, ts = 0, tu = 0;
	printf("per read (n = %u), by machine phase x workspace region: loads / stores / first-touch 128 B lines per trip\n", nreads);
	for(int p = 0; p < NPH; p++) {
		unsigned long long pl = 0, ps = 0, pu = 0;
		for(int r = 0; r < R_N; r++) { pl += cnt[p][r][0]; ps += cnt[p][r][1]; pu += ulines[p][r]; }
		if(!pl && !ps) continue;
		const char* pn = p == 0 ? "begin" : p == 1 ? "control" : (p - 2 < nops ? opnames[p - 2] : "?");
		printf("%-10s trips/read %6.2f  loads %8.1f stores %8.1f lines %7.1f\n", pn, (double)trips[p] / nreads, (double)pl / nreads, (double)ps / nreads, (double)pu / nreads);
		for(int r = 0; r < R_N; r++) if(cnt[p][r][0] + cnt[p][r][1])
			printf("    %-28s loads %8.1f stores %8.1f lines %7.1f\n", RN[r], (double)cnt[p][r][0] / nreads, (double)cnt[p][r][1] / nreads, (double)ulines[p][r] / nreads);
		tl += pl; ts += ps; tu += pu;
	}
	printf("TOTAL      loads %8.1f stores %8.1f first-touch lines %7.1f   (Mach/Lane = registers on the device)\n", (double)tl / nreads, (double)ts / nreads, (double)tu / nreads);
	printf("sizeof: AlignWS %zu Frame %zu h2g_ghit %zu MateArr %zu GoVars %zu MateWS %zu\n", sizeof(AlignWS), sizeof(Frame), sizeof(h2g_ghit), sizeof(MateArr), sizeof(GoVars), sizeof(MateWS));
}
}
