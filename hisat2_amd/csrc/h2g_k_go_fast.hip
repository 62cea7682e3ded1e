// go() fast pass: HI_Aligner::go of the dominant traces with the per-read state in registers + LDS (h2g_fast.h).
//
// One read / pair per LANE, persistent lanes: a lane that completes (or bails) takes the next read of the batch.  Every round
// each lane runs its control flow (registers + LDS words) up to its next primitive request; the wavefront then runs ONE primitive
// — the one most lanes ask for, with ageing so that rare ones are not starved — at a single code site for all its requesters.
// The only HBM traffic of a read is its index lines (the algorithmic bytes), its bases and its result records.
#include "h2g_go_args.h"

using namespace h2g;

#ifndef H2G_FAST_THREADS
#define H2G_FAST_THREADS 512
#endif
#define H2G_FAST_LDS_WORDS (FW_HOT + 2 * H2G_PK_WORDS)

__global__ __launch_bounds__(H2G_FAST_THREADS) void k_go_fast(FastArgs A)
{
	extern __shared__ uint32_t s_mem[];
	const int lane = (int)(threadIdx.x & 63);
	const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	const bool paired = A.paired != 0;
	uint32_t cold[FW_COLD];
	FWords W; W.hot = (FG_LDS uint32_t*)(s_mem + threadIdx.x); W.hot_stride = H2G_FAST_THREADS; W.cold = (FG_PRIV uint32_t*)cold;
	uint32_t* const pk0 = s_mem + FW_HOT * H2G_FAST_THREADS + threadIdx.x;
	uint32_t* const pk1 = pk0 + H2G_PK_WORDS * H2G_FAST_THREADS;
	FCtx C;
	C.g = &A.g; C.ref = &A.ref; C.ls = &A.ls; C.P = &A.P;
	C.rd[0] = A.rd1; C.rd[1] = paired ? A.rd2 : A.rd1;
	C.pk[0] = pk0; C.pk[1] = paired ? pk1 : pk0; C.pk_stride = H2G_FAST_THREADS;
	C.sc = (int64_t*)(A.sc_base + (tid >> 6) * (size_t)(64 * 2 * H2G_COMBINE_MAXLEN * sizeof(int64_t))) + lane; C.sc_stride = 64;
	C.O = A.O;
	C.name[0] = C.name[1] = nullptr; C.namelen[0] = C.namelen[1] = 0;
	FState S;
	S.pc = FPC_DONE; S.op = FOP_NONE; S.bail = FB_NONE; S.read = 0;
	unsigned long long nrank = 0, nside = 0, nsteps = 0, naln = 0, ndone = 0, nbail = 0;
	const uint32_t FOP_FETCH = FOP_COUNT;                      // idle lanes "request" the next reads: refills compete (and age) like any primitive
	uint32_t age[FOP_COUNT + 1];
	for(int k = 0; k <= (int)FOP_COUNT; k++) age[k] = 0;
	bool more = true;
	const uint32_t total = A.total;
#ifdef H2G_GO_PROF
	// wave-level time split (shader clock): [0] fetch [1] control [2] vote [3+op] each primitive; [20+op] lanes executed; [32+op] executions;
	// [40] lanes stepped [41] step calls [47] rounds
	unsigned long long prof[48];
	for(int k = 0; k < 48; k++) prof[k] = 0;
	unsigned long long tp0 = __builtin_readcyclecounter(), tp1;
#define PROF(SLOT) do { tp1 = __builtin_readcyclecounter(); prof[SLOT] += tp1 - tp0; tp0 = tp1; } while(0)
#else
#define PROF(SLOT) do {} while(0)
#endif
	for(;;) {
		// ---- control up to the next primitive request
		{
			const bool run = S.pc != FPC_DONE && S.pc != FPC_BAIL && S.op == FOP_NONE;
#ifdef H2G_GO_PROF
			{ const unsigned long long rm = __ballot(run); if(rm) { prof[40] += __popcll(rm); prof[41]++; } prof[47]++; }
#endif
			if(run) {
				fast_step(C, S, W);
				if(S.pc == FPC_DONE) { nrank += S.nrank; nside += S.nside; nsteps += S.nsteps; naln += S.a0 != 0; ndone++; }
			}
		}
		PROF(1);
		// ---- one primitive (or the refill) for the wave: most requesters, aged
		const bool idle = S.pc == FPC_DONE || S.pc == FPC_BAIL;
		const uint32_t want = idle ? (more ? FOP_FETCH : (uint32_t)FOP_NONE) : S.op;
		uint32_t bestop = FOP_NONE, bestscore = 0;
#pragma unroll
		for(uint32_t op = 1; op <= FOP_COUNT; op++) {
			const uint32_t c = (uint32_t)__popcll(__ballot(want == op));
			const uint32_t sc = c ? c + 6 * age[op] : 0;
			if(sc > bestscore) { bestscore = sc; bestop = op; }
			age[op] = c ? age[op] + 1 : 0;
		}
		if(bestop == FOP_NONE) break;                            // nothing in flight, nothing left to fetch
#pragma unroll
		for(uint32_t op = 1; op <= FOP_COUNT; op++) if(op == bestop) age[op] = 0;
		PROF(2);
		if(bestop == FOP_FETCH) {
			// reads that left the fast path: their ids go to the general machine's list
			const bool b = S.pc == FPC_BAIL && S.bail != FB_NONE;
			const unsigned long long bm = __ballot(b);
			if(bm) {
				uint32_t base = 0;
				if(lane == 0) base = atomicAdd(A.bail_count, (uint32_t)__popcll(bm));
				base = (uint32_t)__shfl((int)base, 0);
				if(b) {
					A.bail_list[base + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull))] = S.read;
					atomicAdd(A.counters + 96 + (S.bail < FB_COUNT ? S.bail : (uint32_t)FB_OTHER), 1ull);
					nbail++; S.bail = FB_NONE;
				}
			}
			const unsigned long long im = __ballot(idle);
			const uint32_t n = (uint32_t)__popcll(im);
			uint32_t base = 0;
			if(lane == 0) base = atomicAdd(A.work, n);
			base = (uint32_t)__shfl((int)base, 0);
			if(base + n >= total) more = false;
			const uint32_t mine = base + (uint32_t)__popcll(im & ((1ull << lane) - 1ull));
			if(idle && mine < total) {
				bool ok = fg_pack_read(A.rd1, mine, pk0, H2G_FAST_THREADS);
				C.name[0] = A.names1 + A.noffs1[mine]; C.namelen[0] = A.noffs1[mine + 1] - A.noffs1[mine];
				if(paired) {
					ok = fg_pack_read(A.rd2, mine, pk1, H2G_FAST_THREADS) && ok;
					C.name[1] = A.names2 + A.noffs2[mine]; C.namelen[1] = A.noffs2[mine + 1] - A.noffs2[mine];
				}
				fast_begin(C, S, mine, paired, ok);
			}
			PROF(0);
		} else {
#ifdef H2G_GO_PROF
			prof[20 + bestop] += __popcll(__ballot(S.op == bestop)); prof[32 + bestop]++;
#endif
			if(S.op == bestop) fast_exec(C, S, W, bestop);
			PROF(3 + bestop);
		}
	}
	// bails not yet handed on (the batch ran out before another refill)
	{
		const bool b = S.pc == FPC_BAIL && S.bail != FB_NONE;
		const unsigned long long bm = __ballot(b);
		if(bm) {
			uint32_t base = 0;
			if(lane == 0) base = atomicAdd(A.bail_count, (uint32_t)__popcll(bm));
			base = (uint32_t)__shfl((int)base, 0);
			if(b) {
				A.bail_list[base + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull))] = S.read;
				atomicAdd(A.counters + 96 + (S.bail < FB_COUNT ? S.bail : (uint32_t)FB_OTHER), 1ull);
				nbail++;
			}
		}
	}
#ifdef H2G_GO_PROF
	if(lane == 0) for(int k = 0; k < 48; k++) if(prof[k]) atomicAdd(A.counters + 128 + k, prof[k]);
#endif
	wave_add(A.counters + 0, nrank);
	wave_add(A.counters + 1, nside);
	wave_add(A.counters + 2, nsteps);
	wave_add(A.counters + 4, naln);
	wave_add(A.counters + 6, ndone);
	wave_add(A.counters + 7, nbail);
}

extern "C" void h2g_go_fast_geometry(uint32_t* g) { g[0] = H2G_FAST_THREADS; g[1] = H2G_FAST_LDS_WORDS * H2G_FAST_THREADS * 4u; }
extern "C" int h2g_go_fast_launch(const FastArgs* a, unsigned grid, hipStream_t st) {
	const unsigned lds = H2G_FAST_LDS_WORDS * H2G_FAST_THREADS * 4u;
	static bool lds_ok = false;   // more than 64 KB of dynamic LDS is an opt-in
	if(!lds_ok) { if(hipFuncSetAttribute((const void*)k_go_fast, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError(); lds_ok = true; }
	hipLaunchKernelGGL(k_go_fast, dim3(grid), dim3(H2G_FAST_THREADS), lds, st, *a);
	return (int)hipGetLastError();
}
