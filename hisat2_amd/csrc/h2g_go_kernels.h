// h2g_go_kernels.h — the go() kernel (HI_Aligner::go per read / per pair) around the micro-op machine of h2g_machine.h.
//
// One lane = one read (pair) in flight; a lane that finishes takes the next read of the batch at once (wave-aggregated
// atomic), so the 64 lanes of a wavefront are always populated.  Each trip of the kernel's loop every lane first runs its
// own control flow up to its next primitive request (mach_step), then the wavefront votes and executes ONE primitive —
// the one with the most (oldest) requesters — at ONE code site for all of them (mach_exec with a wave-uniform op).  Lanes
// whose request lost the vote simply wait; because lanes never idle for lack of reads, waiting costs latency, not throughput.
//
// This header is compiled once per translation unit with that unit's capacities (-DAL_MAX_*): h2g_k_go_linear.hip,
// h2g_k_go_graph.hip and their *_big.hip siblings (large workspaces: the second pass over reads whose lists overflowed, and
// option sets beyond the default capacities).  h2g_kernels.hip sees only GoArgs and the extern "C" launchers of H2G_GO_UNIT.
#pragma once
#include <hip/hip_runtime.h>
#include "h2g_core.h"
#include "h2g_align.h"
#include "h2g_go_args.h"

using namespace h2g;

// packs read i of `rd` into this lane's LDS slot: H2G_PK_WORDS 2-bit words then H2G_PK_WORDS/2 N-mask words, word k of lane
// t at pk[k * 256 + t] (lane-interleaved => conflict-free ds_read_b32)
__device__ __forceinline__ void pack_read(const DReads& rd, uint32_t i, uint32_t* pk, DReads* view) {
	const uint32_t ro = rd.offs[i], rl = rd.offs[i + 1] - ro;
	view->pk_read = 0xffffffffu;
	if(rl > H2G_PK_MAXLEN) return;
	for(uint32_t w = 0; w < (rl + 15) / 16; w++) {
		uint32_t bits = 0, mask = 0;
		for(uint32_t k = 0; k < 16 && w * 16 + k < rl; k++) {
			const uint32_t c = rd.codes[ro + w * 16 + k];
			bits |= (c & 3u) << (2 * k);
			mask |= (c > 3u ? 1u : 0u) << k;
		}
		pk[w * 256 + threadIdx.x] = bits;
		uint32_t& mw = pk[(H2G_PK_WORDS + (w >> 1)) * 256 + threadIdx.x];
		mw = (w & 1) ? (mw | (mask << 16)) : mask;
	}
	view->pk_read = i;
}

#define H2G_PK_LANE_WORDS (H2G_PK_WORDS + H2G_PK_WORDS / 2)

template <bool GRAPH, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k_go(GoArgs A)
{
	extern __shared__ uint32_t s_pk[];   // [mates][H2G_PK_LANE_WORDS][256]
	const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	const int lane = (int)(threadIdx.x & 63);
	const bool paired = A.paired != 0;
	AlnCtx C; C.g = &A.g; C.ref = &A.ref; C.ls = &A.ls; C.P = &A.P;
	C.sw = A.sw_base ? A.sw_base + tid * A.sw_stride : nullptr;
	C.alts = &A.alts; C.gws = A.gws_base ? (GraphWS*)(A.gws_base + tid * A.gws_stride) : nullptr; C.graph = GRAPH;
	Mach M;
	M.ws = (AlignWS*)(A.pool + tid * A.ws_stride);
	M.L.pc = PC_IDLE; M.L.op = OP_NONE;
	M.rd[0] = A.rd1; M.rd[1] = paired ? A.rd2 : A.rd1;
	M.rd[0].pk = s_pk + threadIdx.x; M.rd[0].pk_stride = 256;
	M.rd[1].pk = s_pk + H2G_PK_LANE_WORDS * 256 + threadIdx.x; M.rd[1].pk_stride = 256;
	M.name[0] = M.name[1] = nullptr; M.namelen[0] = M.namelen[1] = 0; M.read = 0;
	const uint32_t total = A.list ? *A.nlist : A.rd1.n;
	unsigned long long nrank = 0, nsteps = 0, naln = 0, novf = 0, nside = 0;
	bool more = true;
	uint32_t age[OP_COUNT];
#pragma unroll
	for(int k = 0; k < (int)OP_COUNT; k++) age[k] = 0;
	for(;;) {
		if(M.L.pc == PC_FINISHED) {                      // the read this lane carried is done: account, free the lane
			nrank += M.ws->nrank; nsteps += M.ws->nsteps; nside += M.ws->nside;
			naln += (M.L.a0 != 0) && !(A.defer_overflow && M.L.a1 != 0); novf += M.L.a1 != 0;
			M.L.pc = PC_IDLE;
		}
		if(more) {                                       // idle lanes take the next reads of the batch
			const bool idle = M.L.pc == PC_IDLE;
			const unsigned long long need = __ballot(idle);
			if(need) {
				const int leader = __ffsll((long long)need) - 1;
				const uint32_t cnt = (uint32_t)__popcll(need);
				uint32_t base = 0;
				if(lane == leader) base = atomicAdd(A.work, cnt);
				base = (uint32_t)__shfl((int)base, leader);
				if(idle) {
					const uint32_t j = base + (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
					if(j < total) {
						const uint32_t i = A.list ? A.list[j] : j;
						pack_read(A.rd1, i, s_pk, &M.rd[0]);
						M.name[0] = A.names1 + A.noffs1[i]; M.namelen[0] = A.noffs1[i + 1] - A.noffs1[i];
						if(paired) {
							pack_read(A.rd2, i, s_pk + H2G_PK_LANE_WORDS * 256, &M.rd[1]);
							M.name[1] = A.names2 + A.noffs2[i]; M.namelen[1] = A.noffs2[i + 1] - A.noffs2[i];
						}
						mach_begin(M, i, paired);
					}
				}
				if(base + cnt >= total) more = false;
			}
		}
		if(M.L.pc != PC_IDLE && M.L.op == OP_NONE) mach_step(C, M);   // control flow up to the next primitive request
		// vote: the primitive with the most requesters, aged so that a rare request cannot starve behind common ones
		uint32_t best = 0, bestScore = 0;
#pragma unroll
		for(int k = 1; k < (int)OP_COUNT; k++) {
			const uint32_t c = (uint32_t)__popcll(__ballot(M.L.op == (uint32_t)k));
			if(c) {
				const uint32_t sc = c + age[k];
				if(sc > bestScore) { bestScore = sc; best = (uint32_t)k; }
				age[k] += 2;
			} else age[k] = 0;
		}
		if(best == 0) {
			if(!more && __ballot(M.L.pc != PC_IDLE) == 0ull) break;
			continue;
		}
#pragma unroll
		for(int k = 1; k < (int)OP_COUNT; k++) if((uint32_t)k == best) age[k] = 0;
		best = (uint32_t)__builtin_amdgcn_readfirstlane((int)best);
		if(M.L.op == best) mach_exec(C, M, best, A.O, paired);
	}
	wave_add(A.counters + 0, nrank);
	wave_add(A.counters + 1, nside);
	wave_add(A.counters + 2, nsteps);
	wave_add(A.counters + 4, naln);
	wave_add(A.counters + 5, novf);
}

// the extern "C" face of one translation unit (declared in h2g_go_args.h)
#define H2G_GO_UNIT(NAME, GRAPH, WAVES) \
	extern "C" size_t h2g_go_ws_bytes_##NAME() { return (sizeof(AlignWS) + 255) & ~(size_t)255; } \
	extern "C" size_t h2g_go_gws_bytes_##NAME() { return (GRAPH) ? ((sizeof(GraphWS) + 255) & ~(size_t)255) : 0; } \
	extern "C" int h2g_go_waves_##NAME() { return (WAVES); } \
	extern "C" void h2g_go_caps_##NAME(uint32_t* c) { c[0] = AL_MAX_GHITS; c[1] = AL_MAX_RESULTS; c[2] = AL_MAX_SEARCHED; c[3] = AL_MAX_DEPTH; c[4] = AL_MAX_PARTIAL; } \
	extern "C" int h2g_go_launch_##NAME(const GoArgs* a, unsigned grid, hipStream_t st) { \
		const unsigned lds = (a->paired ? 2u : 1u) * H2G_PK_LANE_WORDS * 256u * 4u; \
		hipLaunchKernelGGL((k_go<GRAPH, WAVES>), dim3(grid), dim3(256), lds, st, *a); \
		return (int)hipGetLastError(); }
