// h2g_go_kernels.h — the go() kernel (HI_Aligner::go per read / per pair) around the micro-op machine of h2g_machine.h.
//
// Reads in flight are SLOTS, not lanes.  A workgroup (H2G_GO_THREADS threads) owns H2G_GO_SLOTS slots of per-read state in HBM
// (AlignWS + the machine's registers + the packed read) and keeps, in LDS, one ring of slot ids per primitive of the machine
// plus a ring of free slots.  Each wavefront loops: pick the primitive with the longest ring, pop up to 64 slots that all
// wait for THAT primitive, run it for all of them at one code site (64 of 64 lanes busy on the same latency-bound loop),
// let each lane run its read's control flow up to the next primitive request (mach_step), and push the slots into the
// rings of what they asked for.  Free slots are refilled from the batch through one global counter.  No lane ever waits
// for an unrelated lane's primitive; nothing crosses a workgroup, so LDS atomics and workgroup-scope fences are all the
// synchronisation there is.
//
// Compiled once per translation unit with that unit's capacities (-DAL_MAX_*): h2g_k_go_linear.hip, h2g_k_go_graph.hip and
// their *_big.hip siblings (large workspaces: the second pass over reads whose lists overflowed, and option sets beyond the
// default capacities).  h2g_kernels.hip sees only GoArgs and the extern "C" launchers of H2G_GO_UNIT.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include "h2g_core.h"
#include "h2g_align.h"
#include "h2g_go_args.h"

using namespace h2g;

#ifndef H2G_GO_THREADS
#define H2G_GO_THREADS 512
#endif
#ifndef H2G_GO_SLOTS
#define H2G_GO_SLOTS 1024        // reads in flight per workgroup (power of two)
#endif
#define H2G_GO_NQ ((int)SITE_COUNT)   // one ring per request site of the machine (h2g_machine.h H2G_MACH_SITES); ring 0 holds the free slots
#define H2G_PK_LANE_WORDS (H2G_PK_WORDS + H2G_PK_WORDS / 2)
#define H2G_RING_EMPTY 0xffffu

// per-read state of a slot beside its AlignWS: the machine registers between two primitives, and the packed read(s)
struct GoSlot {
	Lane     L;
	uint32_t read;
	uint32_t ro[2], rl[2];           // offset / length of the read in each read set
	uint32_t pk_ok[2];
	uint32_t pk[2][H2G_PK_LANE_WORDS];
};

// packs read i of `rd`: H2G_PK_WORDS 2-bit words then H2G_PK_WORDS/2 N-mask words; false = longer than the packed form holds.
// The codes are fetched four at a time (one unaligned dword per four bases; the buffer is padded past its last read).
__device__ __forceinline__ bool pack_read(const DReads& rd, uint32_t i, uint32_t* pk) {
	const uint32_t ro = rd.offs[i], rl = rd.offs[i + 1] - ro;
	for(uint32_t w = 0; w < H2G_PK_LANE_WORDS; w++) pk[w] = 0;
	if(rl > H2G_PK_MAXLEN) return false;
	const uint8_t* src = rd.codes + ro;
	for(uint32_t w = 0; w < (rl + 15) / 16; w++) {
		uint32_t bits = 0, mask = 0;
#pragma unroll
		for(uint32_t q = 0; q < 4; q++) {
			const uint32_t at = w * 16 + q * 4;
			if(at >= rl) break;
			uint32_t four;
			__builtin_memcpy(&four, src + at, 4);
			const uint32_t left = rl - at;
			if(left < 4) four &= (1u << (8 * left)) - 1u;             // bases of the next read: not ours
			// each byte is a code 0..4: two low bits to the 2-bit word, bit 2 (N) to the mask
			const uint32_t lo = four & 0x03030303u, n = (four >> 2) & 0x01010101u;
			const uint32_t b = (lo | (lo >> 6) | (lo >> 12) | (lo >> 18)) & 0xffu;
			const uint32_t m = (n | (n >> 7) | (n >> 14) | (n >> 21)) & 0xfu;
			bits |= b << (8 * q); mask |= m << (4 * q);
		}
		pk[w] = bits;
		pk[H2G_PK_WORDS + (w >> 1)] |= (w & 1) ? (mask << 16) : mask;
	}
	return true;
}

static_assert(H2G_GO_NQ <= 64, "the ring census is one lane per ring");
struct GoLds {
	uint32_t head[H2G_GO_NQ], tail[H2G_GO_NQ];
	uint16_t ring[H2G_GO_NQ][H2G_GO_SLOTS];
};

// pushes this lane's slot (if `valid`) into ring q; lanes of the wave may push to different rings
__device__ __forceinline__ void ring_push(GoLds* Q, bool valid, uint32_t q, uint32_t slot, int lane) {
	unsigned long long todo = __ballot(valid);
	while(todo) {                                                  // one aggregated reservation per ring present in the wave
		const int first = __ffsll((long long)todo) - 1;
		const uint32_t k = (uint32_t)__shfl((int)q, first);
		const unsigned long long m = __ballot(valid && q == k);
		uint32_t base = 0;
		if(lane == first) base = atomicAdd(&Q->tail[k], (uint32_t)__popcll(m));
		base = (uint32_t)__shfl((int)base, first);
		if(valid && q == k) {
			const uint32_t pos = (base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & (H2G_GO_SLOTS - 1);
			__atomic_store_n(&Q->ring[k][pos], (uint16_t)slot, __ATOMIC_RELAXED);
		}
		todo &= ~m;
	}
}

// pops up to 64 slots of ring q for the wave: returns the count (wave-uniform); lane i < count gets its slot
__device__ __forceinline__ uint32_t ring_pop(GoLds* Q, uint32_t q, int lane, uint32_t* slot) {
	uint32_t n = 0, h = 0;
	if(lane == 0) {
		for(;;) {
			h = __atomic_load_n(&Q->head[q], __ATOMIC_RELAXED);
			const uint32_t t = __atomic_load_n(&Q->tail[q], __ATOMIC_RELAXED);
			n = t - h;
			if(n == 0) break;
			if(n > 64) n = 64;
			if(atomicCAS(&Q->head[q], h, h + n) == h) break;
		}
	}
	n = (uint32_t)__shfl((int)n, 0); h = (uint32_t)__shfl((int)h, 0);
	if((uint32_t)lane < n) {
		const uint32_t pos = (h + (uint32_t)lane) & (H2G_GO_SLOTS - 1);
		uint16_t v;
		while((v = __atomic_load_n(&Q->ring[q][pos], __ATOMIC_RELAXED)) == H2G_RING_EMPTY) __builtin_amdgcn_s_sleep(1);   // reserved, being written
		__atomic_store_n(&Q->ring[q][pos], (uint16_t)H2G_RING_EMPTY, __ATOMIC_RELAXED);
		*slot = v;
	}
	return n;
}

// as ring_pop, for the lanes [at, at + room) of the wave: tops a thin trip up from another ring of the same primitive.  Returns the count popped.
__device__ __forceinline__ uint32_t ring_pop_at(GoLds* Q, uint32_t q, int lane, uint32_t at, uint32_t room, uint32_t* slot) {
	uint32_t n = 0, h = 0;
	if(lane == 0) {
		for(;;) {
			h = __atomic_load_n(&Q->head[q], __ATOMIC_RELAXED);
			const uint32_t t = __atomic_load_n(&Q->tail[q], __ATOMIC_RELAXED);
			n = t - h;
			if(n == 0) break;
			if(n > room) n = room;
			if(atomicCAS(&Q->head[q], h, h + n) == h) break;
		}
	}
	n = (uint32_t)__shfl((int)n, 0); h = (uint32_t)__shfl((int)h, 0);
	if((uint32_t)lane >= at && (uint32_t)lane < at + n) {
		const uint32_t pos = (h + (uint32_t)lane - at) & (H2G_GO_SLOTS - 1);
		uint16_t v;
		while((v = __atomic_load_n(&Q->ring[q][pos], __ATOMIC_RELAXED)) == H2G_RING_EMPTY) __builtin_amdgcn_s_sleep(1);
		__atomic_store_n(&Q->ring[q][pos], (uint16_t)H2G_RING_EMPTY, __ATOMIC_RELAXED);
		*slot = v;
	}
	return n;
}
// the rings (request sites) that wait for primitive `op`, as a bit set over ring ids
__device__ __forceinline__ unsigned long long mach_rings_of_op(uint32_t op) {
	unsigned long long m = 0;
#define X(OPC, PC) if(op == OPC) m |= 1ull << SITE_##PC;
	H2G_MACH_SITES(X)
#undef X
	return m;
}
#ifndef H2G_GO_TOPUP
#define H2G_GO_TOPUP 40      // a trip that popped fewer slots than this from the longest ring is topped up from the other rings of the same primitive
#endif

// UNIT tells the builds of different translation units (capacities) apart: same template arguments would be ONE symbol
template <bool GRAPH, int WAVES_PER_SIMD, int UNIT>
__global__ __launch_bounds__(H2G_GO_THREADS, WAVES_PER_SIMD) void k_go(GoArgs A)
{
	extern __shared__ uint32_t s_mem[];
	GoLds* Q = reinterpret_cast<GoLds*>(s_mem);
	uint32_t* s_pk = s_mem + (sizeof(GoLds) + 3) / 4;      // [mates][H2G_PK_LANE_WORDS][H2G_GO_THREADS], lane-interleaved
	const int lane = (int)(threadIdx.x & 63);
	const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	const bool paired = A.paired != 0;
	for(uint32_t k = threadIdx.x; k < (uint32_t)H2G_GO_NQ * H2G_GO_SLOTS; k += blockDim.x) (&Q->ring[0][0])[k] = H2G_RING_EMPTY;
	if(threadIdx.x < (uint32_t)H2G_GO_NQ) { Q->head[threadIdx.x] = 0; Q->tail[threadIdx.x] = 0; }
	__syncthreads();
	for(uint32_t k = threadIdx.x; k < H2G_GO_SLOTS; k += blockDim.x) Q->ring[0][k] = (uint16_t)k;   // every slot starts free
	if(threadIdx.x == 0) Q->tail[0] = H2G_GO_SLOTS;
	__syncthreads();
	AlnCtx C; C.g = &A.g; C.ref = &A.ref; C.ls = &A.ls; C.P = &A.P;
	C.sw = A.sw_base ? A.sw_base + tid * A.sw_stride : nullptr;
	// combineWith temp_scores: one block per wave, lane-interleaved
	C.sc = (int64_t*)(A.sc_base + (tid >> 6) * (size_t)(64 * 2 * H2G_COMBINE_MAXLEN * sizeof(int64_t))) + lane; C.sc_stride = 64;
	C.ssdb = &A.ssdb; C.rdid_base = A.rdid_base;
#if H2G_EXT_OPTS
	ctx_ext_opts(C, A.P);
#endif
	C.alts = &A.alts; C.gws = A.gws_base ? (GraphWS*)(A.gws_base + tid * A.gws_stride) : nullptr; C.graph = GRAPH;
	const size_t slot0 = (size_t)blockIdx.x * H2G_GO_SLOTS;
	Mach M;
	M.rd[0] = A.rd1; M.rd[1] = paired ? A.rd2 : A.rd1;
	uint32_t* const my_pk0 = s_pk + threadIdx.x;
	uint32_t* const my_pk1 = s_pk + H2G_PK_LANE_WORDS * H2G_GO_THREADS + threadIdx.x;
	M.rd[0].pk = my_pk0; M.rd[0].pk_stride = H2G_GO_THREADS;
	M.rd[1].pk = my_pk1; M.rd[1].pk_stride = H2G_GO_THREADS;
	M.name[0] = M.name[1] = nullptr; M.namelen[0] = M.namelen[1] = 0; M.read = 0;
	M.ws = nullptr;
	M.out = &A.O; M.paired_input = paired;
	const uint32_t total = A.list ? *A.nlist : A.rd1.n;
	unsigned long long nrank = 0, nsteps = 0, naln = 0, novf = 0, nside = 0;
	bool more = true;                                         // reads left in the batch (wave-local view)
#ifdef H2G_GO_PROF
	// wave-level time split (shader clock ticks): [0] choose + pop + load [1] control + push [2] (unused) [3+op] each primitive;
	// [20+op] slots executed; [32+op] executions; [47] trips
	unsigned long long prof[48], prof_ctl[16];
	uint32_t trip_op = 0;
	for(int k = 0; k < 48; k++) prof[k] = 0;
	for(int k = 0; k < 16; k++) prof_ctl[k] = 0;
	unsigned long long tp0 = __builtin_readcyclecounter(), tp1;
#define PROF(SLOT) do { tp1 = __builtin_readcyclecounter(); prof[SLOT] += tp1 - tp0; tp0 = tp1; } while(0)
#else
#define PROF(SLOT) do {} while(0)
#endif
	for(;;) {
		// ---- choose: the primitive with the longest ring; free slots are refilled when reads remain and nothing is long
		uint32_t cnt = 0;
		if(lane < H2G_GO_NQ) cnt = __atomic_load_n(&Q->tail[lane], __ATOMIC_RELAXED) - __atomic_load_n(&Q->head[lane], __ATOMIC_RELAXED);
		const uint32_t nfree = (uint32_t)__shfl((int)cnt, 0);
		uint32_t bestc = (lane >= 1 && lane < H2G_GO_NQ) ? cnt : 0, bestq = (uint32_t)lane;
		for(int o = 32; o > 0; o >>= 1) {
			const uint32_t oc = (uint32_t)__shfl_xor((int)bestc, o), oq = (uint32_t)__shfl_xor((int)bestq, o);
			if(oc > bestc || (oc == bestc && oq < bestq)) { bestc = oc; bestq = oq; }
		}
		bestc = (uint32_t)__shfl((int)bestc, 0); bestq = (uint32_t)__shfl((int)bestq, 0);
		const bool fetch = more && nfree > 0 && (bestc < 64 || nfree >= H2G_GO_SLOTS / 4);
		bool have = false;          // this lane carries a slot in this trip
		uint32_t slot = 0;
		GoSlot* gs = nullptr;
		if(fetch) {
			// ---- new reads into free slots
			const uint32_t n = ring_pop(Q, 0, lane, &slot);
			if(n == 0) continue;
#ifdef H2G_GO_PROF
			trip_op = 0;
#endif
			uint32_t base = 0;
			if(lane == 0) base = atomicAdd(A.work, n);
			base = (uint32_t)__shfl((int)base, 0);
			if(base + n >= total) more = false;
			const bool got = (uint32_t)lane < n && base + (uint32_t)lane < total;
			ring_push(Q, (uint32_t)lane < n && !got, 0, slot, lane);      // slots without a read go back
			if(got) {
				have = true;
				const uint32_t i = A.list ? A.list[base + (uint32_t)lane] : base + (uint32_t)lane;
				M.ws = (AlignWS*)(A.pool + (slot0 + slot) * A.ws_stride);
				gs = (GoSlot*)((uint8_t*)M.ws + A.slot_off);
				C.gsl = GRAPH ? (GraphSlot*)((uint8_t*)M.ws + A.gsl_off) : nullptr;
				gs->read = i;
				gs->pk_ok[0] = pack_read(A.rd1, i, gs->pk[0]) ? 1u : 0u;
				for(uint32_t w = 0; w < H2G_PK_LANE_WORDS; w++) my_pk0[w * H2G_GO_THREADS] = gs->pk[0][w];
				M.rd[0].pk_read = gs->pk_ok[0] ? i : 0xffffffffu;
				M.name[0] = A.names1 + A.noffs1[i]; M.namelen[0] = A.noffs1[i + 1] - A.noffs1[i];
				if(paired) {
					gs->pk_ok[1] = pack_read(A.rd2, i, gs->pk[1]) ? 1u : 0u;
					for(uint32_t w = 0; w < H2G_PK_LANE_WORDS; w++) my_pk1[w * H2G_GO_THREADS] = gs->pk[1][w];
					M.rd[1].pk_read = gs->pk_ok[1] ? i : 0xffffffffu;
					M.name[1] = A.names2 + A.noffs2[i]; M.namelen[1] = A.noffs2[i + 1] - A.noffs2[i];
				}
				mach_begin(M, i, paired);
				gs->ro[0] = M.ro[0]; gs->ro[1] = M.ro[1]; gs->rl[0] = M.rl[0]; gs->rl[1] = M.rl[1];
			}
			PROF(0);
		} else {
			if(bestc == 0) {
				// nothing queued in this workgroup: done when the batch is exhausted and every slot is free again
				if(!more && nfree == H2G_GO_SLOTS) break;
				__builtin_amdgcn_s_sleep(8);
				if(more) { uint32_t w = 0; if(lane == 0) w = __atomic_load_n(A.work, __ATOMIC_RELAXED); if((uint32_t)__shfl((int)w, 0) >= total) more = false; }
				continue;
			}
			const uint32_t op = mach_site_op(bestq);      // the ring is a request site: one primitive, one resume pc
			uint32_t n = ring_pop(Q, bestq, lane, &slot);
			if(n == 0) continue;
			// A thin trip (the machine behind a fast pass works on the hard reads only: a few hundred slots over ~34 rings) is topped up from the
			// other request sites of the SAME primitive: the primitive still runs at one code site for every lane, the control flow behind it
			// resumes at a few pcs instead of one.  Measured on the repeat-structured leg: profiles/r05_NOTES.md.
			if(n < H2G_GO_TOPUP) {
				unsigned long long cand = __ballot(cnt > 0 && lane >= 1 && lane < H2G_GO_NQ) & mach_rings_of_op(op) & ~(1ull << bestq);
				while(cand && n < 64) {
					const uint32_t q2 = (uint32_t)__ffsll((long long)cand) - 1u;
					cand &= cand - 1ull;
					n += ring_pop_at(Q, q2, lane, n, 64u - n, &slot);
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			have = (uint32_t)lane < n;
			if(have) {
				M.ws = (AlignWS*)(A.pool + (slot0 + slot) * A.ws_stride);
				gs = (GoSlot*)((uint8_t*)M.ws + A.slot_off);
				C.gsl = GRAPH ? (GraphSlot*)((uint8_t*)M.ws + A.gsl_off) : nullptr;
				M.L = gs->L;
				M.read = gs->read;
				M.ro[0] = gs->ro[0]; M.ro[1] = gs->ro[1]; M.rl[0] = gs->rl[0]; M.rl[1] = gs->rl[1];
				for(uint32_t w = 0; w < H2G_PK_LANE_WORDS; w++) my_pk0[w * H2G_GO_THREADS] = gs->pk[0][w];
				M.rd[0].pk_read = gs->pk_ok[0] ? M.read : 0xffffffffu;
				if(paired) {
					for(uint32_t w = 0; w < H2G_PK_LANE_WORDS; w++) my_pk1[w * H2G_GO_THREADS] = gs->pk[1][w];
					M.rd[1].pk_read = gs->pk_ok[1] ? M.read : 0xffffffffu;
				}
			}
#ifdef H2G_GO_PROF
			prof[20 + op] += n; prof[32 + op]++; prof[47]++; trip_op = op;
#endif
			PROF(0);
			if(have) mach_exec(C, M, op);          // ONE primitive, one code site, every lane that carries a slot
			PROF(3 + op);
		}
		// ---- control flow of each read up to its next primitive request; then hand the slots on
		uint32_t nextq = 0;
		if(have) {
			if(M.L.pc != PC_FINISHED) mach_step(C, M);
			if(M.L.pc == PC_FINISHED && M.L.op == OP_NONE) {       // OP_FINISH ran: account, the slot is free
				nrank += M.ws->nrank; nsteps += M.ws->nsteps; nside += M.ws->nside;
				naln += (M.L.a0 != 0) && !(A.defer_overflow && M.L.a1 != 0); novf += M.L.a1 != 0;
				nextq = 0;
			} else {
				nextq = mach_site_of(M.L.pc);
				gs->L = M.L;
				if(A.dbg_buf && M.read == A.dbg_read) {
					const uint32_t at = atomicAdd(A.dbg_buf, 8u);
					if(at + 9 < (1u << 20)) { uint32_t* d = A.dbg_buf + 1 + at; d[0] = M.L.pc; d[1] = M.L.op; d[2] = M.L.a0; d[3] = M.L.a1; d[4] = M.L.a2; d[5] = M.L.a3; d[6] = M.L.a4; d[7] = M.L.a5; }
				}
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		ring_push(Q, have, nextq, slot, lane);
#ifdef H2G_GO_PROF
		tp1 = __builtin_readcyclecounter(); prof[1] += tp1 - tp0; prof_ctl[trip_op] += tp1 - tp0; tp0 = tp1;
#endif
	}
#ifdef H2G_GO_PROF
	if(lane == 0) for(int k = 0; k < 48; k++) if(prof[k]) atomicAdd(A.counters + 16 + k, prof[k]);
	if(lane == 0) for(int k = 0; k < 16; k++) if(prof_ctl[k]) atomicAdd(A.counters + 80 + k, prof_ctl[k]);
#endif
	wave_add(A.counters + 0, nrank);
	wave_add(A.counters + 1, nside);
	wave_add(A.counters + 2, nsteps);
	wave_add(A.counters + 4, naln);
	wave_add(A.counters + 5, novf);
}

// the extern "C" face of one translation unit (declared in h2g_go_args.h)
#define H2G_GO_ALIGN256(X) (((X) + 255) & ~(size_t)255)
#define H2G_GO_UNIT(NAME, GRAPH, WAVES, UNIT) \
	extern "C" size_t h2g_go_ws_bytes_##NAME() { return H2G_GO_ALIGN256(sizeof(AlignWS)) + H2G_GO_ALIGN256(sizeof(GoSlot)) + ((GRAPH) ? H2G_GO_ALIGN256(sizeof(GraphSlot)) : 0); } \
	extern "C" size_t h2g_go_slot_off_##NAME() { return H2G_GO_ALIGN256(sizeof(AlignWS)); } \
	extern "C" size_t h2g_go_gsl_off_##NAME() { return H2G_GO_ALIGN256(sizeof(AlignWS)) + H2G_GO_ALIGN256(sizeof(GoSlot)); } \
	extern "C" size_t h2g_go_gws_bytes_##NAME() { return (GRAPH) ? H2G_GO_ALIGN256(sizeof(GraphWS)) : 0; } \
	extern "C" size_t h2g_go_sw_bytes_##NAME(uint32_t maxlen, int wide) { return sw_scratch_bytes(maxlen, wide != 0); }   /* SwLaneState holds H2G_GHIT_EDITS of THIS unit */ \
	extern "C" int h2g_go_waves_##NAME() { return (WAVES); } \
	extern "C" void h2g_go_geometry_##NAME(uint32_t* g) { g[0] = H2G_GO_THREADS; g[1] = H2G_GO_SLOTS; \
		g[2] = (uint32_t)((sizeof(GoLds) + 3) / 4 * 4); g[3] = H2G_PK_LANE_WORDS * H2G_GO_THREADS * 4u; /* LDS: rings + one pack region per mate */ } \
	extern "C" void h2g_go_caps_##NAME(uint32_t* c) { c[0] = AL_MAX_GHITS; c[1] = AL_MAX_RESULTS; c[2] = AL_MAX_SEARCHED; c[3] = AL_MAX_DEPTH; c[4] = AL_MAX_PARTIAL; } \
	extern "C" int h2g_go_launch_##NAME(const GoArgs* a, unsigned grid, hipStream_t st) { \
		const unsigned lds = (unsigned)((sizeof(GoLds) + 3) / 4 * 4) + (a->paired ? 2u : 1u) * H2G_PK_LANE_WORDS * H2G_GO_THREADS * 4u; \
		static std::atomic<unsigned long long> lds_ok{0};   /* more than 64 KB of dynamic LDS is an opt-in, per device: a mask over device ids */ \
		int dev_ = 0; (void)hipGetDevice(&dev_); const unsigned long long bit_ = 1ull << (dev_ & 63); \
		if(!(lds_ok.load(std::memory_order_relaxed) & bit_)) { if(hipFuncSetAttribute((const void*)k_go<GRAPH, WAVES, UNIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError(); lds_ok.fetch_or(bit_, std::memory_order_relaxed); } \
		hipLaunchKernelGGL((k_go<GRAPH, WAVES, UNIT>), dim3(grid), dim3(H2G_GO_THREADS), lds, st, *a); \
		return (int)hipGetLastError(); }
