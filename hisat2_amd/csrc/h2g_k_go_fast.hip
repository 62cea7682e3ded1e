// go() fast pass: HI_Aligner::go of the dominant traces with a COMPACT per-read state (h2g_fast.h).
//
// Reads in flight are slots of FG_SLOT_WORDS words (about 1 KB: 40 state words, 50 hot words, the packed reads, the cold words), contiguous
// in HBM.  A workgroup owns H2G_FAST_SLOTS of them and keeps, in LDS, one queue of slot ids per request site of the fast machine
// (primitive + resume pc) plus the free queue.  Each wave loops: pop up to 64 slots of the longest queue, load their state in one go
// (registers + a per-lane LDS staging area: 27 16-byte loads per lane, no dependent chain), run THAT primitive for all of them at one
// code site, let every lane run its read's control flow on registers / LDS up to the next request, store the state and push the slot to
// the queue of what it asked for.  Reads that leave the fast path go to the general machine's list.
// h2g_k_go_fast_graph.hip compiles this file once more with FG_GRAPH = 1 (graph indexes: h2g_fast.h) under its own symbol names.
#include <atomic>
#include "h2g_go_args.h"

using namespace h2g;

#ifndef FG_KERNEL
#define FG_KERNEL   k_go_fast
#define FG_LAUNCH   h2g_go_fast_launch
#define FG_GEOMETRY h2g_go_fast_geometry
#endif

#ifndef H2G_FAST_THREADS
#define H2G_FAST_THREADS 512
#endif
#ifndef H2G_FAST_SLOTS
#define H2G_FAST_SLOTS 1024       // reads in flight per workgroup (power of two)
#endif
// The scratch of the graph primitives (GraphWS: group walk + ALT-aware extension) lives in the lane's PRIVATE segment (round 6): the hardware interleaves a
// wave's private memory dword by dword, so a wave-wide access to one field is 256 contiguous bytes.  In global memory at one GraphWS per lane the same access
// was 64 lines, 17 KB apart: 40 GB written per million pairs (profiles/r05_g_graph_pmc_traffic.json).  Round 5's attempt at the default capacities (20 KB per
// lane) was refused by the runtime; h2g_k_go_fast_graph.hip sets the capacities a read of the fast path can use (it holds four edits per hit and five
// coordinates per resolution: everything beyond is a bail as before).
#ifndef FG_GWS_PRIVATE
#define FG_GWS_PRIVATE FG_GRAPH
#endif
#define FG_STAGE_WORDS (FW_HOT + 2 * H2G_PK_WORDS)                     // staged in LDS per lane: hot words + packed reads
#define FG_SLOT_WORDS  (((FS_WORDS + FW_HOT + 2 * H2G_PK_WORDS + FW_COLD) + 3) & ~3)   // 16-byte multiple
#define FG_NQ ((int)FQ_COUNT)
#define FG_RING_EMPTY 0xffffu
static_assert(FG_NQ <= 64, "the queue census is one lane per queue");
static_assert(FS_WORDS % 4 == 0, "16-byte loads of the state");

struct FastLds {
	uint32_t head[FG_NQ], tail[FG_NQ];
	uint16_t ring[FG_NQ][H2G_FAST_SLOTS];
};

__device__ __forceinline__ void fq_push(FastLds* Q, bool valid, uint32_t q, uint32_t slot, int lane) {
	unsigned long long todo = __ballot(valid);
	while(todo) {                                                  // one aggregated reservation per queue present in the wave
		const int first = __ffsll((long long)todo) - 1;
		const uint32_t k = (uint32_t)__shfl((int)q, first);
		const unsigned long long m = __ballot(valid && q == k);
		uint32_t base = 0;
		if(lane == first) base = atomicAdd(&Q->tail[k], (uint32_t)__popcll(m));
		base = (uint32_t)__shfl((int)base, first);
		if(valid && q == k) {
			const uint32_t pos = (base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & (H2G_FAST_SLOTS - 1);
			__atomic_store_n(&Q->ring[k][pos], (uint16_t)slot, __ATOMIC_RELAXED);
		}
		todo &= ~m;
	}
}
__device__ __forceinline__ uint32_t fq_pop(FastLds* Q, uint32_t q, int lane, uint32_t* slot) {
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
		const uint32_t pos = (h + (uint32_t)lane) & (H2G_FAST_SLOTS - 1);
		uint16_t v;
		while((v = __atomic_load_n(&Q->ring[q][pos], __ATOMIC_RELAXED)) == FG_RING_EMPTY) __builtin_amdgcn_s_sleep(1);   // reserved, being written
		__atomic_store_n(&Q->ring[q][pos], (uint16_t)FG_RING_EMPTY, __ATOMIC_RELAXED);
		*slot = v;
	}
	return n;
}

__device__ __forceinline__ void fk_ctx(const FastArgs* A, uint32_t* stage, uint32_t* sm, FCtx& C, FWords& W) {
	const bool paired = A->paired != 0;
	W.hot = (FG_LDS uint32_t*)stage; W.hot_stride = H2G_FAST_THREADS; W.cold = sm + FS_WORDS + FG_STAGE_WORDS;
	uint32_t* const pk0 = stage + FW_HOT * H2G_FAST_THREADS;
	C.g = &A->g; C.ref = &A->ref; C.ls = &A->ls; C.P = &A->P;
	C.rd[0] = A->rd1; C.rd[1] = paired ? A->rd2 : A->rd1;
	C.pk[0] = pk0; C.pk[1] = paired ? pk0 + H2G_PK_WORDS * H2G_FAST_THREADS : pk0; C.pk_stride = H2G_FAST_THREADS;
	C.sc = nullptr; C.sc_stride = 0;
#if FG_GRAPH
	{
		const size_t tid = (size_t)blockIdx.x * H2G_FAST_THREADS + threadIdx.x;
		C.alts = &A->alts;
#if !FG_GWS_PRIVATE
		C.gws = reinterpret_cast<GraphWS*>(A->gws_base + tid * A->gws_stride);
#endif
		C.sc = reinterpret_cast<int64_t*>(A->sc_base + (tid >> 6) * (size_t)(64 * 2 * H2G_COMBINE_MAXLEN * sizeof(int64_t))) + (threadIdx.x & 63); C.sc_stride = 64;
	}
#endif
	C.O = A->O;
	C.name[0] = C.name[1] = nullptr; C.namelen[0] = C.namelen[1] = 0;
}
// One trip of this lane's slot with the state in registers from its load to its store: the primitive `op` (FOP_NONE: a slot taking up
// read `begin`), then the control flow up to the next request.  Returns the state's word 0 (pc, op, bail); the whole state is in the slot.
__device__ __forceinline__ uint32_t fk_trip(const FastArgs* A, uint32_t* stage, uint32_t* sm, uint32_t op, uint32_t begin, uint32_t packed_ok) {
	FCtx C; FWords W;
	fk_ctx(A, stage, sm, C, W);
#if FG_GRAPH && FG_GWS_PRIVATE
	GraphWS gws_private;
	C.gws = &gws_private;
#endif
	FState S;
	if(begin != H2G_MAX) {
		const bool paired = A->paired != 0;
		C.name[0] = A->names1 + A->noffs1[begin]; C.namelen[0] = A->noffs1[begin + 1] - A->noffs1[begin];
		if(paired) { C.name[1] = A->names2 + A->noffs2[begin]; C.namelen[1] = A->noffs2[begin + 1] - A->noffs2[begin]; }
		fast_begin(C, S, begin, paired, packed_ok != 0);
	} else {
		uint32_t w[FS_WORDS];
		const uint4* src = reinterpret_cast<const uint4*>(sm);
#pragma unroll
		for(uint32_t k = 0; k < FS_WORDS / 4; k++) { const uint4 v = src[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
		__builtin_memcpy(&S, w, sizeof S);
		// the work counters are 16-bit fields: the primitive counts from zero, a read that would wrap them leaves the fast path
		const uint32_t nr0 = S.nrank, ns0 = S.nside, nt0 = S.nsteps;
		S.nrank = 0; S.nside = 0; S.nsteps = 0;
		fast_exec(C, S, W, op);
		const uint32_t nr_ = nr0 + S.nrank, ns_ = ns0 + S.nside, nt_ = nt0 + S.nsteps;
		if(S.pc != FPC_BAIL && (nr_ > 0xffffu || ns_ > 0xffffu || nt_ > 0xffffu)) { S.pc = FPC_BAIL; S.bail = FB_OTHER; }
		S.nrank = nr_ & 0xffffu; S.nside = ns_ & 0xffffu; S.nsteps = nt_ & 0xffffu;
	}
#ifdef FG_DBG_TRACE
	const uint32_t dbg_a[6] = {S.a0, S.a1, S.a2, S.a3, S.a4, S.a5};
	const uint32_t dbg_pc0 = S.pc, dbg_op0 = S.op;
#endif
	if(S.pc != FPC_DONE && S.pc != FPC_BAIL && S.op == FOP_NONE) fast_step(C, S, W);
#ifdef FG_DBG_TRACE
	if(A->dbg_buf && S.read == A->dbg_read) {
		const uint32_t at = atomicAdd(A->dbg_buf, 12u);
		if(at + 13 < (1u << 20)) {
			uint32_t* d = A->dbg_buf + 1 + at;
			d[0] = op; d[1] = dbg_pc0 | (dbg_op0 << 8); for(int k = 0; k < 6; k++) d[2 + k] = dbg_a[k]; d[8] = S.pc | (S.op << 8) | ((uint32_t)(S.sp & 15) << 16); d[9] = S.nrank | (S.nsteps << 16); d[10] = W.ld(FW_CO + 2); d[11] = S.a4;
		}
	}
#endif
	uint32_t w[FS_WORDS];
	__builtin_memcpy(w, &S, sizeof S);
	uint4* dst = reinterpret_cast<uint4*>(sm);
#pragma unroll
	for(uint32_t k = 0; k < FS_WORDS / 4; k++) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
	return w[0];
}

__global__ __launch_bounds__(H2G_FAST_THREADS) void FG_KERNEL(const FastArgs* __restrict__ A)
{
	extern __shared__ uint32_t s_mem[];
	FastLds* Q = reinterpret_cast<FastLds*>(s_mem);
	uint32_t* const stage = s_mem + (sizeof(FastLds) + 3) / 4 + threadIdx.x;      // word w of this lane at stage[w * H2G_FAST_THREADS]
	const int lane = (int)(threadIdx.x & 63);
	const bool paired = A->paired != 0;
	for(uint32_t k = threadIdx.x; k < (uint32_t)FG_NQ * H2G_FAST_SLOTS; k += blockDim.x) (&Q->ring[0][0])[k] = FG_RING_EMPTY;
	if(threadIdx.x < (uint32_t)FG_NQ) { Q->head[threadIdx.x] = 0; Q->tail[threadIdx.x] = 0; }
	__syncthreads();
	for(uint32_t k = threadIdx.x; k < H2G_FAST_SLOTS; k += blockDim.x) Q->ring[0][k] = (uint16_t)k;   // every slot starts free
	if(threadIdx.x == 0) Q->tail[0] = H2G_FAST_SLOTS;
	__syncthreads();
	uint32_t* const pk0 = stage + FW_HOT * H2G_FAST_THREADS;
	uint32_t* const pk1 = pk0 + H2G_PK_WORDS * H2G_FAST_THREADS;
	uint32_t* const slots0 = A->slots + (size_t)blockIdx.x * H2G_FAST_SLOTS * FG_SLOT_WORDS;
	unsigned long long nrank = 0, nside = 0, nsteps = 0, naln = 0, ndone = 0, nbail = 0;
	bool more = true;
	const uint32_t total = A->total, tail_n = A->tail;
#ifdef H2G_GO_PROF
	// wave-level time split (shader clock): [0] choose + pop + load [1] control [2] store [16] release fence [17] push [3+op] each primitive [15] new reads; [20+op] slots executed;
	// [32+op] executions; [46] slots stepped [47] trips
	unsigned long long prof[48], prof_ctl[32], prof_n[32];
	for(int k = 0; k < 48; k++) prof[k] = 0;
	for(int k = 0; k < 32; k++) { prof_ctl[k] = 0; prof_n[k] = 0; }
	uint32_t trip_site = 0;
	unsigned long long tp0 = __builtin_readcyclecounter(), tp1;
#define PROF(SLOT) do { tp1 = __builtin_readcyclecounter(); prof[SLOT] += tp1 - tp0; tp0 = tp1; } while(0)
#else
#define PROF(SLOT) do {} while(0)
#endif
	for(;;) {
		// ---- choose: the site with the longest queue; free slots are refilled when reads remain and nothing is long
		uint32_t cnt = 0;
		if(lane < FG_NQ) cnt = __atomic_load_n(&Q->tail[lane], __ATOMIC_RELAXED) - __atomic_load_n(&Q->head[lane], __ATOMIC_RELAXED);
		const uint32_t nfree = (uint32_t)__shfl((int)cnt, 0);
		uint32_t bestc = (lane >= 1 && lane < FG_NQ) ? cnt : 0, bestq = (uint32_t)lane;
		for(int o = 32; o > 0; o >>= 1) {
			const uint32_t oc = (uint32_t)__shfl_xor((int)bestc, o), oq = (uint32_t)__shfl_xor((int)bestq, o);
			if(oc > bestc || (oc == bestc && oq < bestq)) { bestc = oc; bestq = oq; }
		}
		bestc = (uint32_t)__shfl((int)bestc, 0); bestq = (uint32_t)__shfl((int)bestq, 0);
		const bool fetch = more && nfree > 0 && (bestc < 64 || nfree >= H2G_FAST_SLOTS / 4);
		bool have = false;
		bool tail = false;
		uint32_t slot = 0, begin = H2G_MAX, packed_ok = 0, trip_op = FOP_NONE;
		uint32_t* sm = nullptr;                                   // this lane's slot in HBM
		if(fetch) {
			const uint32_t n = fq_pop(Q, 0, lane, &slot);
			if(n == 0) continue;
			uint32_t base = 0;
			if(lane == 0) base = atomicAdd(A->work, n);
			base = (uint32_t)__shfl((int)base, 0);
			if(base + n >= total) more = false;
			const bool got = (uint32_t)lane < n && base + (uint32_t)lane < total;
			fq_push(Q, (uint32_t)lane < n && !got, 0, slot, lane);      // slots without a read go back
			if(got) {
				have = true;
				begin = base + (uint32_t)lane;
				sm = slots0 + (size_t)slot * FG_SLOT_WORDS;
				bool ok = fg_pack_read(A->rd1, begin, pk0, H2G_FAST_THREADS);
				if(paired) ok = fg_pack_read(A->rd2, begin, pk1, H2G_FAST_THREADS) && ok;
				packed_ok = ok ? 1u : 0u;
				// the packed reads are written to the slot once
#pragma unroll
				for(uint32_t k = 0; k < 2 * H2G_PK_WORDS; k++) sm[FS_WORDS + FW_HOT + k] = pk0[k * H2G_FAST_THREADS];
			}
#ifdef H2G_GO_PROF
			trip_site = 0;
#endif
			PROF(15);
		} else {
			if(bestc == 0) {
				if(!more && nfree == H2G_FAST_SLOTS) break;           // the batch is exhausted and every slot is free again
				__builtin_amdgcn_s_sleep(8);
				if(more) { uint32_t w = 0; if(lane == 0) w = __atomic_load_n(A->work, __ATOMIC_RELAXED); if((uint32_t)__shfl((int)w, 0) >= total) more = false; }
				continue;
			}
			const uint32_t op = fg_queue_op(bestq);
			const uint32_t n = fq_pop(Q, bestq, lane, &slot);
			if(n == 0) continue;
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			have = (uint32_t)lane < n;
			// (FastArgs::tail: the last reads in flight of an exhausted batch are a few latency chains per workgroup; they can go to the general
			// machine's pass, which is in flight anyway — profiles/r04_NOTES.md §1, §4)
			tail = !more && H2G_FAST_SLOTS - nfree <= tail_n;
			if(have && tail) sm = slots0 + (size_t)slot * FG_SLOT_WORDS;
			if(have && !tail) {
				sm = slots0 + (size_t)slot * FG_SLOT_WORDS;
				// hot words + packed reads into this lane's LDS staging area: 16-byte loads, nothing depends on anything
				const uint4* hsrc = reinterpret_cast<const uint4*>(sm + FS_WORDS);
#pragma unroll
				for(uint32_t k = 0; k < FG_STAGE_WORDS / 4; k++) {
					const uint4 v = hsrc[k];
					stage[(4 * k) * H2G_FAST_THREADS] = v.x; stage[(4 * k + 1) * H2G_FAST_THREADS] = v.y; stage[(4 * k + 2) * H2G_FAST_THREADS] = v.z; stage[(4 * k + 3) * H2G_FAST_THREADS] = v.w;
				}
#pragma unroll
				for(uint32_t k = FG_STAGE_WORDS & ~3u; k < FG_STAGE_WORDS; k++) stage[k * H2G_FAST_THREADS] = sm[FS_WORDS + k];
			}
#ifdef H2G_GO_PROF
			prof[20 + op] += n; prof[32 + op]++; trip_site = bestq;
#endif
			PROF(0);
			trip_op = op;
		}
		// ---- control flow of each read up to its next primitive request; then hand the slots on
		uint32_t nextq = 0;
#ifdef H2G_GO_PROF
		prof[46] += __popcll(__ballot(have)); prof[47]++;
#endif
		uint32_t w0 = 0;
		if(have && tail) w0 = (uint32_t)FPC_BAIL | ((uint32_t)FB_TAIL << 12);     // (pc, bail reason of state word 0; the read id is state word 8 in the slot)
		else if(have) w0 = fk_trip(A, stage, sm, trip_op, begin, packed_ok);
#ifdef H2G_GO_PROF
		{ const unsigned long long t_ = __builtin_readcyclecounter(); prof_ctl[trip_site & 31] += t_ - tp0; prof_n[trip_site & 31]++; }
#endif
		PROF(1);
		const uint32_t pc = w0 & 0xffu, why = (w0 >> 12) & 0x1fu;
		if(have) {
			if(pc == FPC_DONE) { const uint32_t c26 = sm[26], c27 = sm[27]; nrank += c26 & 0xffffu; nside += c26 >> 16; nsteps += c27 & 0xffffu; naln += sm[2] != 0; ndone++; }
			else if(pc != FPC_BAIL) {
				uint4* hdst = reinterpret_cast<uint4*>(sm + FS_WORDS);
#pragma unroll
				for(uint32_t k = 0; k < FW_HOT / 4; k++)
					hdst[k] = make_uint4(stage[(4 * k) * H2G_FAST_THREADS], stage[(4 * k + 1) * H2G_FAST_THREADS], stage[(4 * k + 2) * H2G_FAST_THREADS], stage[(4 * k + 3) * H2G_FAST_THREADS]);
#pragma unroll
				for(uint32_t k = FW_HOT & ~3u; k < FW_HOT; k++) sm[FS_WORDS + k] = stage[k * H2G_FAST_THREADS];
				nextq = fg_queue_of(pc);
			}
		}
		// reads that left the fast path: their ids go to the general machine's list
		{
			const bool b = have && pc == FPC_BAIL;
			const unsigned long long bm = __ballot(b);
			if(bm) {
				uint32_t base = 0;
				if(lane == 0) base = atomicAdd(A->bail_count, (uint32_t)__popcll(bm));
				base = (uint32_t)__shfl((int)base, 0);
				if(b) {
					A->bail_list[base + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull))] = sm[8];       // state word 8 = the read id
					atomicAdd(A->counters + 96 + (why < FB_COUNT ? why : (uint32_t)FB_OTHER), 1ull);
					nbail++;
				}
			}
		}
		PROF(2);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		PROF(16);
		fq_push(Q, have, nextq, slot, lane);
		PROF(17);
	}
#ifdef H2G_GO_PROF
	if(lane == 0) for(int k = 0; k < 48; k++) if(prof[k]) atomicAdd(A->counters + 128 + k, prof[k]);
	if(lane == 0) for(int k = 0; k < 32; k++) if(prof_n[k]) { atomicAdd(A->counters + 176 + k, prof_ctl[k]); atomicAdd(A->counters + 208 + k, prof_n[k]); }
#endif
	wave_add(A->counters + 120, nrank);     // (slots of its own: the general machine's passes count in 0..5 / 64..69)
	wave_add(A->counters + 121, nside);
	wave_add(A->counters + 122, nsteps);
	wave_add(A->counters + 123, naln);
	wave_add(A->counters + 6, ndone);
	wave_add(A->counters + 7, nbail);
}

#define FG_LDS_BYTES ((unsigned)(((sizeof(FastLds) + 3) / 4 + (size_t)FG_STAGE_WORDS * H2G_FAST_THREADS) * 4))
extern "C" void FG_GEOMETRY(uint32_t* g) {
	g[0] = H2G_FAST_THREADS; g[1] = FG_LDS_BYTES; g[2] = H2G_FAST_SLOTS; g[3] = FG_SLOT_WORDS * 4u;
#if FG_GRAPH
	g[4] = FG_GWS_PRIVATE ? 0u : (uint32_t)sizeof(GraphWS);
#endif
}
// `a` is the argument block in DEVICE memory
extern "C" int FG_LAUNCH(const FastArgs* a, unsigned grid, hipStream_t st) {
	// more than 64 KB of dynamic LDS is an opt-in — per DEVICE (a process may drive several: hisat2-align-amd --gpus N), so the flag is a mask over device ids
	static std::atomic<unsigned long long> lds_ok{0};
	int dev = 0; (void)hipGetDevice(&dev);
	const unsigned long long bit = 1ull << (dev & 63);
	if(!(lds_ok.load(std::memory_order_relaxed) & bit)) { if(hipFuncSetAttribute((const void*)FG_KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError(); lds_ok.fetch_or(bit, std::memory_order_relaxed); }
	hipLaunchKernelGGL(FG_KERNEL, dim3(grid), dim3(H2G_FAST_THREADS), FG_LDS_BYTES, st, a);
	return (int)hipGetLastError();
}
