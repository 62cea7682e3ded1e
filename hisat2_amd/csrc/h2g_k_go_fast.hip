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
#include "h2g_fast_prof.h"

using namespace h2g;

#ifndef FG_KERNEL
#define FG_KERNEL   k_go_fast
#define FG_LAUNCH   h2g_go_fast_launch
#define FG_GEOMETRY h2g_go_fast_geometry
#endif
#define FG_CAT_(a, b) a##b
#define FG_CAT(a, b) FG_CAT_(a, b)
#define FG_KERNEL_DRAIN FG_CAT(FG_KERNEL, _drain)      // the same loop over the reads a launch of FG_KERNEL left in flight (FastArgs::adopt_list)
#define FG_LAUNCH_DRAIN FG_CAT(FG_LAUNCH, _drain)

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
	uint32_t quit;                      // the workgroup hands its reads in flight on (FastArgs::orphan_T)
	uint32_t parked;                    // slots that hold a read listed for the drain launch already (FastArgs::mate_handover): not free, not in flight
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
// Pops up to one slot of queue `q` for every lane with `want` set (in lane order).  Returns whether THIS lane got one.
__device__ __forceinline__ bool fq_pop_into(FastLds* Q, uint32_t q, int lane, bool want, uint32_t* slot) {
	const unsigned long long wm = __ballot(want);
	if(wm == 0) return false;
	const uint32_t nwant = (uint32_t)__popcll(wm);
	uint32_t n = 0, h = 0;
	if(lane == 0) {
		for(;;) {
			h = __atomic_load_n(&Q->head[q], __ATOMIC_RELAXED);
			const uint32_t t = __atomic_load_n(&Q->tail[q], __ATOMIC_RELAXED);
			n = t - h;
			if(n == 0) break;
			if(n > nwant) n = nwant;
			if(atomicCAS(&Q->head[q], h, h + n) == h) break;
		}
	}
	n = (uint32_t)__shfl((int)n, 0); h = (uint32_t)__shfl((int)h, 0);
	const uint32_t r = (uint32_t)__popcll(wm & ((1ull << lane) - 1ull));
	const bool got = want && r < n;
	if(got) {
		const uint32_t pos = (h + r) & (H2G_FAST_SLOTS - 1);
		uint16_t v;
		while((v = __atomic_load_n(&Q->ring[q][pos], __ATOMIC_RELAXED)) == FG_RING_EMPTY) __builtin_amdgcn_s_sleep(1);   // reserved, being written
		__atomic_store_n(&Q->ring[q][pos], (uint16_t)FG_RING_EMPTY, __ATOMIC_RELAXED);
		*slot = v;
	}
	return got;
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
	C.mate_handover = FG_ALIGN_MATE ? 0u : A->mate_handover;
	C.defer_slow = FG_GRAPH ? 1u : 0u;
	C.name[0] = C.name[1] = nullptr; C.namelen[0] = C.namelen[1] = 0;
}
// the state <-> its slot (16-byte accesses, nothing depends on anything)
__device__ __forceinline__ void fk_load_state(FState& S, const uint32_t* sm) {
	uint32_t w[FS_WORDS];
	const uint4* src = reinterpret_cast<const uint4*>(sm);
#pragma unroll
	for(uint32_t k = 0; k < FS_WORDS / 4; k++) { const uint4 v = src[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
	__builtin_memcpy(&S, w, sizeof S);
}
__device__ __forceinline__ void fk_store_state(const FState& S, uint32_t* sm) {
	uint32_t w[FS_WORDS];
	__builtin_memcpy(w, &S, sizeof S);
	uint4* dst = reinterpret_cast<uint4*>(sm);
#pragma unroll
	for(uint32_t k = 0; k < FS_WORDS / 4; k++) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
// One trip of this lane's read with the state in registers: the primitive `op` (FOP_NONE: a slot taking up read `begin`), then the control flow up to
// the next request.
__device__ __forceinline__ void fk_trip(const FastArgs* A, uint32_t* stage, uint32_t* sm, FState& S, uint32_t op, uint32_t begin, uint32_t packed_ok) {
	FCtx C; FWords W;
	fk_ctx(A, stage, sm, C, W);
#if FG_GRAPH && FG_GWS_PRIVATE
	GraphWS gws_private;
	C.gws = &gws_private;
#endif
	if(begin != H2G_MAX) {
		const bool paired = A->paired != 0;
		C.name[0] = A->names1 + A->noffs1[begin]; C.namelen[0] = A->noffs1[begin + 1] - A->noffs1[begin];
		if(paired) { C.name[1] = A->names2 + A->noffs2[begin]; C.namelen[1] = A->noffs2[begin + 1] - A->noffs2[begin]; }
		fast_begin(C, S, begin, paired, packed_ok != 0);
	} else {
		// the work counters are 16-bit fields: the primitive counts from zero, a read that would wrap them leaves the fast path
		const uint32_t nr0 = S.nrank, ns0 = S.nside, nt0 = S.nsteps;
		S.nrank = 0; S.nside = 0; S.nsteps = 0;
		fast_exec(C, S, W, op == FOP_PER_LANE ? (uint32_t)S.op : op);
		const uint32_t nr_ = nr0 + S.nrank, ns_ = ns0 + S.nside, nt_ = nt0 + S.nsteps;
		if(S.pc != FPC_BAIL && (nr_ > 0xffffu || ns_ > 0xffffu || nt_ > 0xffffu)) { S.pc = FPC_BAIL; S.bail = FB_OTHER; }
		S.nrank = nr_ & 0xffffu; S.nside = ns_ & 0xffffu; S.nsteps = nt_ & 0xffffu;
	}
	if(S.pc != FPC_DONE && S.pc != FPC_BAIL && S.op == FOP_NONE) fast_step(C, S, W);
}

// a read that leaves its lane: state + hot words to its slot (the packed reads and the cold words are there already)
__device__ __forceinline__ void fk_store_read(const FState& S, uint32_t* sm, const uint32_t* stage) {
	fk_store_state(S, sm);
	uint4* hdst = reinterpret_cast<uint4*>(sm + FS_WORDS);
#pragma unroll
	for(uint32_t k = 0; k < FW_HOT / 4; k++)
		hdst[k] = make_uint4(stage[(4 * k) * H2G_FAST_THREADS], stage[(4 * k + 1) * H2G_FAST_THREADS], stage[(4 * k + 2) * H2G_FAST_THREADS], stage[(4 * k + 3) * H2G_FAST_THREADS]);
#pragma unroll
	for(uint32_t k = FW_HOT & ~3u; k < FW_HOT; k++) sm[FS_WORDS + k] = stage[k * H2G_FAST_THREADS];
}
// ... and a read that enters one: state into registers, hot words + packed reads into this lane's LDS staging area (16-byte loads, nothing depends on anything)
__device__ __forceinline__ void fk_load_read(FState& S, const uint32_t* sm, uint32_t* stage) {
	fk_load_state(S, sm);
	const uint4* hsrc = reinterpret_cast<const uint4*>(sm + FS_WORDS);
#pragma unroll
	for(uint32_t k = 0; k < FG_STAGE_WORDS / 4; k++) {
		const uint4 v = hsrc[k];
		stage[(4 * k) * H2G_FAST_THREADS] = v.x; stage[(4 * k + 1) * H2G_FAST_THREADS] = v.y; stage[(4 * k + 2) * H2G_FAST_THREADS] = v.z; stage[(4 * k + 3) * H2G_FAST_THREADS] = v.w;
	}
#pragma unroll
	for(uint32_t k = FG_STAGE_WORDS & ~3u; k < FG_STAGE_WORDS; k++) stage[k * H2G_FAST_THREADS] = sm[FS_WORDS + k];
}
// appends the slots of the lanes with `valid` set to the launch's orphan list (one reservation per wave)
__device__ __forceinline__ void fk_orphan(const FastArgs* A, bool valid, uint32_t slot, int lane) {
	const unsigned long long m = __ballot(valid);
	if(m == 0) return;
	uint32_t base = 0;
	if(lane == 0) base = atomicAdd(A->orphan_count, (uint32_t)__popcll(m));
	base = (uint32_t)__shfl((int)base, 0);
	if(valid) A->orphan_list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = blockIdx.x * H2G_FAST_SLOTS + slot;
}

#define FG_Q_NONE 0xffffffffu
// The loop of one wave (round 6: reads stay in their lane).  A lane that has run a read's control flow up to its next request KEEPS the read — state in
// registers, hot words in its LDS staging area — and the wave looks at what its own lanes ask for next to the queues: it runs the site that fills most of
// its lanes (own lanes first: they cost no load).  Only the lanes that do not take part in that trip store their read (state + hot words; the packed reads
// never change) and push its slot; free lanes pop from the site's queue.  Through round 5 every trip stored and re-loaded every read: half of the kernel's
// fabric requests.  Which reads run together changes; what a read computes does not (every read is a function of itself).
//
// The end of a batch (round 6, lease L: `profiles/r06_l_*`): once the batch is exhausted a workgroup's population decays with the reads' own latency chains — a third of the
// kernel's time went by with 256 workgroups holding a few dozen reads each.  With FastArgs::orphan_T set, a workgroup that can fetch no more and holds at most that many reads
// stores them and lists their slots (`orphan_list`), and leaves; FG_KERNEL_DRAIN — this loop with ADOPT — takes the listed reads up on a few workgroups next to the following
// batch's launch: slots copied into its own pool, resumed at the request they were waiting at.  A read is a function of itself: which launch finishes it changes nothing it writes.
template <bool ADOPT>
__device__ __forceinline__ void fk_loop(const FastArgs* __restrict__ A, uint32_t* s_mem)
{
	FastLds* Q = reinterpret_cast<FastLds*>(s_mem);
	uint32_t* const stage = s_mem + (sizeof(FastLds) + 3) / 4 + threadIdx.x;      // word w of this lane at stage[w * H2G_FAST_THREADS]
	const int lane = (int)(threadIdx.x & 63);
	const unsigned long long lt = (1ull << lane) - 1ull;
	const bool paired = A->paired != 0;
	for(uint32_t k = threadIdx.x; k < (uint32_t)FG_NQ * H2G_FAST_SLOTS; k += blockDim.x) (&Q->ring[0][0])[k] = FG_RING_EMPTY;
	if(threadIdx.x < (uint32_t)FG_NQ) { Q->head[threadIdx.x] = 0; Q->tail[threadIdx.x] = 0; }
	if(threadIdx.x == 0) { Q->quit = 0; Q->parked = 0; }
	__syncthreads();
	for(uint32_t k = threadIdx.x; k < H2G_FAST_SLOTS; k += blockDim.x) Q->ring[0][k] = (uint16_t)k;   // every slot starts free
	if(threadIdx.x == 0) Q->tail[0] = H2G_FAST_SLOTS;
	__syncthreads();
	uint32_t* const pk0 = stage + FW_HOT * H2G_FAST_THREADS;
	uint32_t* const pk1 = pk0 + H2G_PK_WORDS * H2G_FAST_THREADS;
	uint32_t* const slots0 = A->slots + (size_t)blockIdx.x * H2G_FAST_SLOTS * FG_SLOT_WORDS;
	unsigned long long nrank = 0, nside = 0, nsteps = 0, naln = 0, ndone = 0, nbail = 0;
	const uint32_t total = ADOPT ? *A->adopt_count : A->total, tail_n = A->tail;
	const uint32_t orphan_T = ADOPT ? 0u : A->orphan_T;
	unsigned long long* const cnt = A->counters + A->cnt_off;
	bool more = total != 0;
	FPROF_DECL;
	// what this lane holds between two trips
	FState S;
	bool keep = false;              // a slot is in this lane: a read up to its next request (myq >= 1) or a slot whose read is over (myq == FQ_FREE)
	uint32_t slot = 0, myq = 0;
	uint32_t* sm = nullptr;         // the slot in HBM
	__builtin_memset(&S, 0, sizeof S);
	for(;;) {
		// ---- choose: per site, the lanes a trip would fill = this wave's own lanes waiting for it + its queue
		uint32_t cnt = 0, own = 0;
		if(lane < FG_NQ) cnt = __atomic_load_n(&Q->tail[lane], __ATOMIC_RELAXED) - __atomic_load_n(&Q->head[lane], __ATOMIC_RELAXED);
#pragma unroll
		for(uint32_t q = 0; q < (uint32_t)FG_NQ; q++) { const uint32_t c = (uint32_t)__popcll(__ballot(keep && myq == q)); if((uint32_t)lane == q) own = c; }
		const uint32_t nfree = (uint32_t)__shfl((int)cnt, 0), own_free = (uint32_t)__shfl((int)own, 0);
		uint32_t key = 0, bestq = (uint32_t)lane;
		if(lane >= 1 && lane < FG_NQ) { const uint32_t act = own + cnt > 64u ? 64u : own + cnt; key = act ? ((act << 8) | own) : 0u; }
		for(int o = 32; o > 0; o >>= 1) {
			const uint32_t ok = (uint32_t)__shfl_xor((int)key, o), oq = (uint32_t)__shfl_xor((int)bestq, o);
			if(ok > key || (ok == key && oq < bestq)) { key = ok; bestq = oq; }
		}
		key = (uint32_t)__shfl((int)key, 0); bestq = (uint32_t)__shfl((int)bestq, 0);
		const uint32_t bestact = key >> 8, avail = nfree + own_free;
		uint32_t parked = 0;
		if(!ADOPT && !FG_ALIGN_MATE && orphan_T) { if(lane == 0) parked = __atomic_load_n(&Q->parked, __ATOMIC_RELAXED); parked = (uint32_t)__shfl((int)parked, 0); }
		// new reads: when that trip would fill more lanes than any site's, or a quarter of the slots lie free
		const bool fetch = more && avail > 0 && ((avail > 64u ? 64u : avail) > bestact || nfree >= H2G_FAST_SLOTS / 4);
		const uint32_t qstar = fetch ? (uint32_t)FQ_FREE : (bestact ? bestq : FG_Q_NONE);
		// ---- the batch is exhausted and the workgroup has thinned out: every read in flight goes to the drain launch (this wave's here, the queued ones behind the loop)
		if(!ADOPT && orphan_T) {
			uint32_t quit = 0;
			if(lane == 0) {
				quit = __atomic_load_n(&Q->quit, __ATOMIC_RELAXED);
				if(!quit && !more && H2G_FAST_SLOTS - nfree - parked <= orphan_T) { __atomic_store_n(&Q->quit, 1u, __ATOMIC_RELAXED); quit = 1; }
			}
			if(__shfl((int)quit, 0)) {
				const bool out = keep && myq != FQ_FREE;
				if(out) fk_store_read(S, sm, stage);
				fk_orphan(A, out, slot, lane);
				break;
			}
		}
		// ---- the lanes that do not take part hand their slots on
		{
			const bool out = keep && myq != qstar;
			if(out && myq != FQ_FREE) fk_store_read(S, sm, stage);
			FPROF(2);
			if(__ballot(out)) {
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				fq_push(Q, out, myq, slot, lane);
			}
			if(out) keep = false;
			FPROF(17);
		}
		if(qstar == FG_Q_NONE) {
			if(!more && nfree + parked == H2G_FAST_SLOTS) break;  // the batch is exhausted and every slot is free again (or parked for the drain launch)
			__builtin_amdgcn_s_sleep(8);
			if(more) { uint32_t w = 0; if(lane == 0) w = __atomic_load_n(A->work, __ATOMIC_RELAXED); if((uint32_t)__shfl((int)w, 0) >= total) more = false; }
			continue;
		}
		bool active = false, tail = false, fresh = false;
		uint32_t begin = H2G_MAX, packed_ok = 0, trip_op = FOP_NONE;
		if(fetch) {
			const bool got = fq_pop_into(Q, FQ_FREE, lane, !keep, &slot);    // (a lane whose read is over re-uses its slot without the queue)
			const bool has = keep || got;
			const unsigned long long hm = __ballot(has);
			const uint32_t n = (uint32_t)__popcll(hm);
			if(n == 0) continue;
			uint32_t base = 0;
			if(lane == 0) base = atomicAdd(A->work, n);
			base = (uint32_t)__shfl((int)base, 0);
			if(base + n >= total) more = false;
			const uint32_t r = (uint32_t)__popcll(hm & lt);
			active = has && base + r < total;
			if(has && !active) { keep = true; myq = FQ_FREE; }        // a slot without a read goes back with the next trip's hand-ons
			if(ADOPT && active) {
				// a read another launch left in flight: its slot into this workgroup's pool, the read into this lane — it waits for the request it was stored at
				sm = slots0 + (size_t)slot * FG_SLOT_WORDS;
				const uint32_t sw = A->adopt_slot_words;               // (16-byte multiples both; the other build's slot ends 4 words earlier or later)
				const uint32_t* const src = A->adopt_slots + (size_t)A->adopt_list[base + r] * sw;
				const uint4* s4 = reinterpret_cast<const uint4*>(src); uint4* d4 = reinterpret_cast<uint4*>(sm);
				const uint32_t nq = (sw < (uint32_t)FG_SLOT_WORDS ? sw : (uint32_t)FG_SLOT_WORDS) / 4;
#pragma unroll 8
				for(uint32_t k = 0; k < nq; k++) d4[k] = s4[k];
				fk_load_read(S, src, stage);
				if(S.op == FOP_NONE) trip_op = FOP_NONE;                // parked between two states (FOP_HANDOVER): its control flow runs now, below
				else { keep = true; myq = fg_queue_of_state(S); active = false; }
			}
			if(!ADOPT && active) {
				keep = false;
				begin = base + r;
				sm = slots0 + (size_t)slot * FG_SLOT_WORDS;
				bool ok = fg_pack_read(A->rd1, begin, pk0, H2G_FAST_THREADS);
				if(paired) ok = fg_pack_read(A->rd2, begin, pk1, H2G_FAST_THREADS) && ok;
				packed_ok = ok ? 1u : 0u;
				// the packed reads are written to the slot once
#pragma unroll
				for(uint32_t k = 0; k < 2 * H2G_PK_WORDS; k++) sm[FS_WORDS + FW_HOT + k] = pk0[k * H2G_FAST_THREADS];
			}
			FPROF_SITE(0);
			FPROF(15);
		} else {
			trip_op = fg_queue_op(qstar);
			fresh = fq_pop_into(Q, qstar, lane, !keep, &slot);
			if(__ballot(fresh)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			active = keep || fresh;
			// (FastArgs::tail: the last reads in flight of an exhausted batch are a few latency chains per workgroup; they can go to the general
			// machine's pass, which is in flight anyway — profiles/r04_NOTES.md §1, §4)
			tail = !more && H2G_FAST_SLOTS - nfree <= tail_n;
			if(fresh) {
				sm = slots0 + (size_t)slot * FG_SLOT_WORDS;
				if(tail) S.read = sm[8];                              // (state word 8 = the read id)
				else fk_load_read(S, sm, stage);
			}
			FPROF_EXEC(trip_op > 11u ? 11u : trip_op, __popcll(__ballot(active)), __popcll(__ballot(fresh)));
			FPROF_SITE(qstar);
			FPROF(0);
		}
		// ---- control flow of each read up to its next primitive request
		FPROF_TRIP(__popcll(__ballot(active)));
		FPROF_TBIN(A->counters, __popcll(__ballot(active)), H2G_FAST_SLOTS - nfree);
		if(active && tail) { S.pc = FPC_BAIL; S.bail = FB_TAIL; }
		else if(active) fk_trip(A, stage, sm, S, trip_op, begin, packed_ok);
		FPROF_CTL();
		FPROF(1);
		if(!ADOPT && !FG_ALIGN_MATE) {
			// a pair that needs alignMate: parked in its slot and listed for the drain launch (the alignMate build of this loop) — up to a quarter of the workgroup's slots, then handed on
			const bool ho = active && S.op == FOP_HANDOVER;
			const unsigned long long hm = __ballot(ho);
			if(hm) {
				uint32_t old = 0;
				if(lane == 0) old = atomicAdd(&Q->parked, (uint32_t)__popcll(hm));
				old = (uint32_t)__shfl((int)old, 0);
				const bool ok = ho && old + (uint32_t)__popcll(hm & lt) < H2G_FAST_SLOTS / 4;
				const uint32_t nfail = (uint32_t)__popcll(hm) - (uint32_t)__popcll(__ballot(ok));
				if(lane == 0 && nfail) atomicSub(&Q->parked, nfail);
				if(ho) S.op = FOP_NONE;
				if(ok) fk_store_read(S, sm, stage);
				fk_orphan(A, ok, slot, lane);
				if(ok) { active = false; keep = false; }
				else if(ho) { S.pc = FPC_BAIL; S.bail = FB_MATE; }
			}
		}
		const uint32_t pc = S.pc;
		if(active) {
			keep = true;
			if(pc == FPC_DONE) { nrank += S.nrank; nside += S.nside; nsteps += S.nsteps; naln += S.a0 != 0; ndone++; myq = FQ_FREE; }
			else if(pc == FPC_BAIL) myq = FQ_FREE;
			else myq = fg_queue_of_state(S);
		}
		// reads that left the fast path: their ids go to the general machine's list
		{
			const bool b = active && pc == FPC_BAIL;
			const unsigned long long bm = __ballot(b);
			if(bm) {
				uint32_t base = 0;
				if(lane == 0) base = atomicAdd(A->bail_count, (uint32_t)__popcll(bm));
				base = (uint32_t)__shfl((int)base, 0);
				if(b) {
					const uint32_t why = S.bail;
					A->bail_list[base + (uint32_t)__popcll(bm & lt)] = S.read;
					atomicAdd(A->counters + 96 + (why < FB_COUNT ? why : (uint32_t)FB_OTHER), 1ull);
					nbail++;
				}
			}
		}
		FPROF(16);
	}
	if(!ADOPT && orphan_T) {
		// the reads that wait in the queues: listed once every wave has left the loop (a workgroup that ran dry has empty queues)
		__syncthreads();
		if(Q->quit) {
			for(uint32_t q = 1 + (threadIdx.x >> 6); q < (uint32_t)FG_NQ; q += H2G_FAST_THREADS / 64) {
				const uint32_t h = Q->head[q], n = Q->tail[q] - h;
				for(uint32_t i0 = 0; i0 < n; i0 += 64) {
					const bool v = i0 + (uint32_t)lane < n;
					fk_orphan(A, v, v ? (uint32_t)Q->ring[q][(h + i0 + (uint32_t)lane) & (H2G_FAST_SLOTS - 1)] : 0u, lane);
				}
			}
		}
	}
	FPROF_FLUSH(A->counters);
	wave_add(cnt + 120, nrank);     // (slots of its own: the general machine's passes count in 0..5 / 64..69; the drain launch's cnt_off moves its four)
	wave_add(cnt + 121, nside);
	wave_add(cnt + 122, nsteps);
	wave_add(cnt + 123, naln);
	wave_add(A->counters + 6, ndone);
	wave_add(A->counters + 7, nbail);
}
__global__ __launch_bounds__(H2G_FAST_THREADS) void FG_KERNEL(const FastArgs* __restrict__ A)
{
	extern __shared__ uint32_t s_mem[];
	fk_loop<false>(A, s_mem);
}
__global__ __launch_bounds__(H2G_FAST_THREADS) void FG_KERNEL_DRAIN(const FastArgs* __restrict__ A)
{
	extern __shared__ uint32_t s_mem[];
	fk_loop<true>(A, s_mem);
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
extern "C" int FG_LAUNCH_DRAIN(const FastArgs* a, unsigned grid, hipStream_t st) {
	static std::atomic<unsigned long long> lds_ok{0};
	int dev = 0; (void)hipGetDevice(&dev);
	const unsigned long long bit = 1ull << (dev & 63);
	if(!(lds_ok.load(std::memory_order_relaxed) & bit)) { if(hipFuncSetAttribute((const void*)FG_KERNEL_DRAIN, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError(); lds_ok.fetch_or(bit, std::memory_order_relaxed); }
	hipLaunchKernelGGL(FG_KERNEL_DRAIN, dim3(grid), dim3(H2G_FAST_THREADS), FG_LDS_BYTES, st, a);
	return (int)hipGetLastError();
}
