// h2g_kernels.hip — gfx950 kernels and the C ABI of libh2g.so (include/h2g.h).
//
// Integer / byte work bound by random 64 B HBM reads: no MFMA anywhere.  Launch geometry: 256-thread
// workgroups (4 waves of 64), grids of >= 8 blocks per CU x 256 CUs so all 8 XCDs stay busy; sides are
// read with 16 B-per-lane vector loads so one side = one 64 B request.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <thread>
#include <atomic>
#include <chrono>
#include <algorithm>
#define H2G_EXT_OPTS 0      // likewise -I / --fr --rf --ff / --nofw --norc (h2g_align.h): go() units only
#define H2G_HAPLOTYPE 0     // the primitive kernels of this unit (k_extend_alts, k_adjust_alt) run without haplotype lists; go() units: h2g_graph.h
#include "h2g_core.h"
#include "h2g_host_index.h"
#include "h2g_align.h"
#include "h2g_graph.h"
#include "h2g_sw.h"
#include "h2g_graph_staged.h"
#include "h2g_local_pack.h"
#include "h2g_splice_host.h"
#include "h2g_splice_db_host.h"
#include "h2g_go_args.h"   // GoArgs + the extern "C" face of the go() units (their AlignWS layouts are opaque on this side)

using namespace h2g;

// ------------------------------------------------------------------------------------------ error plumbing
static thread_local char g_err[512] = "";
static int set_err(const char* what, hipError_t e) {
	snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
	return H2G_ERR_DEVICE;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if(e_ != hipSuccess) return set_err(#call, e_); } while(0)

extern "C" const char* h2g_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------ handles
struct h2g_index {
	HostIndex host;
	bool synthetic = false;
	int device = 0;
	DGfm dg;
	DRef dr;
	DLocalSet dls;
	DAlts dalts;
	bool has_local = false;
	uint32_t n_local = 0;                                  // local indexes loaded (their per-index host objects are not kept: load_local_pack)
	bool has_splice_alts = false;                         // the ALT list holds splice sites / exons (a _tran index)
	std::vector<h2g_splice_site> alt_sites;               // splice-site ALTs of a --ss index: part of every database (SpliceSiteDB::read(gfm, alts))
	DSpliceDB dssdb;                                       // h2g_index_set_splice_sites (device arrays; freed and replaced on every call)
	void* d_ssdb[4] = {nullptr, nullptr, nullptr, nullptr};
	size_t ssdb_cap[4] = {0, 0, 0, 0};                     // bytes behind each (they grow by doubling: temporary splice sites arrive wave after wave)
	HostSpliceDB h_ssdb;                                   // the host copy h2g_index_add_splice_sites merges into
	const float* d_spl[3] = {nullptr, nullptr, nullptr};   // SpliceSiteDB::probscore tables (donor, acceptor halves), uploaded with the index
	std::vector<DLocalDesc> h_ldesc;   // host copy of the local-index descriptors (bucketing of h2g_ext_search)
	std::vector<void*> allocs;
	uint64_t device_bytes = 0;
};

// The general machine's pass over run k's hand-ons is a LATENCY CHAIN: its length is its longest reads' (hundreds of dependent trips of
// 60-100 us on a 115 KB workspace), not their number — 30 ms behind a 14.5 ms fast pass on the random GRCh38-size genome, 118 ms behind a 25 ms
// fast pass on repeat-structured sequence (profiles/r04_NOTES.md §4-§6).  The stream therefore keeps up to H2G_MSTREAMS_MAX such passes in flight,
// one per machine stream, next to the fast passes of the following runs: run k's pass goes to machine stream k % M and the hand-on list, the
// counters and the argument block are buffered M + 1 deep.  M is a property of the stream (h2g_stream_tune "mstreams", H2G_MSTREAMS; default
// H2G_MSTREAMS_DEFAULT).  A caller sees no difference: results are complete when a fetch / sync returns.
#define H2G_MSTREAMS_MAX 8
#define H2G_MSTREAMS_DEFAULT 8
#define H2G_NBUF (H2G_MSTREAMS_MAX + 1)
// CUs the fast pass's persistent grid leaves free for the machine passes: -1 = as many as M passes may hold; 0 = none — the fast pass asks for every
// CU and the hardware dispatcher places machine workgroups as CUs come free (the fast pass draws its reads from one counter, so a late workgroup costs nothing)
#define H2G_FAST_RESERVE_DEFAULT 0
// the fast pass's two scheduling choices for PAIRED batches on a linear index (measured at GRCh38 size: profiles/r04_NOTES.md §4): reads a
// workgroup hands on at the tail of an exhausted batch, and whether alignMate runs inside the pass (k_go_fast_am) or in the machine's pass
#define H2G_DEFAULT_TAIL 16
// The end of a batch (h2g_k_go_fast.hip, fk_loop): workgroups of a fast launch that can fetch no more and hold at most H2G_ORPHAN_T reads list them and leave; the drain launch
// (H2G_DRAIN_GRID workgroups on the stream's drain stream; the machine's pass waits for it) finishes them next to the following run's fast launch.  Drain launches run one after
// the other, so one has to be shorter than a step: 64 workgroups (lease Q: with 32 and the alignMate pairs in it the drain launch, 10 ms, became the step).  Batches below
// H2G_ORPHAN_MIN_UNITS keep the single launch (h2g_stream_tune "orphan": -1 this policy, 0 off, n the threshold; "drain_grid"; "mate_handover").
#define H2G_ORPHAN_T 512
#define H2G_ORPHAN_MIN_UNITS 200000
#define H2G_DRAIN_GRID 64
#define H2G_FAST_POOLS 3
#define H2G_DEFAULT_ALIGN_MATE 0
#ifdef H2G_GO_PROF
#define H2G_CNT_BLOCK 1024u         // (+ the time-resolved bins of h2g_fast_prof.h at [512, 768))
#else
#define H2G_CNT_BLOCK 512u
#endif
//                            // counter words per go_run generation: [0, 256) the main / fast pass (+ their profiling slots), [256, 512) the second pass
#define H2G_MACH_MAXGRID 48u       // workgroups of ONE machine pass behind a fast pass
#define H2G_MACH_TOTAL 128u        // ... and of all machine passes in flight together ("mach_total": a pass gets at most mach_total / mstreams workgroups)
// Resident batches (round 6): a stream holds up to H2G_MAX_BATCHES read sets WITH their result rows (h2g_stream_select_batch).  The named fields of h2g_stream are the
// selected batch's; the others are parked in h2g_stream::parked.  A run captures its batch's pointers in its argument block when it is queued, so runs over different batches
// are in flight together — fast passes and up to eight machine passes — each writing its own rows (a streaming caller's steady state: SURVEY §8(d) configs[2] is 10 M pairs, not one
// million ten times).
#define H2G_MAX_BATCHES 16
#define H2G_BATCH_FIELDS(X) X(n_reads) X(max_read_len) X(d_codes) X(d_offs) X(d_quals) X(has_quals) X(d_names) X(d_name_offs) X(names_cap) X(has_names) \
	X(d_codes2) X(d_offs2) X(d_quals2) X(d_names2) X(d_name_offs2) X(has_mates) X(has_quals2) X(d_rout) X(d_aln) X(aln_alloc) X(aln_slots) X(d_pout) X(paln_alloc) X(pair_slots)
struct BatchCtx {
	size_t n_reads = 0; uint32_t max_read_len = 0;
	uint8_t* d_codes = nullptr; uint32_t* d_offs = nullptr; char* d_quals = nullptr; bool has_quals = false;
	char* d_names = nullptr; uint32_t* d_name_offs = nullptr; size_t names_cap = 0; bool has_names = false;
	uint8_t* d_codes2 = nullptr; uint32_t* d_offs2 = nullptr; char* d_quals2 = nullptr; char* d_names2 = nullptr; uint32_t* d_name_offs2 = nullptr; bool has_mates = false, has_quals2 = false;
	ReadOut* d_rout = nullptr; h2g_alnres* d_aln = nullptr; size_t aln_alloc = 0; uint32_t aln_slots = 0;
	PairOut* d_pout = nullptr; h2g_alnres* d_paln[2] = {nullptr, nullptr}; size_t paln_alloc = 0; uint32_t pair_slots = 0;
};
struct h2g_stream {
	h2g_index* ix = nullptr;
	hipStream_t st = nullptr;
	// the general machine's pass over a fast pass's hand-ons runs on one of these, next to the fast passes of the FOLLOWING two batches:
	// its few reads are long latency chains (about a fast pass's duration whatever their number), so two such passes are kept in flight
	hipStream_t mst[H2G_MSTREAMS_MAX] = {};
	unsigned mstreams = H2G_MSTREAMS_DEFAULT;   // machine passes kept in flight (1 .. H2G_MSTREAMS_MAX)
	bool st2_busy = false;            // a machine stream may hold work
	hipEvent_t ev_fast[H2G_NBUF], ev_mach[H2G_NBUF];
	unsigned gen = 0;                 // go_run generation: bail list, counters and argument block are buffered H2G_NBUF deep by gen % H2G_NBUF
	unsigned long long* cnt_cur = nullptr;   // the counter block of the last go_run
	uint32_t last_bails = 0;
	uint32_t* h_bails = nullptr;      // pinned: the hand-on count of the last two fast passes (sizes the machine's share of the CUs)
	size_t max_reads = 0, max_bases = 0, n_reads = 0;
	uint32_t max_read_len = 0;
	uint8_t* d_codes = nullptr;
	uint32_t* d_offs = nullptr;
	char* d_quals = nullptr;
	bool has_quals = false;
	h2g_seed_result* d_seed = nullptr;
	char* d_names = nullptr;
	uint32_t* d_name_offs = nullptr;
	size_t names_cap = 0;
	bool has_names = false;
	// per-lane scratch of the go() kernels: [0] the main pass, [1] the second pass over overflowed reads (large workspace)
	struct GoPool {
		uint8_t* ws = nullptr;  size_t ws_bytes = 0;      // AlignWS x lanes of the unit last run
		uint8_t* gws = nullptr; size_t gws_bytes = 0;     // GraphWS x lanes (graph indexes only)
		uint8_t* sw = nullptr;  size_t sw_stride = 0, sw_lanes = 0;   // Smith-Waterman scratch (only with bowtie2_dp != 0)
		uint8_t* sc = nullptr;  size_t sc_lanes = 0;                  // combineWith temp_scores per lane
	} pool[2 * H2G_MSTREAMS_MAX];     // [2 * m + 0] main pass, [2 * m + 1] second pass of machine stream m (m = 0 also: passes on the first stream)
	uint32_t* dbg_buf = nullptr;      // development hook (H2G_GO_DBG_READ)
	uint32_t* d_ovf_list[H2G_MSTREAMS_MAX] = {};   // per machine stream: read ids whose workspace overflowed in the main pass (+ their count behind the list)
	unsigned ovf_cur = 0;             // the one the last run used
	uint32_t* d_bail_list[H2G_NBUF] = {};  // read ids the fast pass handed on to the general machine (+ their count behind the list)
	void* d_fast_args[2 * H2G_NBUF] = {};  // [gen % NBUF] the fast launch's, [NBUF + gen % NBUF] its drain launch's
	void* h_fast_args = nullptr;      // pinned staging of the argument blocks (H2G_NBUF of them): the upload never makes the host wait for the stream      // the fast pass's argument block (device copy)
	// the fast pass's reads in flight (h2g_k_go_fast.hip): pool gen % 3 — the drain launch of run k reads run k's pool while the fast passes of runs k + 1 and k + 2 run
	uint32_t* d_fast_slots[H2G_FAST_POOLS] = {}; size_t fast_slot_bytes[H2G_FAST_POOLS] = {};
	hipEvent_t ev_pool[H2G_FAST_POOLS];                              // the drain launch that last read pool p is over
	uint32_t* d_orphans[H2G_NBUF] = {};                              // slots a fast launch left in flight (+ their count behind the list)
	size_t orphan_cap = 0;
	hipStream_t dst = nullptr;                                       // the drain launches' stream: one after the other, each next to the following run's fast launch
	uint32_t* d_drain_slots = nullptr; size_t drain_slot_bytes = 0;  // the drain launch's own pool
	uint8_t* d_drain_sc = nullptr; size_t drain_sc_bytes = 0;        // graph indexes: its combineWith scores per lane
	hipEvent_t ev_drain[H2G_NBUF];                                   // run k's drain launch is over (its machine pass waits for it)
	bool dst_busy = false;
	hipEvent_t ev_bails[H2G_NBUF];                                   // h_bails[gen % NBUF] has arrived
	hipEvent_t ev_dr[2];                                             // (timed) around the last drain launch
	bool ran_drain = false;
	const uint32_t* orph_cur = nullptr;                              // the orphan count of the last run
	uint8_t* d_fast_gws = nullptr; size_t fast_gws_bytes = 0;       // graph indexes: GraphWS per lane of the fast kernel (scratch of one primitive)
	uint8_t* d_fast_sc = nullptr; size_t fast_sc_bytes = 0;         // ... and combineWith's temp_scores per lane
	bool ran_fast = false;
	// development / measurement knobs (h2g_stream_tune; their H2G_* environment names are read ONCE, when the stream is created)
	struct Tune { int fast = 1, blocks_per_cu = 0, pair_slots = 0, no_second_pass = 0; unsigned mach_div = 0 /* auto */, mach_min = 4, mach_total = H2G_MACH_TOTAL; int mach_total_auto = 1; int fast_reserve = H2G_FAST_RESERVE_DEFAULT; long dbg_read = -1;
	              int tail = H2G_DEFAULT_TAIL, tail_auto = 1, align_mate = H2G_DEFAULT_ALIGN_MATE; int orphan = -1 /* auto */, drain_grid = H2G_DRAIN_GRID, mate_handover = -1 /* auto */; } tune;
	h2g_align_params last_p; int last_paired = -1;   // the option set of the last go_run (a different one waits for the machine streams)
	uint32_t aln_slots = 0;           // alignment records kept per unpaired read in d_aln (>= -k of the last run)
	uint32_t pair_slots = 0;          // report events kept per mate in d_paln (>= H2G_PAIR_RES_CAP; grows with -k)
	size_t paln_alloc = 0;
	size_t aln_alloc = 0;             // records allocated behind d_aln
	uint8_t* d_sw_ws = nullptr;   // h2g_sw_align: H/E/F workspace of one batch of problems
	size_t sw_ws_bytes = 0;
	SwLaneState* d_sw_states = nullptr;
	size_t sw_states = 0;
	ReadOut* d_rout = nullptr;
	h2g_alnres* d_aln = nullptr;
	uint8_t* d_codes2 = nullptr;
	uint32_t* d_offs2 = nullptr;
	char* d_quals2 = nullptr;
	char* d_names2 = nullptr;
	uint32_t* d_name_offs2 = nullptr;
	bool has_mates = false, has_quals2 = false;
	PairOut* d_pout = nullptr;
	h2g_alnres* d_paln[2] = {nullptr, nullptr};
	// records with more than H2G_MAX_EDITS edits keep their lists here (MachOut::ledits): one part of ledits_cap edits and one cursor per machine stream
	h2g_edit* d_ledits = nullptr; uint32_t* d_ledits_cur = nullptr; size_t ledits_cap = 0; unsigned ledits_parts = 0;
	unsigned ledits_touched = 0;   // parts a run has written to since the last h2g_set_reads (bit per machine stream): the span h2g_align_fetch_long_edits reports
	h2g_alnres* d_paln_ovf = nullptr; size_t paln_ovf_cap = 0; unsigned paln_ovf_parts = 0;   // pairs with more records than pair_slots per mate (MachOut::ovf): one part of paln_ovf_cap records per machine stream
	unsigned long long* d_counters = nullptr;   // [8]
	void* d_tmp[4] = {nullptr, nullptr, nullptr, nullptr};
	size_t tmp_sz[4] = {0, 0, 0, 0};
	hipEvent_t ev[12];
	bool ran_seed = false, ran_align = false;
	h2g_counters last;
	bool mstreams_warm = false;          // every machine stream has run its two kernels once (go_run, large batches)
	unsigned long long* d_warm_cnt = nullptr;   // counter block of those empty launches
	BatchCtx parked[H2G_MAX_BATCHES];   // the batches that are not selected ([cur_batch] is unused: its fields are the stream's own)
	unsigned cur_batch = 0;
};

// Large arrays travel through page-locked staging buffers filled by several threads: a pageable hipMemcpy of a human-size index (4.7 GB) was most of its 1.2 s load (round 6).
// Chunk k of a copy is thread k mod T's: wait for its staging buffer, memcpy (from a host vector or straight from a mapped index file: the page faults spread over the threads
// too), hipMemcpyAsync on the thread's own stream.
struct Stager {
	enum { T = 6, NB = 2 };
	static constexpr size_t CH = (size_t)16 << 20;
	uint8_t* buf[T][NB] = {}; hipStream_t st[T] = {}; hipEvent_t ev[T][NB] = {};
	int device = 0; bool ok = false;
	bool init(int dev) {
		device = dev; ok = true;
		for(int t = 0; t < T && ok; t++) {
			ok = hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking) == hipSuccess;
			for(int b = 0; b < NB && ok; b++) ok = hipHostMalloc((void**)&buf[t][b], CH, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&ev[t][b], hipEventDisableTiming) == hipSuccess;
		}
		if(!ok) (void)hipGetLastError();
		return ok;
	}
	~Stager() { for(int t = 0; t < T; t++) { for(int b = 0; b < NB; b++) { if(buf[t][b]) (void)hipHostFree(buf[t][b]); if(ev[t][b]) (void)hipEventDestroy(ev[t][b]); } if(st[t]) (void)hipStreamDestroy(st[t]); } }
	bool copy(void* dst, const void* src, size_t bytes) {
		return gather(dst, bytes, [src](uint8_t* out, size_t off, size_t len) { memcpy(out, (const uint8_t*)src + off, len); });
	}
	// the same pipeline with the staging buffers filled by `fill(out, off, len)`: bytes [off, off + len) of an array that exists nowhere on the host in one piece
	template <class F>
	bool gather(void* dst, size_t bytes, F fill) {
		std::atomic<bool> good{true};
		std::thread th[T];
		const size_t nch = (bytes + CH - 1) / CH;
		for(int t = 0; t < T; t++) th[t] = std::thread([&, t]() {
			if(hipSetDevice(device) != hipSuccess) { good = false; return; }
			size_t n = 0;
			for(size_t k = (size_t)t; k < nch; k += T, n++) {
				const int b = (int)(n % NB);
				const size_t off = k * CH, len = bytes - off < CH ? bytes - off : CH;
				if(n >= NB && hipEventSynchronize(ev[t][b]) != hipSuccess) { good = false; return; }
				fill(buf[t][b], off, len);
				if(hipMemcpyAsync((uint8_t*)dst + off, buf[t][b], len, hipMemcpyHostToDevice, st[t]) != hipSuccess || hipEventRecord(ev[t][b], st[t]) != hipSuccess) { good = false; return; }
			}
			if(hipStreamSynchronize(st[t]) != hipSuccess) good = false;
		});
		for(int t = 0; t < T; t++) th[t].join();
		if(!good) (void)hipGetLastError();
		return good;
	}
};
static int upload_raw(h2g_index* ix, const void* src, size_t n, const void** out, size_t pad, Stager* sg) {
	void* p = nullptr;
	const size_t bytes = n + pad;
	HIPCHK(hipMalloc(&p, bytes));
	ix->allocs.push_back(p);
	ix->device_bytes += bytes;
	HIPCHK(hipMemset((uint8_t*)p + n, 0, pad));                       // (the padding only: the rest is about to be written)
	if(n) {
		if(sg && sg->ok && n >= ((size_t)32 << 20)) { if(!sg->copy(p, src, n)) { snprintf(g_err, sizeof g_err, "index upload through the staging buffers failed"); return H2G_ERR_DEVICE; } }
		else HIPCHK(hipMemcpy(p, src, n, hipMemcpyHostToDevice));
	}
	*out = p;
	return H2G_OK;
}
template <typename T>
static int upload(h2g_index* ix, const std::vector<T>& v, const T** out, size_t pad = 64, Stager* sg = nullptr) {
	const void* p = nullptr;
	const int rc = upload_raw(ix, v.data(), v.size() * sizeof(T), &p, pad ? pad : 64, sg);
	*out = (const T*)p;
	return rc;
}

extern "C" void h2g_load_opts_init(h2g_load_opts* o) { o->device = 0; o->load_local = 1; }

static void fill_dgfm(const HostGfm& g, uint32_t minK, DGfm* d) {
	for(int i = 0; i < 5; i++) d->fchr[i] = g.fchr[i];
	d->len = g.p.len; d->gbwtLen = g.p.gbwtLen; d->ftabLim = g.p.linear ? g.p.len : g.p.gbwtLen;
	d->sideGbwtLen = g.p.sideGbwtLen; d->sideGbwtSz = g.p.sideGbwtSz; d->lineRate = g.p.lineRate;
	d->offRate = g.p.offRate; d->offMask = g.p.offMask; d->ftabChars = g.p.ftabChars;
	d->nFrag = g.nFrag; d->nPat = g.nPat; d->nZ = (uint32_t)g.zOffs.size();
	d->zoff = g.zOffs.empty() ? H2G_MAX : g.zOffs[0];
	d->minK = minK; d->linear = g.p.linear;
}

extern "C" h2g_status h2g_index_load(const char* base, const h2g_load_opts* opts, h2g_index** out) {
	if(!base || !out) return H2G_ERR_ARG;
	h2g_load_opts o;
	h2g_load_opts_init(&o);
	if(opts) o = *opts;
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if(e != hipSuccess || ndev <= 0) { snprintf(g_err, sizeof g_err, "no HIP device (libh2g has no CPU path)"); return H2G_ERR_DEVICE; }
	if(o.device < 0 || o.device >= ndev) return H2G_ERR_ARG;
	HIPCHK(hipSetDevice(o.device));
	h2g_index* ix = new h2g_index();
	ix->device = o.device;
	// the local indexes (.5 / .6: 2.1 GB of a human-size index) are packed on a thread of their own, straight from the files, while the global
	// index is read, parsed and uploaded here
	LocalPack lp;
	LocalPlan lpl;                                           // the local indexes' pieces in the mapped files: streamed to the device below, never assembled on the host
	int lrc = 1;                                             // 1: no local files (an index built without them loads for rank / search only)
	std::thread tlocal;
	if(o.load_local) tlocal = std::thread([&]() { try { lrc = plan_local_pack(base, 0, lp, lpl); if(lrc == -1) lrc = 1; } catch(...) { lrc = -2; } });   // (bad_alloc / length_error on a corrupt file must not escape the thread)
	struct Joiner { std::thread& t; ~Joiner() { if(t.joinable()) t.join(); } } joiner{tlocal};      // (every early return below waits for it)
	int rc;
	const bool ltime = getenv("H2G_LOAD_TIMING") != nullptr;
	const auto tl0 = std::chrono::steady_clock::now();
	auto lsec = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - tl0).count(); };
	BigViews bv;                                             // sides, SA sample and reference bases stay in the mapped files until they are on the device
	Stager sg;
	sg.init(o.device);
	try { rc = load_host_index(base, false, ix->host, false, &bv); } catch(...) { rc = -2; }
	const double t_parse = lsec();   // a length read from a corrupt file: an allocation failure is a format error, not std::terminate across the C ABI
	if(rc != 0) { if(tlocal.joinable()) tlocal.join(); delete ix; snprintf(g_err, sizeof g_err, "cannot read index %s", base); return rc == -1 ? H2G_ERR_IO : H2G_ERR_FORMAT; }
	const HostGfm& g = ix->host.g;
	fill_dgfm(g, ix->host.minK, &ix->dg);
	int s = H2G_OK;
	{
		const void *ds_ = nullptr, *do_ = nullptr;
		if((s = upload_raw(ix, bv.sides, bv.sides_n, &ds_, 256, &sg)) || (s = upload_raw(ix, bv.offs, bv.offs_n, &do_, 64, &sg))) { h2g_index_free(ix); return s; }
		ix->dg.sides = (const uint8_t*)ds_; ix->dg.offs = (const uint32_t*)do_;
	}
	if((s = upload(ix, g.ftab, &ix->dg.ftab)) ||
	   (s = upload(ix, g.eftab, &ix->dg.eftab)) ||
	   (s = upload(ix, g.rstarts, &ix->dg.rstarts)) || (s = upload(ix, g.plen, &ix->dg.plen)) ||
	   (s = upload(ix, g.zOffs, &ix->dg.zoffs))) { h2g_index_free(ix); return s; }
	const HostRef& r = ix->host.r;
	ix->dr.nrefs = r.nrefs;
	{
		const void* db_ = nullptr;
		if((s = upload_raw(ix, bv.buf, bv.buf_n, &db_, 64, &sg))) { h2g_index_free(ix); return s; }
		ix->dr.buf = (const uint8_t*)db_;
	}
	const double t_global = lsec();
	if((s = upload(ix, r.rec_start, &ix->dr.rec_start)) ||
	   (s = upload(ix, r.rec_len, &ix->dr.rec_len)) || (s = upload(ix, r.rec_bufoff, &ix->dr.rec_bufoff)) ||
	   (s = upload(ix, r.refRecOffs, &ix->dr.refRecOffs)) || (s = upload(ix, r.refLens, &ix->dr.refLens))) { h2g_index_free(ix); return s; }
	static_assert(sizeof(HostAlt) == sizeof(DAlt), "HostAlt mirrors DAlt");
	ix->dalts.a = nullptr; ix->dalts.n = 0; ix->dalts.maxAltsTried = 16;      // --max-altstried default hisat2.cpp:521
	if(!g.p.linear && !ix->host.alts.empty()) {
		std::vector<uint64_t> packed;                      // the ALTs followed by the haplotype table (pack_alts, h2g_graph.h)
		pack_alts(ix->host.alts, ix->host.hap_left, ix->host.hap_right, ix->host.hap_maxright, ix->host.hap_first, ix->host.hap_ids, packed);
		const uint64_t* da = nullptr;
		if((s = upload(ix, packed, &da))) { h2g_index_free(ix); return s; }
		ix->dalts.a = reinterpret_cast<const DAlt*>(da);
		ix->dalts.n = (uint32_t)ix->host.alts.size();
		std::vector<uint32_t> bk;
		alt_buckets(reinterpret_cast<const DAlt*>(ix->host.alts.data()), ix->dalts.n, bk);
		const uint32_t* dbk = nullptr;
		if(!bk.empty()) { if((s = upload(ix, bk, &dbk))) { h2g_index_free(ix); return s; } ix->dalts.bucket = dbk; ix->dalts.nbucket = (uint32_t)bk.size(); }
		// a --ss index: its splice-site ALTs are graph edges AND known sites of the database (SpliceSiteDB::read(gfm, alts))
		splice_sites_of_alts(reinterpret_cast<const uint32_t*>(ix->host.alts.data()), ix->host.alts.size(), sizeof(HostAlt) / 4, g.rstarts.data(), g.nFrag, g.p.len, ix->alt_sites);
		for(const HostAlt& a : ix->host.alts) if(a.type == 5) ix->dalts.has_splice = 1;
	}
	memset(&ix->dls, 0, sizeof ix->dls);
	if(tlocal.joinable()) tlocal.join();
	const double t_join = lsec();
	if(lrc == -2) { h2g_index_free(ix); snprintf(g_err, sizeof g_err, "cannot read the local indexes of %s", base); return H2G_ERR_FORMAT; }
	if(o.load_local && lrc == 0 && !lp.desc.empty()) {
		while(lp.first.size() <= g.nPat) lp.first.push_back((uint32_t)lp.desc.size());
		const DLocalDesc* dd; const uint8_t* ds = nullptr; const uint16_t* dw = nullptr; const uint32_t* df; const uint32_t* dz;
		{	// the two packed arrays (sides: 128-byte aligned per index, 256 bytes of padding behind; 16-bit words: 64 entries of padding), gathered chunk by chunk out of the files
			const size_t sb = lpl.nsides_tot + 256, wb = (lpl.nwords_tot + 64) * 2;
			void *ps = nullptr, *pw = nullptr;
			HIPCHK(hipMalloc(&ps, sb + 64)); ix->allocs.push_back(ps); ix->device_bytes += sb + 64;
			HIPCHK(hipMalloc(&pw, wb + 64)); ix->allocs.push_back(pw); ix->device_bytes += wb + 64;
			HIPCHK(hipMemset((uint8_t*)ps + sb, 0, 64)); HIPCHK(hipMemset((uint8_t*)pw + wb, 0, 64));
			bool okc = sg.ok;
			if(okc) okc = sg.gather(ps, sb, [&](uint8_t* out_, size_t off_, size_t len_) { local_fill_sides(lpl, out_, off_, len_); }) &&
			              sg.gather(pw, wb, [&](uint8_t* out_, size_t off_, size_t len_) { local_fill_words(lpl, out_, off_, len_); });
			if(!okc) {   // (no staging buffers: through one host chunk at a time)
				std::vector<uint8_t> tmp((size_t)16 << 20);
				for(size_t off_ = 0; off_ < sb; off_ += tmp.size()) { const size_t len_ = sb - off_ < tmp.size() ? sb - off_ : tmp.size(); local_fill_sides(lpl, tmp.data(), off_, len_); HIPCHK(hipMemcpy((uint8_t*)ps + off_, tmp.data(), len_, hipMemcpyHostToDevice)); }
				for(size_t off_ = 0; off_ < wb; off_ += tmp.size()) { const size_t len_ = wb - off_ < tmp.size() ? wb - off_ : tmp.size(); local_fill_words(lpl, tmp.data(), off_, len_); HIPCHK(hipMemcpy((uint8_t*)pw + off_, tmp.data(), len_, hipMemcpyHostToDevice)); }
			}
			ds = (const uint8_t*)ps; dw = (const uint16_t*)pw;
		}
		if((s = upload(ix, lp.desc, &dd)) || (s = upload(ix, lp.first, &df)) || (s = upload(ix, lp.zoffs, &dz))) { h2g_index_free(ix); return s; }
		ix->dls = lp.view(dd, ds, dw, df, dz);
		ix->h_ldesc = lp.desc;
		ix->host.local_first = lp.first;                     // (h2g_local_index_of)
		ix->n_local = (uint32_t)lp.desc.size();
		ix->has_local = true;
		for(const HostAlt& a : ix->host.alts) if(a.type >= 5) ix->has_splice_alts = true;   // ALT_SPLICESITE / ALT_EXON alt.h:38-39
	}
	{   // splice-site probability tables (spliced alignment): 1.4 MB, index-independent
		std::vector<float> d, a1, a2;
		splice_tables(d, a1, a2);
		if((s = upload(ix, d, &ix->d_spl[0])) || (s = upload(ix, a1, &ix->d_spl[1])) || (s = upload(ix, a2, &ix->d_spl[2]))) { h2g_index_free(ix); return s; }
	}
	if(!ix->alt_sites.empty()) { h2g_index* tmp_ = ix; const h2g_status rs_ = h2g_index_set_splice_sites(tmp_, nullptr, 0, 0); if(rs_ != H2G_OK) { h2g_index_free(ix); return rs_; } }
	if(ltime) fprintf(stderr, "index load: parse %.3f s, global arrays on the device %.3f s, local pack joined %.3f s, all %.3f s (%.2f GB)\n", t_parse, t_global, t_join, lsec(), ix->device_bytes / 1e9);
	*out = ix;
	return H2G_OK;
}

extern "C" h2g_status h2g_index_get_info(const h2g_index* ix, h2g_index_info* o) {
	if(!ix || !o) return H2G_ERR_ARG;
	const GfmParams& p = ix->host.g.p;
	o->len = p.len; o->gbwtLen = p.gbwtLen; o->numNodes = p.numNodes; o->lineRate = p.lineRate; o->offRate = p.offRate;
	o->ftabChars = p.ftabChars; o->eftabLen = p.eftabLen; o->linear = p.linear; o->sideSz = p.sideSz;
	o->sideGbwtSz = p.sideGbwtSz; o->sideGbwtLen = p.sideGbwtLen; o->numSides = p.numSides; o->offsLen = p.offsLen;
	o->ftabLen = p.ftabLen; o->nPat = ix->host.g.nPat; o->nFrag = ix->host.g.nFrag; o->nZ = (uint32_t)ix->host.g.zOffs.size();
	o->minK = ix->host.minK; o->nLocal = ix->n_local; o->nRefRecs = (uint32_t)ix->host.r.rec_len.size();
	o->device_bytes = ix->device_bytes;
	return H2G_OK;
}

static h2g_status upload_splice_db(h2g_index* ix) {
	const HostSpliceDB& h = ix->h_ssdb;
	const uint32_t window = ix->dssdb.window;
	ix->dssdb = DSpliceDB();
	ix->dssdb.window = window;
	if(h.fw.empty()) return H2G_OK;
	const void* src[4] = {h.fw.data(), h.bw.data(), h.fw_first.data(), h.bw_first.data()};
	const size_t bytes[4] = {h.fw.size() * sizeof(DSpliceSite), h.bw.size() * sizeof(DSpliceSite), h.fw_first.size() * 4, h.bw_first.size() * 4};
	for(int k = 0; k < 4; k++) {
		if(ix->ssdb_cap[k] < bytes[k]) {
			if(ix->d_ssdb[k]) (void)hipFree(ix->d_ssdb[k]);
			ix->d_ssdb[k] = nullptr; ix->ssdb_cap[k] = 0;
			const size_t cap = bytes[k] * 2 < 4096 ? 4096 : bytes[k] * 2;
			HIPCHK(hipMalloc(&ix->d_ssdb[k], cap));
			ix->ssdb_cap[k] = cap;
		}
		HIPCHK(hipMemcpy(ix->d_ssdb[k], src[k], bytes[k], hipMemcpyHostToDevice));
	}
	ix->dssdb.fw = (const DSpliceSite*)ix->d_ssdb[0]; ix->dssdb.bw = (const DSpliceSite*)ix->d_ssdb[1];
	ix->dssdb.fw_first = (const uint32_t*)ix->d_ssdb[2]; ix->dssdb.bw_first = (const uint32_t*)ix->d_ssdb[3];
	ix->dssdb.n = (uint32_t)h.fw.size();
	return H2G_OK;
}

extern "C" h2g_status h2g_index_set_splice_sites(h2g_index* ix, const h2g_splice_site* sites, size_t n, uint32_t window) {
	if(!ix || (n && !sites)) return H2G_ERR_ARG;
	HIPCHK(hipSetDevice(ix->device));
	HIPCHK(hipDeviceSynchronize());
	ix->dssdb = DSpliceDB();
	ix->dssdb.window = window;
	ix->h_ssdb = HostSpliceDB();
	std::vector<h2g_splice_site> all(ix->alt_sites);      // the index's own sites first: of equal sites the first is kept
	if(n) all.insert(all.end(), sites, sites + n);
	// an intron longer than a splice edit holds (20 bits) can never be placed by the aligner (max_intronlen is capped there): such sites
	// are refused by name instead of being joined with a truncated length
	for(const h2g_splice_site& x : all) if(x.right > x.left && x.right - x.left - 1 > H2G_SPL_MAXLEN) {
		snprintf(g_err, sizeof g_err, "splice site %u:%u-%u: an intron of more than %u bases", x.tidx, x.left, x.right, (unsigned)H2G_SPL_MAXLEN);
		return H2G_ERR_UNSUPPORTED;
	}
	if(all.empty()) return H2G_OK;
	build_splice_db(all.data(), all.size(), ix->host.g.nPat, ix->h_ssdb);
	return upload_splice_db(ix);
}

// The sites met since the last call (or known ones whose smallest read id went down) join the database: a merge into the sorted host
// copy and one upload into buffers that only ever grow — no sort of everything, no free / malloc per wave of reads.  The caller has
// nothing in flight on this index (the waves of the temporary-splice-site scheme are synchronous by construction).
extern "C" h2g_status h2g_index_add_splice_sites(h2g_index* ix, const h2g_splice_site* delta, size_t n) {
	if(!ix || (n && !delta)) return H2G_ERR_ARG;
	if(n == 0) return H2G_OK;
	HIPCHK(hipSetDevice(ix->device));
	for(size_t i = 0; i < n; i++) if(delta[i].right > delta[i].left && delta[i].right - delta[i].left - 1 > H2G_SPL_MAXLEN) {
		snprintf(g_err, sizeof g_err, "splice site %u:%u-%u: an intron of more than %u bases", delta[i].tidx, delta[i].left, delta[i].right, (unsigned)H2G_SPL_MAXLEN);
		return H2G_ERR_UNSUPPORTED;
	}
	HIPCHK(hipDeviceSynchronize());
	// without a database yet (no h2g_index_set_splice_sites before this call) the index's own splice-site ALTs seed it, as they seed every
	// database (SpliceSiteDB::read(gfm, alts), splice_site.cpp:653); the window stays what it was (0 until a set call names one)
	if(ix->h_ssdb.fw.empty() && !ix->alt_sites.empty()) build_splice_db(ix->alt_sites.data(), ix->alt_sites.size(), ix->host.g.nPat, ix->h_ssdb);
	merge_splice_db(ix->h_ssdb, delta, n, ix->host.g.nPat);
	return upload_splice_db(ix);
}

extern "C" void h2g_index_free(h2g_index* ix) {
	if(!ix) return;
	for(void* p : ix->d_ssdb) if(p) (void)hipFree(p);
	for(void* p : ix->allocs) (void)hipFree(p);
	delete ix;
}

// ------------------------------------------------------------------------------------------ synthetic sides
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}

// fills payloads with random symbols; per-side symbol counts go to cnt[side*4 + c]
// (sideSz 64: 6 payload words, occ at +48; sideSz 128: 6.5 payload words, random F/M bits, zero headers, occ at +112)
__global__ void k_synth_fill(uint8_t* sides, uint32_t* cnt, uint64_t nsides, uint64_t seed, uint32_t sideSz) {
	uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
	if(s >= nsides) return;
	uint64_t* w = reinterpret_cast<uint64_t*>(sides + s * sideSz);
	uint32_t c[4] = {0, 0, 0, 0};
	for(int k = 0; k < 6; k++) {
		uint64_t v = splitmix64(seed + s * 16 + k);
		w[k] = v;
		for(int cc = 0; cc < 4; cc++) c[cc] += count_word(v, cc, 32);
	}
	if(sideSz == 128) {
		for(int k = 6; k < 13; k++) w[k] = splitmix64(seed + s * 16 + k);
		w[13] = 0;
		for(int cc = 0; cc < 4; cc++) c[cc] += count_word(w[6], cc, 16);
	}
	for(int cc = 0; cc < 4; cc++) cnt[s * 4 + cc] = c[cc];
}
// exclusive prefix over sides in three passes: per-chunk sums, scan of the chunk sums, write-back
#define SYNTH_CHUNKS 16384
__global__ void k_synth_chunk_sum(const uint32_t* cnt, uint64_t nsides, uint32_t* chunk) {
	uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
	if(t >= SYNTH_CHUNKS) return;
	uint64_t per = (nsides + SYNTH_CHUNKS - 1) / SYNTH_CHUNKS, a = t * per, b = a + per < nsides ? a + per : nsides;
	uint32_t c[4] = {0, 0, 0, 0};
	for(uint64_t s = a; s < b; s++) for(int k = 0; k < 4; k++) c[k] += cnt[s * 4 + k];
	for(int k = 0; k < 4; k++) chunk[t * 4 + k] = c[k];
}
__global__ void k_synth_chunk_scan(uint32_t* chunk, uint32_t* totals) {
	int c = threadIdx.x;
	if(c >= 4) return;
	uint32_t run = 0;
	for(uint32_t t = 0; t < SYNTH_CHUNKS; t++) { uint32_t v = chunk[t * 4 + c]; chunk[t * 4 + c] = run; run += v; }
	totals[c] = run;
}
__global__ void k_synth_write(uint8_t* sides, const uint32_t* cnt, uint64_t nsides, const uint32_t* chunk, uint32_t sideSz) {
	uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
	if(t >= SYNTH_CHUNKS) return;
	uint64_t per = (nsides + SYNTH_CHUNKS - 1) / SYNTH_CHUNKS, a = t * per, b = a + per < nsides ? a + per : nsides;
	uint32_t run[4];
	for(int k = 0; k < 4; k++) run[k] = chunk[t * 4 + k];
	for(uint64_t s = a; s < b; s++) {
		uint32_t* occ = reinterpret_cast<uint32_t*>(sides + s * sideSz + sideSz - 16);
		for(int k = 0; k < 4; k++) { occ[k] = run[k]; run[k] += cnt[s * 4 + k]; }
	}
}

static h2g_status synth_sides(uint64_t num_sides, uint64_t seed, int device, uint32_t sideSz, h2g_index** out) {
	const uint32_t syms = sideSz == 64 ? 192u : H2G_GSIDE_SYMS;
	if(!out || num_sides == 0 || num_sides * (uint64_t)syms >= 0xffffffffull) return H2G_ERR_ARG;
	int ndev = 0;
	if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { snprintf(g_err, sizeof g_err, "no HIP device"); return H2G_ERR_DEVICE; }
	HIPCHK(hipSetDevice(device));
	h2g_index* ix = new h2g_index();
	ix->synthetic = true;
	ix->device = device;
	void *sides = nullptr, *cnt = nullptr, *tot = nullptr;
	const size_t bytes = num_sides * sideSz + 256;
	{   // measurement knob: H2G_SIDES_MTYPE=uncached|finegrained allocates the side array with that memory type
		const char* mt = getenv("H2G_SIDES_MTYPE");
		if(mt && !strcmp(mt, "uncached")) HIPCHK(hipExtMallocWithFlags(&sides, bytes, hipDeviceMallocUncached));
		else if(mt && !strcmp(mt, "finegrained")) HIPCHK(hipExtMallocWithFlags(&sides, bytes, hipDeviceMallocFinegrained));
		else HIPCHK(hipMalloc(&sides, bytes));
	}
	HIPCHK(hipMalloc(&cnt, num_sides * 16));
	HIPCHK(hipMalloc(&tot, 16));
	ix->allocs.push_back(sides);
	ix->device_bytes = num_sides * sideSz;
	hipLaunchKernelGGL(k_synth_fill, dim3((unsigned)((num_sides + 255) / 256)), dim3(256), 0, 0, (uint8_t*)sides, (uint32_t*)cnt, num_sides, seed, sideSz);
	void* chunk = nullptr;
	HIPCHK(hipMalloc(&chunk, SYNTH_CHUNKS * 16));
	hipLaunchKernelGGL(k_synth_chunk_sum, dim3(SYNTH_CHUNKS / 256), dim3(256), 0, 0, (const uint32_t*)cnt, num_sides, (uint32_t*)chunk);
	hipLaunchKernelGGL(k_synth_chunk_scan, dim3(1), dim3(64), 0, 0, (uint32_t*)chunk, (uint32_t*)tot);
	hipLaunchKernelGGL(k_synth_write, dim3(SYNTH_CHUNKS / 256), dim3(256), 0, 0, (uint8_t*)sides, (const uint32_t*)cnt, num_sides, (const uint32_t*)chunk, sideSz);
	uint32_t totals[4];
	HIPCHK(hipMemcpy(totals, tot, 16, hipMemcpyDeviceToHost));
	(void)hipFree(cnt); (void)hipFree(tot); (void)hipFree(chunk);
	GfmParams& p = ix->host.g.p;
	uint32_t len = (uint32_t)(num_sides * syms - 1);
	if(sideSz == 64) p.init(len, len + 1, len + 1, 6, 4, 10, 0, 4);
	else p.init(len - 1000, len + 1, len - 500, 7, 4, 10, 0, 4);   // gbwtLen != len + 1 => graph (gfm.h:139)
	p.numSides = (uint32_t)num_sides;
	ix->host.g.fchr[0] = 0;
	for(int c = 0; c < 4; c++) ix->host.g.fchr[c + 1] = ix->host.g.fchr[c] + totals[c];
	fill_dgfm(ix->host.g, 16, &ix->dg);
	ix->dg.sides = (const uint8_t*)sides;
	ix->dg.nZ = 0;
	*out = ix;
	return H2G_OK;
}
extern "C" h2g_status h2g_index_synth_sides(uint64_t num_sides, uint64_t seed, int device, h2g_index** out) {
	return synth_sides(num_sides, seed, device, 64, out);
}
extern "C" h2g_status h2g_index_synth_graph_sides(uint64_t num_sides, uint64_t seed, int device, h2g_index** out) {
	return synth_sides(num_sides, seed, device, 128, out);
}

// ------------------------------------------------------------------------------------------ stream
// A stream of this library is 2 + H2G_MSTREAMS_MAX HIP streams (the fast launches', the drain launches', one per machine pass in flight) whose kernels must be able to run side by side.  ROCclr maps HIP streams onto
// GPU_MAX_HW_QUEUES hardware queues (4 by default) and streams that share a queue serialise — measured: four machine passes "in flight" on the
// default 4 queues ran one after the other (repeat-structured leg: 127 ms per step against 78 ms with 16 queues; profiles/r05_NOTES.md).  The variable is
// read when the HIP runtime initialises, so it is set when this library is loaded (never overriding the caller's own setting); a process that
// initialises HIP before loading libh2g.so sets it itself (bench.py, hisat2-align-amd and hisat2_amd/api.py do).
__attribute__((constructor)) static void h2g_want_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

extern "C" h2g_status h2g_stream_create(h2g_index* ix, size_t max_reads, size_t max_bases, h2g_stream** out) {
	if(!ix || !out) return H2G_ERR_ARG;
	HIPCHK(hipSetDevice(ix->device));
	h2g_stream* s = new h2g_stream();
	s->ix = ix; s->max_reads = max_reads; s->max_bases = max_bases;
	HIPCHK(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
	for(int k = 0; k < H2G_MSTREAMS_MAX; k++) HIPCHK(hipStreamCreateWithFlags(&s->mst[k], hipStreamNonBlocking));
	HIPCHK(hipStreamCreateWithFlags(&s->dst, hipStreamNonBlocking));
	HIPCHK(hipHostMalloc((void**)&s->h_fast_args, sizeof(FastArgs) * 2 * H2G_NBUF));
	HIPCHK(hipHostMalloc((void**)&s->h_bails, 4 * H2G_NBUF)); for(int k = 0; k < H2G_NBUF; k++) s->h_bails[k] = 0;
	for(int k = 0; k < H2G_NBUF; k++) { HIPCHK(hipEventCreateWithFlags(&s->ev_fast[k], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&s->ev_mach[k], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&s->ev_bails[k], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&s->ev_drain[k], hipEventDisableTiming)); }
	for(int k = 0; k < H2G_FAST_POOLS; k++) HIPCHK(hipEventCreateWithFlags(&s->ev_pool[k], hipEventDisableTiming));
	for(int k = 0; k < 2; k++) HIPCHK(hipEventCreate(&s->ev_dr[k]));
	for(int i = 0; i < 12; i++) HIPCHK(hipEventCreate(&s->ev[i]));
	HIPCHK(hipMalloc((void**)&s->d_counters, H2G_NBUF * H2G_CNT_BLOCK * sizeof(unsigned long long)));
	HIPCHK(hipMemset(s->d_counters, 0, H2G_NBUF * H2G_CNT_BLOCK * sizeof(unsigned long long)));
	s->cnt_cur = s->d_counters;
	if(max_reads) {
		HIPCHK(hipMalloc((void**)&s->d_codes, max_bases + 64));
		HIPCHK(hipMalloc((void**)&s->d_quals, max_bases + 64));
		HIPCHK(hipMalloc((void**)&s->d_offs, (max_reads + 1) * 4));
		HIPCHK(hipMalloc((void**)&s->d_seed, max_reads * 2 * sizeof(h2g_seed_result)));
	}
	memset(&s->last, 0, sizeof s->last);
	{
		auto env = [](const char* k, long d) { const char* e = getenv(k); return e ? atol(e) : d; };
		s->tune.fast = (int)env("H2G_GO_FAST", 1); s->tune.blocks_per_cu = (int)env("H2G_GO_BLOCKS_PER_CU", 0); s->tune.pair_slots = (int)env("H2G_PAIR_SLOTS", 0);
		s->tune.no_second_pass = (int)env("H2G_GO_NO_SECOND_PASS", 0); s->tune.mach_div = (unsigned)env("H2G_MACH_DIV", 0); s->tune.mach_min = (unsigned)env("H2G_MACH_MIN", 4);
		s->tune.dbg_read = env("H2G_GO_DBG_READ", -1);
		s->tune.tail = (int)env("H2G_FAST_TAIL", H2G_DEFAULT_TAIL); s->tune.tail_auto = getenv("H2G_FAST_TAIL") ? 0 : 1; s->tune.align_mate = (int)env("H2G_FAST_AM", H2G_DEFAULT_ALIGN_MATE);
		s->tune.orphan = (int)env("H2G_FAST_ORPHAN", -1); s->tune.drain_grid = (int)env("H2G_DRAIN_GRID", H2G_DRAIN_GRID); s->tune.mate_handover = (int)env("H2G_FAST_MATE_HANDOVER", -1);
		s->tune.mach_total = (unsigned)env("H2G_MACH_TOTAL", H2G_MACH_TOTAL); s->tune.mach_total_auto = getenv("H2G_MACH_TOTAL") ? 0 : 1; s->tune.fast_reserve = (int)env("H2G_FAST_RESERVE", H2G_FAST_RESERVE_DEFAULT);
		{ const long m = env("H2G_MSTREAMS", H2G_MSTREAMS_DEFAULT); s->mstreams = (unsigned)(m < 1 ? 1 : m > H2G_MSTREAMS_MAX ? H2G_MSTREAMS_MAX : m); }
	}
	*out = s;
	return H2G_OK;
}

extern "C" void h2g_stream_free(h2g_stream* s) {
	if(!s) return;
	(void)hipStreamSynchronize(s->st); (void)hipStreamSynchronize(s->dst); for(int k = 0; k < H2G_MSTREAMS_MAX; k++) (void)hipStreamSynchronize(s->mst[k]);
	(void)hipFree(s->d_codes); (void)hipFree(s->d_quals); (void)hipFree(s->d_offs); (void)hipFree(s->d_seed);
	(void)hipFree(s->d_counters); (void)hipFree(s->d_names); (void)hipFree(s->d_name_offs); for(int k = 0; k < 2 * H2G_MSTREAMS_MAX; k++) { (void)hipFree(s->pool[k].ws); (void)hipFree(s->pool[k].gws); (void)hipFree(s->pool[k].sw); (void)hipFree(s->pool[k].sc); }
	for(int k = 0; k < H2G_MSTREAMS_MAX; k++) (void)hipFree(s->d_ovf_list[k]);
	for(int k = 0; k < H2G_NBUF; k++) { (void)hipFree(s->d_bail_list[k]); (void)hipFree(s->d_fast_args[k]); (void)hipFree(s->d_fast_args[H2G_NBUF + k]); (void)hipFree(s->d_orphans[k]); (void)hipEventDestroy(s->ev_fast[k]); (void)hipEventDestroy(s->ev_mach[k]); (void)hipEventDestroy(s->ev_bails[k]); (void)hipEventDestroy(s->ev_drain[k]); }
	for(int k = 0; k < H2G_FAST_POOLS; k++) { (void)hipFree(s->d_fast_slots[k]); (void)hipEventDestroy(s->ev_pool[k]); }
	(void)hipFree(s->d_drain_slots); (void)hipFree(s->d_drain_sc);
	for(int k = 0; k < 2; k++) (void)hipEventDestroy(s->ev_dr[k]);
	(void)hipFree(s->d_fast_gws); (void)hipFree(s->d_fast_sc); (void)hipFree(s->d_sw_ws); (void)hipFree(s->d_sw_states); (void)hipFree(s->dbg_buf);
	(void)hipFree(s->d_rout); (void)hipFree(s->d_aln); (void)hipFree(s->d_codes2); (void)hipFree(s->d_offs2); (void)hipFree(s->d_quals2);
	(void)hipFree(s->d_names2); (void)hipFree(s->d_name_offs2); (void)hipFree(s->d_pout); (void)hipFree(s->d_paln[0]); (void)hipFree(s->d_paln[1]); (void)hipFree(s->d_paln_ovf); (void)hipFree(s->d_ledits); (void)hipFree(s->d_ledits_cur); (void)hipFree(s->d_warm_cnt);
	for(unsigned b = 0; b < H2G_MAX_BATCHES; b++) if(b != s->cur_batch) {
		BatchCtx& B = s->parked[b];
		(void)hipFree(B.d_codes); (void)hipFree(B.d_offs); (void)hipFree(B.d_quals); (void)hipFree(B.d_names); (void)hipFree(B.d_name_offs); (void)hipFree(B.d_codes2); (void)hipFree(B.d_offs2);
		(void)hipFree(B.d_quals2); (void)hipFree(B.d_names2); (void)hipFree(B.d_name_offs2); (void)hipFree(B.d_rout); (void)hipFree(B.d_aln); (void)hipFree(B.d_pout); (void)hipFree(B.d_paln[0]); (void)hipFree(B.d_paln[1]);
	}
	for(int i = 0; i < 4; i++) (void)hipFree(s->d_tmp[i]);
	for(int i = 0; i < 12; i++) (void)hipEventDestroy(s->ev[i]);
	(void)hipStreamDestroy(s->st); (void)hipStreamDestroy(s->dst); for(int k = 0; k < H2G_MSTREAMS_MAX; k++) (void)hipStreamDestroy(s->mst[k]); (void)hipHostFree(s->h_bails); (void)hipHostFree(s->h_fast_args);
	delete s;
}

// both streams of a batch context (the second one only ever holds the machine pass behind a fast pass)
static hipError_t sync_all(h2g_stream* s) {
	hipError_t e = hipStreamSynchronize(s->st);
	if(e == hipSuccess && s->dst_busy) { e = hipStreamSynchronize(s->dst); s->dst_busy = false; }
	if(e == hipSuccess && s->st2_busy) {
		for(int k = 0; k < H2G_MSTREAMS_MAX && e == hipSuccess; k++) e = hipStreamSynchronize(s->mst[k]);
		s->st2_busy = false;
	}
	return e;
}
extern "C" void* h2g_stream_hip(h2g_stream* s) { return s ? (void*)s->st : nullptr; }
extern "C" h2g_status h2g_stream_sync(h2g_stream* s) {
	if(!s) return H2G_ERR_ARG;
	HIPCHK(sync_all(s));
	return H2G_OK;
}

// Selects resident batch k: the h2g_set_* calls that follow fill it, the runs that follow work on it and write its rows, the fetches read them.  Nothing waits: runs queued
// over other batches go on.
extern "C" h2g_status h2g_stream_select_batch(h2g_stream* s, unsigned k) {
	if(!s || k >= H2G_MAX_BATCHES) return H2G_ERR_ARG;
	if(k == s->cur_batch) return H2G_OK;
	HIPCHK(hipSetDevice(s->ix->device));
	BatchCtx& out = s->parked[s->cur_batch];
#define X(F) out.F = s->F;
	H2G_BATCH_FIELDS(X)
#undef X
	out.d_paln[0] = s->d_paln[0]; out.d_paln[1] = s->d_paln[1];
	BatchCtx& in = s->parked[k];
	if(!in.d_codes && s->max_reads) {      // first use: the read buffers h2g_stream_create gives batch 0
		HIPCHK(hipMalloc((void**)&in.d_codes, s->max_bases + 64));
		HIPCHK(hipMalloc((void**)&in.d_quals, s->max_bases + 64));
		HIPCHK(hipMalloc((void**)&in.d_offs, (s->max_reads + 1) * 4));
	}
	// ... and result rows as large as the rows of the batch that is leaving (the option set a streaming caller runs every batch with): a batch's first run then
	// allocates nothing — an allocation of gigabytes in the middle of a queue of runs waits for all of them
	if(!in.d_pout && out.d_pout && out.paln_alloc) {
		HIPCHK(hipMalloc((void**)&in.d_pout, s->max_reads * sizeof(PairOut)));
		for(int m = 0; m < 2; m++) HIPCHK(hipMalloc((void**)&in.d_paln[m], out.paln_alloc * sizeof(h2g_alnres)));
		in.paln_alloc = out.paln_alloc; in.pair_slots = out.pair_slots;
		// (written once here: the first kernel to store into fresh device memory pays for its mapping — measured in bench.py's loop, lease F: 20 timed steps over 10 batches
		// of which 5 had never been run took 18.5 ms each, 60 steps 12.1)
		HIPCHK(hipMemsetAsync(in.d_pout, 0, s->max_reads * sizeof(PairOut), s->st));
		for(int m = 0; m < 2; m++) HIPCHK(hipMemsetAsync(in.d_paln[m], 0, out.paln_alloc * sizeof(h2g_alnres), s->st));
	}
	if(!in.d_rout && out.d_rout && out.aln_alloc) {
		HIPCHK(hipMalloc((void**)&in.d_rout, s->max_reads * sizeof(ReadOut)));
		HIPCHK(hipMalloc((void**)&in.d_aln, out.aln_alloc * sizeof(h2g_alnres)));
		in.aln_alloc = out.aln_alloc; in.aln_slots = out.aln_slots;
		HIPCHK(hipMemsetAsync(in.d_rout, 0, s->max_reads * sizeof(ReadOut), s->st));
		HIPCHK(hipMemsetAsync(in.d_aln, 0, out.aln_alloc * sizeof(h2g_alnres), s->st));
	}
#define X(F) s->F = in.F;
	H2G_BATCH_FIELDS(X)
#undef X
	s->d_paln[0] = in.d_paln[0]; s->d_paln[1] = in.d_paln[1];
	in = BatchCtx();
	s->cur_batch = k;
	return H2G_OK;
}

static int tmp_buf(h2g_stream* s, int slot, size_t bytes, void** out) {
	if(s->tmp_sz[slot] < bytes) {
		(void)hipFree(s->d_tmp[slot]);
		s->d_tmp[slot] = nullptr; s->tmp_sz[slot] = 0;
		HIPCHK(hipMalloc(&s->d_tmp[slot], bytes + 256));
		s->tmp_sz[slot] = bytes;
	}
	*out = s->d_tmp[slot];
	return H2G_OK;
}

extern "C" h2g_status h2g_set_reads(h2g_stream* s, const uint8_t* codes, const uint32_t* offs, const char* quals, size_t n) {
	if(s) HIPCHK(sync_all(s));        // (a machine pass of the previous batch may still read the buffers this call replaces)
	if(!s || !codes || !offs || n > s->max_reads) return H2G_ERR_ARG;
	size_t nb = offs[n];
	if(nb > s->max_bases) return H2G_ERR_ARG;
	uint32_t mx = 0;
	for(size_t i = 0; i < n; i++) { const uint32_t l = offs[i + 1] - offs[i]; if(l > mx) mx = l; }
	s->max_read_len = mx;
	HIPCHK(hipMemcpyAsync(s->d_codes, codes, nb, hipMemcpyHostToDevice, s->st));
	HIPCHK(hipMemcpyAsync(s->d_offs, offs, (n + 1) * 4, hipMemcpyHostToDevice, s->st));
	s->has_quals = quals != nullptr;
	if(quals) HIPCHK(hipMemcpyAsync(s->d_quals, quals, nb, hipMemcpyHostToDevice, s->st));
	HIPCHK(sync_all(s));
	s->n_reads = n;
	s->has_names = false;
	s->has_mates = false;
	s->ledits_touched = 0;            // (long-edit lists of the batch this one replaces are nobody's any more)
	return H2G_OK;
}

static DReads dreads(const h2g_stream* s) {
	DReads r;
	r.codes = s->d_codes; r.offs = s->d_offs; r.quals = s->has_quals ? s->d_quals : nullptr; r.n = (uint32_t)s->n_reads;
	return r;
}

static unsigned grid_for(size_t n, unsigned block) {
	size_t g = (n + block - 1) / block;
	const size_t cap = 256 * 16;   // >= 8 resident blocks per CU plus slack; grid-stride above that
	return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------------------------------ rank kernels
// variant 0: one lane per query, 4 x dwordx4 per side
// The row of a synthetic query: the high word of the hash scaled into [0, gbwtLen) — ONE v_mul_hi.  Through round 6 this was `h % gbwtLen`: a 64-bit modulo, ~45 vector and ~55 scalar
// instructions per query next to the rank's own ~120 — the micro-benchmark was measuring its own generator (issue time 3.3 of its 5.8 ms: that, not the request pattern, was the gap to the
// chain kernel's 3.5 TB/s).  tests/rank_synth_check.py draws the same rows on the host.
__device__ __forceinline__ uint32_t synth_row(uint64_t h, uint32_t n) { return (uint32_t)(((h >> 32) * (uint64_t)n) >> 32); }
__global__ __launch_bounds__(256) void k_rank_v0(DGfm g, const uint32_t* rows, const uint8_t* cs, uint32_t* out,
                                                 size_t n, uint64_t seed, int synth)
{
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
		uint32_t row; int c;
		if(synth) { uint64_t h = splitmix64(seed + i); row = synth_row(h, g.gbwtLen); c = (int)((h >> 40) & 3); }
		else { row = rows[i]; c = cs[i]; }
		out[i] = rank64(g, row, c);
	}
}

__global__ __launch_bounds__(512) void k_rank_v0_512(DGfm g, const uint32_t* rows, const uint8_t* cs, uint32_t* out, size_t n, uint64_t seed, int synth)
{
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
		uint32_t row; int c;
		if(synth) { uint64_t h = splitmix64(seed + i); row = synth_row(h, g.gbwtLen); c = (int)((h >> 40) & 3); }
		else { row = rows[i]; c = cs[i]; }
		out[i] = rank64(g, row, c);
	}
}

// variant 10: variant 0 without the OUTPUT STREAM (round 6).  In the aligner a rank feeds the next step of its own chain; the 4 bytes per query the micro-benchmark writes
// are its own artefact (2^28 queries: 1 GB of stores next to 17 GB of side lines).  Here every result goes into the run's checksum (the k_checksum function of the full
// output, so the two variants must agree) and every 256th is stored — the samples the comparison with the CPU's mapLF reads.
__global__ __launch_bounds__(256) void k_rank_v0_sampled(DGfm g, uint32_t* out, size_t n, uint64_t seed, unsigned long long* sum)
{
	size_t stride = (size_t)gridDim.x * blockDim.x;
	unsigned long long acc = 0;
	for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
		const uint64_t h = splitmix64(seed + i);
		const uint32_t r = rank64(g, synth_row(h, g.gbwtLen), (int)((h >> 40) & 3));
		if((i & 255) == 0) out[i] = r;
		acc += (unsigned long long)r * (unsigned long long)((i & 1023) + 1);
	}
	for(int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
	if((threadIdx.x & 63) == 0) atomicAdd(sum, acc);
}

// variant 1: four lanes per side — one dwordx4 per lane => one fully coalesced 64 B request per query;
// lanes 0-2 popcount two payload words each, lane 3 holds the four Occ words; 2-step butterfly reduce.
template <int UNROLL>
__global__ __launch_bounds__(256) void k_rank_v1(DGfm g, const uint32_t* rows, const uint8_t* cs, uint32_t* out,
                                                 size_t n, uint64_t seed, int synth)
{
	const unsigned lane = threadIdx.x & 63, sub = lane & 3;
	const size_t ngroups = ((size_t)gridDim.x * blockDim.x) >> 2;
	const size_t gid = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 2;
	for(size_t base = gid * UNROLL; base < n; base += ngroups * UNROLL) {
		uint32_t row[UNROLL], sideNum[UNROLL], charOff[UNROLL]; int c[UNROLL]; uint4 v[UNROLL];
#pragma unroll
		for(int u = 0; u < UNROLL; u++) {
			size_t i = base + u;
			bool ok = i < n;
			if(synth) { uint64_t h = splitmix64(seed + (ok ? i : 0)); row[u] = synth_row(h, g.gbwtLen); c[u] = (int)((h >> 40) & 3); }
			else { row[u] = ok ? rows[i] : 0; c[u] = ok ? cs[i] : 0; }
			sideNum[u] = row[u] / 192u; charOff[u] = row[u] - sideNum[u] * 192u;
			v[u] = reinterpret_cast<const uint4*>(g.sides + (size_t)sideNum[u] * 64)[sub];
		}
#pragma unroll
		for(int u = 0; u < UNROLL; u++) {
			uint64_t w0 = v[u].x | ((uint64_t)v[u].y << 32), w1 = v[u].z | ((uint64_t)v[u].w << 32);
			int n0 = (int)charOff[u] - 64 * (int)sub;
			uint32_t cnt = sub < 3 ? count_word(w0, c[u], n0) + count_word(w1, c[u], n0 - 32) : 0u;
			uint32_t occsel = c[u] == 0 ? v[u].x : c[u] == 1 ? v[u].y : c[u] == 2 ? v[u].z : v[u].w;
			uint32_t part = sub == 3 ? occsel : cnt;
			part += __shfl_xor(part, 1);
			part += __shfl_xor(part, 2);
			if(c[u] == 0 && g.nZ) {
				uint32_t zs = g.zoff / 192u, zc = g.zoff - zs * 192u;
				if(zs == sideNum[u] && zc < charOff[u]) part--;
			}
			if(sub == 0 && base + u < n) out[base + u] = part + (c[u] == 0 ? g.fchr[0] : c[u] == 1 ? g.fchr[1] : c[u] == 2 ? g.fchr[2] : g.fchr[3]);
		}
	}
}

// variant 2: eight lanes per side, one u64 per lane
template <int UNROLL>
__global__ __launch_bounds__(256) void k_rank_v2(DGfm g, const uint32_t* rows, const uint8_t* cs, uint32_t* out,
                                                 size_t n, uint64_t seed, int synth)
{
	const unsigned lane = threadIdx.x & 63, sub = lane & 7;
	const size_t ngroups = ((size_t)gridDim.x * blockDim.x) >> 3;
	const size_t gid = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 3;
	for(size_t base = gid * UNROLL; base < n; base += ngroups * UNROLL) {
		uint32_t row[UNROLL], sideNum[UNROLL], charOff[UNROLL]; int c[UNROLL]; uint64_t w[UNROLL];
#pragma unroll
		for(int u = 0; u < UNROLL; u++) {
			size_t i = base + u;
			bool ok = i < n;
			if(synth) { uint64_t h = splitmix64(seed + (ok ? i : 0)); row[u] = synth_row(h, g.gbwtLen); c[u] = (int)((h >> 40) & 3); }
			else { row[u] = ok ? rows[i] : 0; c[u] = ok ? cs[i] : 0; }
			sideNum[u] = row[u] / 192u; charOff[u] = row[u] - sideNum[u] * 192u;
			w[u] = reinterpret_cast<const uint64_t*>(g.sides + (size_t)sideNum[u] * 64)[sub];
		}
#pragma unroll
		for(int u = 0; u < UNROLL; u++) {
			uint32_t cnt = sub < 6 ? count_word(w[u], c[u], (int)charOff[u] - 32 * (int)sub) : 0u;
			uint32_t occ = (sub == 6 + (unsigned)(c[u] >> 1)) ? (uint32_t)(w[u] >> ((c[u] & 1) * 32)) : 0u;
			uint32_t part = cnt + occ;
			part += __shfl_xor(part, 1);
			part += __shfl_xor(part, 2);
			part += __shfl_xor(part, 4);
			if(c[u] == 0 && g.nZ) {
				uint32_t zs = g.zoff / 192u, zc = g.zoff - zs * 192u;
				if(zs == sideNum[u] && zc < charOff[u]) part--;
			}
			if(sub == 0 && base + u < n) out[base + u] = part + (c[u] == 0 ? g.fchr[0] : c[u] == 1 ? g.fchr[1] : c[u] == 2 ? g.fchr[2] : g.fchr[3]);
		}
	}
}

// ---- measurement-only variants (same results as variant 0; they probe the memory system) ----
// 3: also loads the partner half of the 128 B line (is the fetch unit 64 B or 128 B?)
// 4: nontemporal loads (no L2/MALL allocation)       5: two independent queries in flight per lane
template <int MODE>
__global__ __launch_bounds__(256) void k_rank_exp(DGfm g, uint32_t* out, size_t n, uint64_t seed) {
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride * (MODE == 5 ? 2 : 1)) {
		uint64_t h = splitmix64(seed + i);
		uint32_t row = synth_row(h, g.gbwtLen); int c = (int)((h >> 40) & 3);
		if(MODE == 3) {
			uint32_t sideNum = row / 192u;
			const uint4* q = reinterpret_cast<const uint4*>(g.sides + (size_t)(sideNum ^ 1u) * 64);
			uint4 x = q[0];
			asm volatile("" :: "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
			out[i] = rank64(g, row, c);
		} else if(MODE == 4) {
			uint32_t sideNum = row / 192u, charOff = row - sideNum * 192u;
			const uint64_t* q = reinterpret_cast<const uint64_t*>(g.sides + (size_t)sideNum * 64);
			Side64 s;
#pragma unroll
			for(int k = 0; k < 8; k++) s.w[k] = __builtin_nontemporal_load(q + k);
			out[i] = rank_in_side64(g, s, sideNum, charOff, c);
		} else {
			size_t j = i + stride;
			uint64_t h2 = splitmix64(seed + (j < n ? j : i));
			uint32_t row2 = synth_row(h2, g.gbwtLen); int c2 = (int)((h2 >> 40) & 3);
			uint32_t s1 = row / 192u, s2 = row2 / 192u;
			Side64 a = load_side64(g.sides + (size_t)s1 * 64), b = load_side64(g.sides + (size_t)s2 * 64);
			out[i] = rank_in_side64(g, a, s1, row - s1 * 192u, c);
			if(j < n) out[j] = rank_in_side64(g, b, s2, row2 - s2 * 192u, c2);
		}
	}
}

// ---- graph sides (128 B = one L2 line) --------------------------------------------------------------------
// g0: one lane per query, 8 x dwordx4
__global__ __launch_bounds__(256) void k_rank_g0(DGfm g, const uint32_t* rows, const uint8_t* cs, uint32_t* out,
                                                 size_t n, uint64_t seed, int synth)
{
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
		uint32_t row; int c;
		if(synth) { uint64_t h = splitmix64(seed + i); row = synth_row(h, g.gbwtLen); c = (int)((h >> 40) & 3); }
		else { row = rows[i]; c = cs[i]; }
		out[i] = rank128(g, row, c);
	}
}
// g1: eight lanes per side, one dwordx4 each => one fully coalesced 128 B request per query.  Lanes 0..3 hold
// the symbols (lane 3 only 16 of them), lane 7 the Occ words; 3-step butterfly reduce within the 8-lane group.
template <int UNROLL>
__global__ __launch_bounds__(256) void k_rank_g1(DGfm g, const uint32_t* rows, const uint8_t* cs, uint32_t* out,
                                                 size_t n, uint64_t seed, int synth)
{
	const uint32_t lane = threadIdx.x & 63, sub = lane & 7, grp = lane >> 3;
	const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	for(size_t base = wave * 8 * UNROLL; base < n; base += nwaves * 8 * UNROLL) {
		uint4 v[UNROLL]; uint32_t off[UNROLL], sn[UNROLL]; int cc[UNROLL]; bool ok[UNROLL];
#pragma unroll
		for(int u = 0; u < UNROLL; u++) {
			const size_t i = base + (size_t)u * 8 + grp;
			ok[u] = i < n;
			uint32_t row = 0; int c = 0;
			if(ok[u]) {
				if(synth) { uint64_t h = splitmix64(seed + i); row = synth_row(h, g.gbwtLen); c = (int)((h >> 40) & 3); }
				else { row = rows[i]; c = cs[i]; }
			}
			sn[u] = row / H2G_GSIDE_SYMS; off[u] = row - sn[u] * H2G_GSIDE_SYMS; cc[u] = c;
			v[u] = reinterpret_cast<const uint4*>(g.sides + (size_t)sn[u] * 128)[sub];
		}
#pragma unroll
		for(int u = 0; u < UNROLL; u++) {
			const uint64_t w0 = v[u].x | ((uint64_t)v[u].y << 32), w1 = v[u].z | ((uint64_t)v[u].w << 32);
			const int c = cc[u];
			uint32_t cnt = 0;
			if(sub < 3) cnt = count_word(w0, c, (int)off[u] - 64 * (int)sub) + count_word(w1, c, (int)off[u] - 64 * (int)sub - 32);
			else if(sub == 3) cnt = count_word(w0 & 0xffffffffull, c, (int)off[u] - 192 > 16 ? 16 : (int)off[u] - 192);
			else if(sub == 7) {
				const uint64_t ow = (c & 2) ? w1 : w0;
				cnt = (c & 1) ? (uint32_t)(ow >> 32) : (uint32_t)ow;
			}
			cnt += __shfl_xor(cnt, 1); cnt += __shfl_xor(cnt, 2); cnt += __shfl_xor(cnt, 4);
			if(sub == 0 && ok[u]) {
				if(c == 0 && g.nZ) {
					const uint32_t zs = g.zoff / H2G_GSIDE_SYMS, zc = g.zoff - zs * H2G_GSIDE_SYMS;
					if(zs == sn[u] && zc < off[u]) cnt--;
				}
				const uint32_t fc = c == 0 ? g.fchr[0] : c == 1 ? g.fchr[1] : c == 2 ? g.fchr[2] : g.fchr[3];
				out[base + (size_t)u * 8 + grp] = cnt + fc;
			}
		}
	}
}

__global__ void k_checksum(const uint32_t* v, size_t n, unsigned long long* out) {
	unsigned long long acc = 0;
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride)
		acc += (unsigned long long)v[i] * (unsigned long long)((i & 1023) + 1);
	for(int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
	if((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

static int launch_rank(h2g_stream* s, const uint32_t* d_rows, const uint8_t* d_cs, uint32_t* d_out, size_t n,
                       uint64_t seed, int synth, int variant, int repeats, float* ms)
{
	const DGfm& g = s->ix->dg;
	const bool graph = !g.linear && g.lineRate == 7;
	if(!graph && (!g.linear || g.lineRate != 6)) { snprintf(g_err, sizeof g_err, "rank kernels: 64 B linear or 128 B graph sides only"); return H2G_ERR_UNSUPPORTED; }
	if(graph && g.nZ > 1 && variant == 1) variant = 0;   // the cooperative kernel handles a single '$' row
	if(repeats < 1) repeats = 1;
	HIPCHK(hipEventRecord(s->ev[0], s->st));
	for(int r = 0; r < repeats; r++) {
		if(graph) {
			if(variant == 0) hipLaunchKernelGGL(k_rank_g0, dim3(grid_for(n, 256)), dim3(256), 0, s->st, g, d_rows, d_cs, d_out, n, seed, synth);
			else if(variant == 1) hipLaunchKernelGGL(k_rank_g1<4>, dim3(grid_for(n, 256)), dim3(256), 0, s->st, g, d_rows, d_cs, d_out, n, seed, synth);
			else return H2G_ERR_ARG;
		} else if(variant == 0) {
			hipLaunchKernelGGL(k_rank_v0, dim3(grid_for(n, 256)), dim3(256), 0, s->st, g, d_rows, d_cs, d_out, n, seed, synth);
		} else if(variant == 1) {
			hipLaunchKernelGGL(k_rank_v1<4>, dim3(grid_for(n, 256)), dim3(256), 0, s->st, g, d_rows, d_cs, d_out, n, seed, synth);
		} else if(variant == 2) {
			hipLaunchKernelGGL(k_rank_v2<4>, dim3(grid_for(n, 256)), dim3(256), 0, s->st, g, d_rows, d_cs, d_out, n, seed, synth);
		} else if(variant == 3 && synth) {
			hipLaunchKernelGGL(k_rank_exp<3>, dim3(grid_for(n, 256)), dim3(256), 0, s->st, g, d_out, n, seed);
		} else if(variant == 4 && synth) {
			hipLaunchKernelGGL(k_rank_exp<4>, dim3(grid_for(n, 256)), dim3(256), 0, s->st, g, d_out, n, seed);
		} else if(variant == 5 && synth) {
			hipLaunchKernelGGL(k_rank_exp<5>, dim3(grid_for(n, 256)), dim3(256), 0, s->st, g, d_out, n, seed);
		} else if(variant == 10 && synth) {
			HIPCHK(hipMemsetAsync(s->d_counters + 6, 0, 8, s->st));
			hipLaunchKernelGGL(k_rank_v0_sampled, dim3(grid_for(n, 256)), dim3(256), 0, s->st, g, d_out, n, seed, s->d_counters + 6);
		} else if(variant == 11 || variant == 12) {
			// variant 0's loop in the chain kernel's geometry: ONE 512-thread workgroup per CU (11), two (12)
			hipLaunchKernelGGL(k_rank_v0_512, dim3(variant == 11 ? 256u : 512u), dim3(512), 0, s->st, g, d_rows, d_cs, d_out, n, seed, synth);
		} else if(variant >= 6 && variant <= 9) {
			// variant 0's kernel at a fraction of the chip's occupancy (measurement: the chain kernel sustains more random 64 B requests per second at one 512-thread
			// workgroup per CU than at full occupancy — profiles/r05_NOTES.md): 6: 16 waves per CU, 7: 8, 8: 4, 9: 2
			const unsigned blocks = variant == 6 ? 1024u : variant == 7 ? 512u : variant == 8 ? 256u : 128u;
			hipLaunchKernelGGL(k_rank_v0, dim3(blocks), dim3(256), 0, s->st, g, d_rows, d_cs, d_out, n, seed, synth);
		} else return H2G_ERR_ARG;
	}
	HIPCHK(hipEventRecord(s->ev[1], s->st));
	HIPCHK(sync_all(s));
	HIPCHK(hipGetLastError());
	float t = 0;
	HIPCHK(hipEventElapsedTime(&t, s->ev[0], s->ev[1]));
	s->last.ms_rank = t / repeats;
	if(ms) *ms = t / repeats;
	return H2G_OK;
}

extern "C" h2g_status h2g_rank_bench(h2g_stream* s, const uint32_t* rows, const uint8_t* cs, size_t n, uint32_t* out,
                                     int variant, int device_ptrs, int repeats, float* kernel_ms)
{
	if(!s || !rows || !cs || !out || n == 0) return H2G_ERR_ARG;
	HIPCHK(hipSetDevice(s->ix->device));
	if(device_ptrs) return launch_rank(s, rows, cs, out, n, 0, 0, variant, repeats, kernel_ms);
	void *dr, *dc, *dout;
	int rc;
	if((rc = tmp_buf(s, 0, n * 4, &dr)) || (rc = tmp_buf(s, 1, n, &dc)) || (rc = tmp_buf(s, 2, n * 4, &dout))) return rc;
	HIPCHK(hipMemcpyAsync(dr, rows, n * 4, hipMemcpyHostToDevice, s->st));
	HIPCHK(hipMemcpyAsync(dc, cs, n, hipMemcpyHostToDevice, s->st));
	rc = launch_rank(s, (const uint32_t*)dr, (const uint8_t*)dc, (uint32_t*)dout, n, 0, 0, variant, repeats, kernel_ms);
	if(rc) return rc;
	HIPCHK(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
	return H2G_OK;
}

extern "C" h2g_status h2g_rank_bench_synth(h2g_stream* s, size_t n, uint64_t seed, int variant, int repeats,
                                           float* kernel_ms, uint64_t* checksum)
{
	if(!s || n == 0) return H2G_ERR_ARG;
	HIPCHK(hipSetDevice(s->ix->device));
	void* dout;
	int rc;
	if((rc = tmp_buf(s, 2, n * 4, &dout))) return rc;
	rc = launch_rank(s, nullptr, nullptr, (uint32_t*)dout, n, seed, 1, variant, repeats, kernel_ms);
	if(rc) return rc;
	if(checksum) {
		HIPCHK(hipMemsetAsync(s->d_counters + 7, 0, 8, s->st));
		if(variant != 10) hipLaunchKernelGGL(k_checksum, dim3(1024), dim3(256), 0, s->st, (const uint32_t*)dout, n, s->d_counters + 7);
		unsigned long long v = 0;
		HIPCHK(hipMemcpyAsync(&v, s->d_counters + (variant == 10 ? 6 : 7), 8, hipMemcpyDeviceToHost, s->st));   // (variant 10 sums on the fly: the last repeat's)
		HIPCHK(sync_all(s));
		*checksum = v;
	}
	return H2G_OK;
}

// ---- chains of dependent rank queries (measurement only: tools/chain_bench.py) ----------------------------------------
// What the search loops of go() are made of is not a stream of independent rank queries (k_rank_*) but CHAINS: the row of step i + 1 comes out of
// the rank of step i.  This kernel runs C independent chains per lane — chain k of lane t starts at splitmix64(seed + t * C + k) and walks
// row' = splitmix64(rank << 32 | row) % gbwtLen — so that one can measure, at the occupancy of the compact-state pass (512-thread workgroups, one per
// CU: `lds_bytes` of dynamic LDS bound them the same way the pass's staging area does), what several chains in flight per lane are worth against the
// same number of chains spread over more lanes.  The sum of out[] depends only on (seed, number of chains, steps).
template <int C, bool GRAPH>
__global__ __launch_bounds__(512) void k_rank_chain(DGfm g, uint64_t seed, int steps, size_t nlanes, unsigned long long* sum)
{
	extern __shared__ uint32_t chain_pad[];
	const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	unsigned long long acc = 0;
	if(t < nlanes) {
		uint32_t row[C]; int c[C];
#pragma unroll
		for(int k = 0; k < C; k++) { const uint64_t h = splitmix64(seed + t * C + k); row[k] = (uint32_t)(h % g.gbwtLen); c[k] = (int)((h >> 40) & 3); }
		for(int i = 0; i < steps; i++) {
			// every chain's side is requested before any of them is counted (the counting has data-dependent branches: left to itself the compiler
			// finishes chain k before it issues chain k + 1's loads)
			uint32_t r[C], sn[C], co[C];
			if constexpr(GRAPH) {
				Side128 sd[C];
#pragma unroll
				for(int k = 0; k < C; k++) { sn[k] = row[k] / DGfm::SYMS; co[k] = row[k] - sn[k] * DGfm::SYMS; sd[k] = load_side128(g.sides + (size_t)sn[k] * 128); }
#pragma unroll
				for(int k = 0; k < C; k++) r[k] = rank_in_side128(g, sd[k], sn[k], co[k], c[k]);
			} else {
				Side64 sd[C];
#pragma unroll
				for(int k = 0; k < C; k++) { sn[k] = row[k] / 192u; co[k] = row[k] - sn[k] * 192u; sd[k] = load_side64(g.sides + (size_t)sn[k] * 64); }
#pragma unroll
				for(int k = 0; k < C; k++) r[k] = rank_in_side64(g, sd[k], sn[k], co[k], c[k]);
			}
#pragma unroll
			for(int k = 0; k < C; k++) {
				const uint64_t h = splitmix64(((uint64_t)r[k] << 32) | row[k]);
				row[k] = (uint32_t)(h % g.gbwtLen); c[k] = (int)((h >> 40) & 3);
				acc += r[k];
			}
		}
	}
	if(threadIdx.x == 0 && steps < 0) chain_pad[0] = (uint32_t)acc;   // (keeps the dynamic allocation alive; never taken)
	for(int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
	if((threadIdx.x & 63) == 0 && acc) atomicAdd(sum, acc);
}

// nchains chains of `steps` dependent rank queries, chains_per_lane (1, 2, 4 or 8) of them in each lane, workgroups of `block` threads (64..512) with
// `lds_bytes` of dynamic LDS each.  *checksum = the sum of all rank results (the same for every chains_per_lane at equal seed, nchains and steps).
// out[j] = result j * stride of the LAST h2g_rank_bench_synth run of this stream (its device buffer is still there): what a sampled comparison with the
// CPU's mapLF needs of a 2^28-query run
extern "C" h2g_status h2g_rank_bench_synth_sample(h2g_stream* s, size_t stride, size_t nsample, uint32_t* out) {
	if(!s || !out || stride == 0 || nsample == 0 || !s->d_tmp[2] || s->tmp_sz[2] < ((nsample - 1) * stride + 1) * 4) return H2G_ERR_ARG;
	HIPCHK(hipSetDevice(s->ix->device));
	HIPCHK(sync_all(s));
	HIPCHK(hipMemcpy2D(out, 4, s->d_tmp[2], stride * 4, 4, nsample, hipMemcpyDeviceToHost));
	return H2G_OK;
}
extern "C" H2G_EXPORT h2g_status h2g_rank_chain_bench(h2g_stream* s, size_t nchains, int chains_per_lane, int steps, int block, int lds_bytes, uint64_t seed,
                                                      int repeats, float* kernel_ms, uint64_t* checksum)
{
	if(!s || nchains == 0 || steps < 1 || block < 64 || block > 512 || (block & 63) || lds_bytes < 0 || lds_bytes > 160 * 1024) return H2G_ERR_ARG;
	if(chains_per_lane != 1 && chains_per_lane != 2 && chains_per_lane != 4 && chains_per_lane != 8) return H2G_ERR_ARG;
	if(nchains % (size_t)chains_per_lane) return H2G_ERR_ARG;
	const DGfm& g = s->ix->dg;
	const bool graph = !g.linear && g.lineRate == 7;
	if(!graph && (!g.linear || g.lineRate != 6)) { snprintf(g_err, sizeof g_err, "rank kernels: 64 B linear or 128 B graph sides only"); return H2G_ERR_UNSUPPORTED; }
	HIPCHK(hipSetDevice(s->ix->device));
	const size_t nlanes = nchains / (size_t)chains_per_lane;
	const unsigned grid = (unsigned)((nlanes + (size_t)block - 1) / (size_t)block);
	if(repeats < 1) repeats = 1;
	const void* fn = nullptr;
#define H2G_CHAIN_CASE(CC) case CC: fn = graph ? (const void*)k_rank_chain<CC, true> : (const void*)k_rank_chain<CC, false>; break;
	switch(chains_per_lane) { H2G_CHAIN_CASE(1) H2G_CHAIN_CASE(2) H2G_CHAIN_CASE(4) H2G_CHAIN_CASE(8) }
#undef H2G_CHAIN_CASE
	if(lds_bytes > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) (void)hipGetLastError();
	unsigned long long* dsum = s->d_counters + 7;
	HIPCHK(hipEventRecord(s->ev[0], s->st));
	for(int r = 0; r < repeats; r++) {
		HIPCHK(hipMemsetAsync(dsum, 0, 8, s->st));
		void* args[] = {(void*)&g, (void*)&seed, (void*)&steps, (void*)&nlanes, (void*)&dsum};
		HIPCHK(hipLaunchKernel(fn, dim3(grid), dim3((unsigned)block), args, (size_t)lds_bytes, s->st));
	}
	HIPCHK(hipEventRecord(s->ev[1], s->st));
	unsigned long long v = 0;
	HIPCHK(hipMemcpyAsync(&v, dsum, 8, hipMemcpyDeviceToHost, s->st));
	HIPCHK(sync_all(s));
	HIPCHK(hipGetLastError());
	float t = 0;
	HIPCHK(hipEventElapsedTime(&t, s->ev[0], s->ev[1]));
	if(kernel_ms) *kernel_ms = t / repeats;
	if(checksum) *checksum = v;
	return H2G_OK;
}

// The same for whole graph LF steps on a REAL graph index (glf1_top_fused: the step of a coordinate walk — rank, rank_M, select_F: two or three dependent
// 128 B lines): C walks per lane, either one after the other (staged = 0: what a lane of the pass does today, C times) or stage by stage over all C
// (h2g_graph_staged.h: every walk's next line is requested before any of them is consumed).  A walk that runs into a '$' row starts over somewhere else.
template <int C>
__global__ __launch_bounds__(512) void k_glf_chain(DGfm g, uint64_t seed, int steps, size_t nlanes, int staged, unsigned long long* sum)
{
	extern __shared__ uint32_t chain_pad[];
	const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	unsigned long long acc = 0;
	if(t < nlanes) {
		uint32_t row[C];
#pragma unroll
		for(int k = 0; k < C; k++) { uint64_t h = splitmix64(seed + t * C + k); row[k] = (uint32_t)(h % g.gbwtLen); while(is_zoff(g, row[k])) { h = splitmix64(h); row[k] = (uint32_t)(h % g.gbwtLen); } }
		for(int i = 0; i < steps; i++) {
			uint32_t top[C], node[C];
			if(staged) {
				GlfStage st[C];
#pragma unroll
				for(int k = 0; k < C; k++) glf_stage_a(g, row[k], st[k]);
#pragma unroll
				for(int k = 0; k < C; k++) glf_stage_b(g, st[k]);
#pragma unroll
				for(int k = 0; k < C; k++) glf_stage_c(g, st[k]);
#pragma unroll
				for(int k = 0; k < C; k++) glf_stage_d(g, st[k], &top[k], &node[k]);
			} else {
#pragma unroll
				for(int k = 0; k < C; k++) glf1_top_fused(g, row[k], &top[k], &node[k]);
			}
#pragma unroll
			for(int k = 0; k < C; k++) {
				acc += (unsigned long long)top[k] + node[k];
				uint32_t nr = top[k];
				if(nr >= g.gbwtLen || is_zoff(g, nr)) { uint64_t h = splitmix64(((uint64_t)top[k] << 32) | row[k]); nr = (uint32_t)(h % g.gbwtLen); while(is_zoff(g, nr)) { h = splitmix64(h); nr = (uint32_t)(h % g.gbwtLen); } }
				row[k] = nr;
			}
		}
	}
	if(threadIdx.x == 0 && steps < 0) chain_pad[0] = (uint32_t)acc;
	for(int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
	if((threadIdx.x & 63) == 0 && acc) atomicAdd(sum, acc);
}

extern "C" H2G_EXPORT h2g_status h2g_glf_chain_bench(h2g_stream* s, size_t nchains, int chains_per_lane, int steps, int block, int lds_bytes, uint64_t seed,
                                                     int staged, int repeats, float* kernel_ms, uint64_t* checksum)
{
	if(!s || nchains == 0 || steps < 1 || block < 64 || block > 512 || (block & 63) || lds_bytes < 0 || lds_bytes > 160 * 1024) return H2G_ERR_ARG;
	if(chains_per_lane != 1 && chains_per_lane != 2 && chains_per_lane != 4) return H2G_ERR_ARG;
	if(nchains % (size_t)chains_per_lane) return H2G_ERR_ARG;
	if(s->ix->synthetic || s->ix->dg.linear || s->ix->dg.lineRate != 7) { snprintf(g_err, sizeof g_err, "h2g_glf_chain_bench: needs a real graph (GFM, 128 B side) index"); return H2G_ERR_ARG; }
	const DGfm& g = s->ix->dg;
	HIPCHK(hipSetDevice(s->ix->device));
	const size_t nlanes = nchains / (size_t)chains_per_lane;
	const unsigned grid = (unsigned)((nlanes + (size_t)block - 1) / (size_t)block);
	if(repeats < 1) repeats = 1;
	const void* fn = chains_per_lane == 1 ? (const void*)k_glf_chain<1> : chains_per_lane == 2 ? (const void*)k_glf_chain<2> : (const void*)k_glf_chain<4>;
	if(lds_bytes > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) (void)hipGetLastError();
	unsigned long long* dsum = s->d_counters + 7;
	HIPCHK(hipEventRecord(s->ev[0], s->st));
	for(int r = 0; r < repeats; r++) {
		HIPCHK(hipMemsetAsync(dsum, 0, 8, s->st));
		void* args[] = {(void*)&g, (void*)&seed, (void*)&steps, (void*)&nlanes, (void*)&staged, (void*)&dsum};
		HIPCHK(hipLaunchKernel(fn, dim3(grid), dim3((unsigned)block), args, (size_t)lds_bytes, s->st));
	}
	HIPCHK(hipEventRecord(s->ev[1], s->st));
	unsigned long long v = 0;
	HIPCHK(hipMemcpyAsync(&v, dsum, 8, hipMemcpyDeviceToHost, s->st));
	HIPCHK(sync_all(s));
	HIPCHK(hipGetLastError());
	float t = 0;
	HIPCHK(hipEventElapsedTime(&t, s->ev[0], s->ev[1]));
	if(kernel_ms) *kernel_ms = t / repeats;
	if(checksum) *checksum = v;
	return H2G_OK;
}

// ------------------------------------------------------------------------------------------ primitive kernels
__global__ __launch_bounds__(256) void k_fm_search(DGfm g, DReads rd, const h2g_fm_query* q, size_t n, uint32_t khits,
                                                   h2g_fm_hit* out)
{
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
		h2g_fm_query qq = q[i];
		SeqView sv = seq_view(rd, qq.read, qq.fw != 0);
		partial_search_item(g, sv, qq.offset, qq.pseudogeneStop != 0, qq.anchorStop != 0, khits, &out[i]);
	}
}

__global__ __launch_bounds__(256) void k_sa_resolve(DGfm g, const h2g_sa_query* q, size_t n, uint32_t cap, h2g_coord* coords,
                                                    h2g_sa_result* res)
{
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
		h2g_sa_query qq = q[i];
		genome_coords_item(g, qq.top, qq.bot, qq.maxelt, qq.len, qq.rejectStraddle != 0, coords + i * cap, cap, &res[i]);
	}
}

__global__ __launch_bounds__(256) void k_extend(DRef ref, DReads rd, DScoring sc, h2g_ghit* hits, const h2g_ext_args* args,
                                                size_t n, h2g_ext_result* res)
{
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
		h2g_ghit* h = &hits[i];
		SeqView sv = seq_view(rd, h->read, h->fw != 0);
		uint32_t le = 0, re = 0;
		bool ext = extend_item(ref, sc, sv, h, args[i].mm, args[i].max_leftext, args[i].max_rightext, &le, &re);
		res[i].extended = ext; res[i].leftext = le; res[i].rightext = re;
	}
}

// graph index: GenomeHit::extend through the ALT database (known SNPs / insertions / deletions)
__global__ __launch_bounds__(256) void k_extend_alts(DRef ref, DAlts alts, DReads rd, DScoring sc, h2g_ghit* hits, const h2g_ext_args* args,
                                                     size_t n, h2g_ext_result* res, AwaWS* scratch)
{
	const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	AwaWS* W = scratch + tid;
	for(size_t i = tid; i < n; i += stride) {
		h2g_ghit* h = &hits[i];
		SeqView sv = seq_view(rd, h->read, h->fw != 0);
		uint32_t le = 0, re = 0;
		bool ext = extend_item_alts(ref, alts, sc, sv, h, args[i].mm, args[i].max_leftext, args[i].max_rightext, &le, &re, W);
		res[i].extended = ext; res[i].leftext = le; res[i].rightext = re;
	}
}

// GenomeHit::combineWith (hi_aligner.h:1420-2025) per lane: a[i] <- a[i] + b[i] (hit_combine of h2g_align.h, the function OP_COMBINE / FOP_COMBINE run inside go())
__global__ __launch_bounds__(256) void k_combine(DRef ref, DAlts alts, DReads rd, AlnParams P, h2g_ghit* a, const h2g_ghit* b, const int64_t* minsc, size_t n, uint32_t* ok, int64_t* scratch)
{
	const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	int64_t* const t1 = scratch + tid * (size_t)(2 * H2G_COMBINE_MAXLEN);
	for(size_t i = tid; i < n; i += stride) {
		const SeqView sv = seq_view(rd, a[i].read, a[i].fw != 0);
		ok[i] = hit_combine(ref, P.sc, sv, &a[i], &b[i], minsc[i], P.minIntronLen, P.no_spliced != 0, ScVec{t1, 1}, ScVec{t1 + H2G_COMBINE_MAXLEN, 1}, &alts) ? 1u : 0u;
	}
}

__global__ __launch_bounds__(256) void k_adjust_alt(DGfm g, DRef ref, DAlts alts, DReads rd, const h2g_adjust_query* q, size_t n, uint32_t cap,
                                                    h2g_ghit* hits, uint32_t* nhits, AwaWS* scratch)
{
	const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	AwaWS* W = scratch + tid;
	for(size_t i = tid; i < n; i += stride) {
		SeqView sv = seq_view(rd, q[i].read, q[i].fw != 0);
		uint32_t nh = 0, ovf = 0;
		adjust_with_alt(g, ref, alts, sv, q[i].rdoff, q[i].len, q[i].tidx, q[i].toff, q[i].joinedOff, hits + i * cap, &nh, cap, W, &ovf);
		nhits[i] = ovf ? H2G_MAX : nh;
		for(uint32_t k = 0; k < nh; k++) hits[i * cap + k].read = q[i].read;
	}
}

__global__ __launch_bounds__(256) void k_graph_lf(DGfm g, const h2g_glf_query* q, size_t n, uint32_t k, h2g_glf_result* res, h2g_iedges* ie)
{
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
		GRange r;
		IEdges e;
		e.n = 0;
		bool ok;
		if(q[i].single) ok = map_glf1(g, q[i].top, q[i].c, &r);
		else ok = map_glf(g, q[i].top, q[i].bot, q[i].c, k, &r, &e);
		h2g_glf_result o;
		o.ok = ok; o.top = r.top; o.bot = r.bot; o.node_top = r.node_top; o.node_bot = r.node_bot;
		res[i] = o;
		if(ie) {
			ie[i].n = e.n;
			for(uint32_t j = 0; j < e.n && j < H2G_IEDGE_CAP; j++) { ie[i].e[j][0] = e.e[j][0]; ie[i].e[j][1] = e.e[j][1]; }
		}
	}
}

__global__ __launch_bounds__(256) void k_sa_resolve_graph(DGfm g, const h2g_gsa_query* q, const h2g_iedges* ie, size_t n, uint32_t cap,
                                                          h2g_coord* coords, h2g_sa_result* res, GwCtx* scratch)
{
	const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	GwCtx* x = scratch + tid;
	for(size_t i = tid; i < n; i += stride)
		genome_coords_graph_item(g, x, q[i].top, q[i].bot, q[i].node_top, q[i].node_bot, ie ? &ie[i] : nullptr, q[i].maxelt, q[i].len,
		                         q[i].rejectStraddle != 0, coords + i * cap, cap, &res[i]);
}

__global__ __launch_bounds__(256) void k_fm_search_graph(DGfm g, DReads rd, const h2g_fm_query* q, size_t n, uint32_t khits,
                                                         uint32_t kseeds, h2g_fm_hit* out, h2g_iedges* ie)
{
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
		SeqView sv = seq_view(rd, q[i].read, q[i].fw != 0);
		partial_search_graph_item(g, sv, q[i].offset, q[i].pseudogeneStop != 0, q[i].anchorStop != 0, khits, kseeds, &out[i],
		                          ie ? &ie[i] : nullptr);
	}
}

// ------------------------------------------------------------------------------------------ Smith-Waterman
static_assert(sizeof(h2g_sw_result) == sizeof(SwOut), "h2g_sw_result must mirror SwOut");
// Two kernels.  k_sw_fill: one wavefront per problem, register-resident systolic fill (sw_fill_wave), H/E/F streamed to an
// HBM workspace in anti-diagonal-major order (64 contiguous cells per step and matrix; 8-bit cells, 16-bit for a problem with minsc < -254).  k_sw_backtrace: one LANE per
// problem walks the reference's sequential gather/backtrace over that workspace, so 64 backtraces share a wavefront
// instead of 63 lanes idling behind one.
struct SwWs { uint8_t* base; size_t stride, mat_bytes, rf_bytes; };   // per problem: H | E | F (mat_bytes each) | rf (rf_bytes) | mask matrix of the rare second walk (SwMaskTab::direct)
__global__ __launch_bounds__(64) void k_sw_fill(DRef ref, DReads rd, SwParams P, const h2g_sw_query* q, size_t n, SwWs ws)
{
	__shared__ uint8_t s_rf[H2G_SW_MAX_COLS + 8];
	const uint32_t lane = threadIdx.x;
	for(size_t p = blockIdx.x; p < n; p += gridDim.x) {
		const h2g_sw_query qq = q[p];
		SeqView sv = seq_view(rd, qq.read, qq.fw != 0);
		const uint32_t nrow = sv.len;
		const SwRect rect = sw_frame(qq.refoff, nrow, ref.refLens[qq.tidx]);
		const uint32_t ncol = (uint32_t)(rect.refr - rect.refl + 1);
		uint8_t* w = ws.base + p * ws.stride;
		SwMats m;
		m.nrow = nrow; m.ncol = ncol; m.nd = nrow + ncol - 1; m.layout = 1; m.wide = sw_wide_for(qq.minsc);
		m.H = w; m.E = w + ws.mat_bytes; m.F = w + 2 * ws.mat_bytes; m.rf = s_rf;
		{   // reference window: BitPairReference::getStretch semantics (N / outside the sequence = 4)
			RefCursor rc;
			rc.init(&ref, qq.tidx);
			uint8_t* grf = w + 3 * ws.mat_bytes;
			for(uint32_t j = lane; j < ncol; j += 64) { const uint8_t c = (uint8_t)rc.get(rect.refl + (int64_t)j); s_rf[j] = c; grf[j] = c; }
		}
		__syncthreads();
		sw_fill_wave(m, P, sv, lane);
		__syncthreads();
	}
}

__global__ __launch_bounds__(256) void k_sw_backtrace(DRef ref, DReads rd, SwParams P, const h2g_sw_query* q, size_t n, SwWs ws,
                                                      SwLaneState* states, h2g_sw_result* out)
{
	const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	SwLaneState* ls = states + tid;
	for(size_t p = tid; p < n; p += stride) {
		const h2g_sw_query qq = q[p];
		SeqView sv = seq_view(rd, qq.read, qq.fw != 0);
		const uint32_t nrow = sv.len;
		const SwRect rect = sw_frame(qq.refoff, nrow, ref.refLens[qq.tidx]);
		const uint32_t ncol = (uint32_t)(rect.refr - rect.refl + 1);
		uint8_t* w = ws.base + p * ws.stride;
		SwMats m;
		m.nrow = nrow; m.ncol = ncol; m.nd = nrow + ncol - 1; m.layout = 1; m.wide = sw_wide_for(qq.minsc);
		m.H = w; m.E = w + ws.mat_bytes; m.F = w + 2 * ws.mat_bytes; m.rf = w + 3 * ws.mat_bytes;
		uint32_t rnd = qq.rnd;
		const SwOut* o = sw_finish(m, P, sv, rect, qq.minsc, &rnd, ls, reinterpret_cast<uint16_t*>(w + 3 * ws.mat_bytes + ws.rf_bytes));
		SwOut* dst = reinterpret_cast<SwOut*>(&out[p]);
		dst->found_align = o->found_align; dst->found = o->found; dst->best = o->best; dst->score = o->score; dst->off = o->off;
		dst->nedits = o->nedits; dst->gaps = o->gaps; dst->overflow = o->overflow; dst->rnd = o->rnd; dst->refl = o->refl; dst->refr = o->refr;
		for(uint32_t e = 0; e < o->nedits; e++) dst->edits[e] = o->edits[e];
	}
}

static int need_reads(h2g_stream* s) {
	if(s->n_reads == 0) { snprintf(g_err, sizeof g_err, "no read batch set (h2g_set_reads)"); return H2G_ERR_ARG; }
	return H2G_OK;
}
static int need_graph(h2g_stream* s) {
	if(s->ix->synthetic) { snprintf(g_err, sizeof g_err, "synthetic index: rank only"); return H2G_ERR_ARG; }
	if(s->ix->dg.linear || s->ix->dg.lineRate != 7) { snprintf(g_err, sizeof g_err, "not a graph (GFM, 128 B side) index"); return H2G_ERR_ARG; }
	return H2G_OK;
}
static int need_linear(h2g_stream* s) {
	if(s->ix->synthetic) { snprintf(g_err, sizeof g_err, "synthetic index: rank only"); return H2G_ERR_ARG; }
	if(!s->ix->dg.linear || s->ix->dg.lineRate != 6) { snprintf(g_err, sizeof g_err, "graph (GFM) indexes: not built yet"); return H2G_ERR_UNSUPPORTED; }
	return H2G_OK;
}

extern "C" h2g_status h2g_graph_lf(h2g_stream* s, const h2g_glf_query* q, size_t n, uint32_t k, h2g_glf_result* res, h2g_iedges* iedges) {
	if(!s || !q || !res || n == 0) return H2G_ERR_ARG;
	int rc;
	if((rc = need_graph(s))) return rc;
	const uint32_t glen = s->ix->dg.gbwtLen;
	for(size_t i = 0; i < n; i++) {
		if(q[i].c > 3 || q[i].top >= glen) return H2G_ERR_ARG;
		if(!q[i].single && (q[i].bot <= q[i].top + 1 || q[i].bot > glen)) return H2G_ERR_ARG;   // bloc.valid(): bot - top > 1
	}
	HIPCHK(hipSetDevice(s->ix->device));
	void *dq, *dres, *die = nullptr;
	if((rc = tmp_buf(s, 0, n * sizeof *q, &dq)) || (rc = tmp_buf(s, 1, n * sizeof *res, &dres))) return rc;
	if(iedges && (rc = tmp_buf(s, 2, n * sizeof *iedges, &die))) return rc;
	HIPCHK(hipMemcpyAsync(dq, q, n * sizeof *q, hipMemcpyHostToDevice, s->st));
	hipLaunchKernelGGL(k_graph_lf, dim3(grid_for(n, 256)), dim3(256), 0, s->st, s->ix->dg, (const h2g_glf_query*)dq, n, k, (h2g_glf_result*)dres, (h2g_iedges*)die);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(res, dres, n * sizeof *res, hipMemcpyDeviceToHost, s->st));
	if(iedges) HIPCHK(hipMemcpyAsync(iedges, die, n * sizeof *iedges, hipMemcpyDeviceToHost, s->st));
	HIPCHK(sync_all(s));
	return H2G_OK;
}

extern "C" h2g_status h2g_adjust_with_alt(h2g_stream* s, const h2g_adjust_query* q, size_t n, uint32_t cap, h2g_ghit* hits, uint32_t* nhits) {
	if(!s || !q || !hits || !nhits || n == 0 || cap == 0) return H2G_ERR_ARG;
	int rc;
	if((rc = need_reads(s)) || (rc = need_graph(s))) return rc;
	for(size_t i = 0; i < n; i++) if(q[i].read >= s->n_reads || q[i].tidx >= s->ix->dr.nrefs) return H2G_ERR_ARG;
	HIPCHK(hipSetDevice(s->ix->device));
	const unsigned grid = grid_for(n, 256) > 128 ? 128 : grid_for(n, 256);
	void *dq, *dh, *dn, *dscr;
	if((rc = tmp_buf(s, 0, n * sizeof *q, &dq)) || (rc = tmp_buf(s, 1, n * cap * sizeof *hits, &dh)) || (rc = tmp_buf(s, 2, n * 4, &dn)) ||
	   (rc = tmp_buf(s, 3, (size_t)grid * 256 * sizeof(AwaWS), &dscr))) return rc;
	HIPCHK(hipMemcpyAsync(dq, q, n * sizeof *q, hipMemcpyHostToDevice, s->st));
	hipLaunchKernelGGL(k_adjust_alt, dim3(grid), dim3(256), 0, s->st, s->ix->dg, s->ix->dr, s->ix->dalts, dreads(s), (const h2g_adjust_query*)dq, n, cap,
	                   (h2g_ghit*)dh, (uint32_t*)dn, (AwaWS*)dscr);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(hits, dh, n * cap * sizeof *hits, hipMemcpyDeviceToHost, s->st));
	HIPCHK(hipMemcpyAsync(nhits, dn, n * 4, hipMemcpyDeviceToHost, s->st));
	HIPCHK(sync_all(s));
	return H2G_OK;
}

extern "C" h2g_status h2g_sa_resolve_graph(h2g_stream* s, const h2g_gsa_query* q, const h2g_iedges* iedges, size_t n, uint32_t cap,
                                           h2g_coord* coords, h2g_sa_result* res)
{
	if(!s || !q || !coords || !res || n == 0 || cap == 0) return H2G_ERR_ARG;
	int rc;
	if((rc = need_graph(s))) return rc;
	const uint32_t glen = s->ix->dg.gbwtLen;
	for(size_t i = 0; i < n; i++)
		if(q[i].top >= q[i].bot || q[i].bot > glen || q[i].node_top >= q[i].node_bot || q[i].node_bot - q[i].node_top > q[i].bot - q[i].top) return H2G_ERR_ARG;
	HIPCHK(hipSetDevice(s->ix->device));
	const unsigned grid = grid_for(n, 256) > 128 ? 128 : grid_for(n, 256);   // 32 k lanes x sizeof(GwCtx) of scratch
	void *dq, *dco, *dres, *die = nullptr, *dscr;
	if((rc = tmp_buf(s, 0, n * sizeof *q, &dq)) || (rc = tmp_buf(s, 1, n * cap * sizeof *coords, &dco)) ||
	   (rc = tmp_buf(s, 2, n * sizeof *res + (iedges ? n * sizeof *iedges : 0), &dres)) ||
	   (rc = tmp_buf(s, 3, (size_t)grid * 256 * sizeof(GwCtx), &dscr))) return rc;
	HIPCHK(hipMemcpyAsync(dq, q, n * sizeof *q, hipMemcpyHostToDevice, s->st));
	if(iedges) {
		die = (char*)dres + n * sizeof *res;
		HIPCHK(hipMemcpyAsync(die, iedges, n * sizeof *iedges, hipMemcpyHostToDevice, s->st));
	}
	hipLaunchKernelGGL(k_sa_resolve_graph, dim3(grid), dim3(256), 0, s->st, s->ix->dg, (const h2g_gsa_query*)dq, (const h2g_iedges*)die, n, cap,
	                   (h2g_coord*)dco, (h2g_sa_result*)dres, (GwCtx*)dscr);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(coords, dco, n * cap * sizeof *coords, hipMemcpyDeviceToHost, s->st));
	HIPCHK(hipMemcpyAsync(res, dres, n * sizeof *res, hipMemcpyDeviceToHost, s->st));
	HIPCHK(sync_all(s));
	return H2G_OK;
}

extern "C" h2g_status h2g_fm_search_graph(h2g_stream* s, const h2g_fm_query* q, size_t n, uint32_t khits, uint32_t kseeds,
                                          h2g_fm_hit* out, h2g_iedges* iedges)
{
	if(!s || !q || !out || n == 0) return H2G_ERR_ARG;
	int rc;
	if((rc = need_reads(s)) || (rc = need_graph(s))) return rc;
	for(size_t i = 0; i < n; i++) {
		if(q[i].read >= s->n_reads) return H2G_ERR_ARG;
		if(q[i].mode != H2G_FM_PARTIAL) { snprintf(g_err, sizeof g_err, "fm_search: only H2G_FM_PARTIAL built"); return H2G_ERR_UNSUPPORTED; }
	}
	HIPCHK(hipSetDevice(s->ix->device));
	void *dq, *dout, *die = nullptr;
	if((rc = tmp_buf(s, 0, n * sizeof *q, &dq)) || (rc = tmp_buf(s, 1, n * sizeof *out, &dout))) return rc;
	if(iedges && (rc = tmp_buf(s, 2, n * sizeof *iedges, &die))) return rc;
	HIPCHK(hipMemcpyAsync(dq, q, n * sizeof *q, hipMemcpyHostToDevice, s->st));
	hipLaunchKernelGGL(k_fm_search_graph, dim3(grid_for(n, 256)), dim3(256), 0, s->st, s->ix->dg, dreads(s), (const h2g_fm_query*)dq, n,
	                   khits, kseeds, (h2g_fm_hit*)dout, (h2g_iedges*)die);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(out, dout, n * sizeof *out, hipMemcpyDeviceToHost, s->st));
	if(iedges) HIPCHK(hipMemcpyAsync(iedges, die, n * sizeof *iedges, hipMemcpyDeviceToHost, s->st));
	HIPCHK(sync_all(s));
	return H2G_OK;
}

extern "C" h2g_status h2g_fm_search(h2g_stream* s, const h2g_fm_query* q, size_t n, uint32_t khits, h2g_fm_hit* out) {
	if(!s || !q || !out || n == 0) return H2G_ERR_ARG;
	int rc;
	if(s->ix && !s->ix->synthetic && !s->ix->dg.linear && s->ix->dg.lineRate == 7)
		return h2g_fm_search_graph(s, q, n, khits, khits * 2 > 5 ? khits * 2 : 5, out, nullptr);
	if((rc = need_reads(s)) || (rc = need_linear(s))) return rc;
	for(size_t i = 0; i < n; i++) {
		if(q[i].read >= s->n_reads) return H2G_ERR_ARG;
		if(q[i].mode != H2G_FM_PARTIAL) { snprintf(g_err, sizeof g_err, "fm_search: only H2G_FM_PARTIAL built"); return H2G_ERR_UNSUPPORTED; }
	}
	HIPCHK(hipSetDevice(s->ix->device));
	void *dq, *dout;
	if((rc = tmp_buf(s, 0, n * sizeof *q, &dq)) || (rc = tmp_buf(s, 1, n * sizeof *out, &dout))) return rc;
	HIPCHK(hipMemcpyAsync(dq, q, n * sizeof *q, hipMemcpyHostToDevice, s->st));
	hipLaunchKernelGGL(k_fm_search, dim3(grid_for(n, 256)), dim3(256), 0, s->st, s->ix->dg, dreads(s), (const h2g_fm_query*)dq, n, khits, (h2g_fm_hit*)dout);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(out, dout, n * sizeof *out, hipMemcpyDeviceToHost, s->st));
	HIPCHK(sync_all(s));
	return H2G_OK;
}

// ------------------------------------------------------------------------------------------ globalGFMSearch / localGFMSearch
// one query on whichever index this is (the item function of both kernels below; sbase/fbase != nullptr: the local index is in LDS)
template <bool GRAPH>
__device__ __forceinline__ void ext_search_item(const DGfm& g, const DLocalSet& ls, const DReads& rd, const h2g_ext_search_query& qq, uint32_t minK,
                                                uint32_t minK_local, uint32_t kseeds, const uint8_t* sbase, const uint16_t* fbase, h2g_ext_search_hit* o)
{
	const SeqView sv = seq_view(rd, qq.read, qq.fw != 0);
	uint32_t hitlen = 0, top = H2G_MAX, bot = H2G_MAX, nr[2] = {0, 0}, nelt = 0;
	bool us = qq.uniqueStop != 0;
	if(qq.lidx == H2G_MAX) {
		GIdx gx; gx.g = &g;
		if(!GRAPH) nelt = gfm_search(gx, sv, qq.rdoff, &hitlen, &top, &bot, &us, minK, H2G_MAX, kseeds, false, nr);
		else { GRange r; r.top = top; r.bot = bot; r.node_top = r.node_bot = 0; IEdges ie; nelt = gfm_search_graph(g, gx, sv, qq.rdoff, &hitlen, &r, &ie, &us, minK, H2G_MAX, kseeds, false, kseeds, nr); if(nelt > 0) { top = r.top; bot = r.bot; } }
	} else {
		LIdx lx; lx.ls = &ls; lx.d = &ls.desc[qq.lidx]; lx.sbase = sbase; lx.fbase = fbase;
		if(lx.d->len == 0) nelt = 0;
		else if(!GRAPH) nelt = gfm_search(lx, sv, qq.rdoff, &hitlen, &top, &bot, &us, minK_local, qq.maxHitLen, kseeds, true, nr);
		else if(local_is_linear(*lx.d)) {   // a local index without a variant is linear inside a graph index (128 B sides): h2g_align.h LIdxW
			LIdxW lw; lw.ls = &ls; lw.d = lx.d;
			nelt = gfm_search(lw, sv, qq.rdoff, &hitlen, &top, &bot, &us, minK_local, qq.maxHitLen, kseeds, true, nr);
		}
		else { const LGfm x = lgfm_of(ls, *lx.d); GRange r; r.top = top; r.bot = bot; r.node_top = r.node_bot = 0; IEdges ie;
		       nelt = gfm_search_graph(x, lx, sv, qq.rdoff, &hitlen, &r, &ie, &us, minK_local, qq.maxHitLen, kseeds, true, kseeds, nr); top = r.top; bot = r.bot; }
	}
	o->nelt = nelt; o->hitlen = hitlen; o->top = top; o->bot = bot; o->uniqueStop = us ? 1u : 0u; o->nrank = nr[0]; o->nside = nr[1]; o->staged = sbase ? 1u : 0u;
}

// lane per query, sides from HBM: global queries, local queries of thin buckets, graph-index locals
template <bool GRAPH>
__global__ __launch_bounds__(256) void k_ext_search_hbm(DGfm g, DLocalSet ls, DReads rd, const h2g_ext_search_query* q, const uint32_t* order, size_t n,
                                                        uint32_t minK, uint32_t minK_local, uint32_t kseeds, h2g_ext_search_hit* out)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
		const uint32_t k = order[i];
		ext_search_item<GRAPH>(g, ls, rd, q[k], minK, minK_local, kseeds, nullptr, nullptr, &out[k]);
	}
}

// one workgroup per bucket of same-local-index queries: the index's sides and ftab are copied into LDS once (coalesced 16 B
// loads), then every Occ-rank of the bucket's queries is an LDS read.  Linear local indexes (64 B sides of 16-bit words).
__global__ __launch_bounds__(256) void k_ext_search_lds(DGfm g, DLocalSet ls, DReads rd, const h2g_ext_search_query* q, const uint32_t* order,
                                                        const uint2* buckets /* {first, count} into order */, uint32_t nbuckets, uint32_t minK, uint32_t minK_local,
                                                        uint32_t kseeds, h2g_ext_search_hit* out)
{
	extern __shared__ uint4 s_stage[];
	for(uint32_t b = blockIdx.x; b < nbuckets; b += gridDim.x) {
		const uint2 bk = buckets[b];
		const uint32_t lidx = q[order[bk.x]].lidx;
		const DLocalDesc& d = ls.desc[lidx];
		const uint32_t nside16 = (d.sides_bytes + 15) / 16, nftab = (1u << (2 * ls.ftabChars)) + 1;
		const uint4* src = reinterpret_cast<const uint4*>(ls.sides + d.sides_off);
		for(uint32_t i = threadIdx.x; i < nside16; i += blockDim.x) s_stage[i] = src[i];
		uint16_t* s_ftab = reinterpret_cast<uint16_t*>(s_stage + nside16);
		for(uint32_t i = threadIdx.x; i < nftab; i += blockDim.x) s_ftab[i] = ls.words[d.ftab_off + i];
		__syncthreads();
		for(uint32_t i = threadIdx.x; i < bk.y; i += blockDim.x) {
			const uint32_t k = order[bk.x + i];
			ext_search_item<false>(g, ls, rd, q[k], minK, minK_local, kseeds, reinterpret_cast<const uint8_t*>(s_stage), s_ftab, &out[k]);
		}
		__syncthreads();
	}
}

extern "C" uint32_t h2g_local_index_of(const h2g_index* ix, uint32_t tidx, uint32_t toff) {
	if(!ix || !ix->has_local || tidx >= ix->host.local_first.size() - 1) return H2G_MAX;
	const uint32_t a = ix->host.local_first[tidx], b = ix->host.local_first[tidx + 1], k = toff / H2G_LOCAL_INTERVAL;
	return a + k >= b ? H2G_MAX : a + k;
}

extern "C" h2g_status h2g_ext_search(h2g_stream* s, const h2g_ext_search_query* q, size_t n, uint32_t stage_min, h2g_ext_search_hit* out,
                                     h2g_ext_search_stats* stats)
{
	if(!s || !q || !out || n == 0 || n > 0xffffffffull) return H2G_ERR_ARG;
	int rc;
	if((rc = need_reads(s))) return rc;
	if(s->ix->synthetic) { snprintf(g_err, sizeof g_err, "synthetic index: rank only"); return H2G_ERR_ARG; }
	const bool graph = !s->ix->dg.linear;
	const uint32_t nlocal = s->ix->dls.n;
	for(size_t i = 0; i < n; i++) {
		if(q[i].read >= s->n_reads) return H2G_ERR_ARG;
		if(q[i].lidx != H2G_MAX && (!s->ix->has_local || q[i].lidx >= nlocal)) return H2G_ERR_ARG;
	}
	HIPCHK(hipSetDevice(s->ix->device));
	// bucket the local queries by index (host: the queries come from the host anyway); thin buckets and global queries go to the HBM kernel
	std::vector<uint32_t> ord(n);
	for(size_t i = 0; i < n; i++) ord[i] = (uint32_t)i;
	std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return q[a].lidx < q[b].lidx; });
	std::vector<uint32_t> o_lds, o_hbm;
	std::vector<uint2> buckets;
	h2g_ext_search_stats st;
	memset(&st, 0, sizeof st);
	uint32_t max_stage = 0;
	for(size_t i = 0; i < n;) {
		size_t j = i;
		while(j < n && q[ord[j]].lidx == q[ord[i]].lidx) j++;
		const uint32_t lidx = q[ord[i]].lidx;
		const bool local = lidx != H2G_MAX;
		if(local) { st.n_local += j - i; st.n_buckets++; }
		const DLocalDesc* d = local ? &s->ix->h_ldesc[lidx] : nullptr;
		if(local && !graph && stage_min > 0 && j - i >= stage_min && d->len > 0) {
			// a fat bucket is cut into workgroup-sized pieces (each stages its own copy: 25-40 KB against >= 1 MB of side reads saved)
			const uint32_t bytes = (d->sides_bytes + 15) / 16 * 16 + ((1u << (2 * s->ix->dls.ftabChars)) + 1) * 2 + 16;
			for(size_t c = i; c < j; c += 1024) {
				const size_t ce = c + 1024 < j ? c + 1024 : j;
				buckets.push_back(make_uint2((uint32_t)o_lds.size(), (uint32_t)(ce - c)));
				for(size_t k = c; k < ce; k++) o_lds.push_back(ord[k]);
				st.n_buckets_staged++;
				st.lds_bytes_staged += bytes;
			}
			st.n_staged += j - i;
			if(bytes > max_stage) max_stage = bytes;
		} else for(size_t k = i; k < j; k++) o_hbm.push_back(ord[k]);
		i = j;
	}
	void *dq, *dout, *dord, *dbk;
	if((rc = tmp_buf(s, 0, n * sizeof *q, &dq)) || (rc = tmp_buf(s, 1, n * sizeof *out, &dout)) || (rc = tmp_buf(s, 2, n * 4 + 64, &dord)) ||
	   (rc = tmp_buf(s, 3, buckets.size() * sizeof(uint2) + 64, &dbk))) return rc;
	HIPCHK(hipMemcpyAsync(dq, q, n * sizeof *q, hipMemcpyHostToDevice, s->st));
	uint32_t* d_olds = (uint32_t*)dord;
	uint32_t* d_ohbm = d_olds + o_lds.size();
	if(!o_lds.empty()) HIPCHK(hipMemcpyAsync(d_olds, o_lds.data(), o_lds.size() * 4, hipMemcpyHostToDevice, s->st));
	if(!o_hbm.empty()) HIPCHK(hipMemcpyAsync(d_ohbm, o_hbm.data(), o_hbm.size() * 4, hipMemcpyHostToDevice, s->st));
	if(!buckets.empty()) HIPCHK(hipMemcpyAsync(dbk, buckets.data(), buckets.size() * sizeof(uint2), hipMemcpyHostToDevice, s->st));
	h2g_align_params ap;
	align_params_defaults(&ap, !graph);
	const AlnParams P = aln_params_from(ap, true, !graph);
	const uint32_t minK = s->ix->dg.minK, minKl = P.minK_local, kseeds = P.kseeds;
	HIPCHK(hipEventRecord(s->ev[0], s->st));
	if(!buckets.empty()) {
		const unsigned grid = (unsigned)(buckets.size() < 256 * 8 ? buckets.size() : 256 * 8);
		hipLaunchKernelGGL(k_ext_search_lds, dim3(grid), dim3(256), max_stage, s->st, s->ix->dg, s->ix->dls, dreads(s), (const h2g_ext_search_query*)dq, d_olds,
		                   (const uint2*)dbk, (uint32_t)buckets.size(), minK, minKl, kseeds, (h2g_ext_search_hit*)dout);
	}
	HIPCHK(hipEventRecord(s->ev[1], s->st));
	if(!o_hbm.empty()) {
		if(graph) hipLaunchKernelGGL((k_ext_search_hbm<true>), dim3(grid_for(o_hbm.size(), 256)), dim3(256), 0, s->st, s->ix->dg, s->ix->dls, dreads(s),
		                             (const h2g_ext_search_query*)dq, d_ohbm, o_hbm.size(), minK, minKl, kseeds, (h2g_ext_search_hit*)dout);
		else hipLaunchKernelGGL((k_ext_search_hbm<false>), dim3(grid_for(o_hbm.size(), 256)), dim3(256), 0, s->st, s->ix->dg, s->ix->dls, dreads(s),
		                        (const h2g_ext_search_query*)dq, d_ohbm, o_hbm.size(), minK, minKl, kseeds, (h2g_ext_search_hit*)dout);
	}
	HIPCHK(hipEventRecord(s->ev[9], s->st));
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(out, dout, n * sizeof *out, hipMemcpyDeviceToHost, s->st));
	HIPCHK(sync_all(s));
	float t = 0;
	if(hipEventElapsedTime(&t, s->ev[0], s->ev[1]) == hipSuccess) st.ms_staged = t;
	if(hipEventElapsedTime(&t, s->ev[1], s->ev[9]) == hipSuccess) st.ms_hbm = t;
	if(stats) *stats = st;
	return H2G_OK;
}

extern "C" h2g_status h2g_sw_align(h2g_stream* s, const h2g_sw_query* q, size_t n, h2g_sw_result* out, int repeats, float* kernel_ms) {
	if(!s || !q || !out || n == 0) return H2G_ERR_ARG;
	int rc;
	if((rc = need_reads(s))) return rc;
	if(s->ix->synthetic) { snprintf(g_err, sizeof g_err, "synthetic index: rank only"); return H2G_ERR_ARG; }
	for(size_t i = 0; i < n; i++) if(q[i].read >= s->n_reads || q[i].tidx >= s->ix->dr.nrefs) return H2G_ERR_ARG;
	const uint32_t maxlen = s->max_read_len;   // longest read of the resident batch: sizes the per-problem workspace
	if(maxlen == 0 || maxlen > H2G_SW_MAX_ROWS) { snprintf(g_err, sizeof g_err, "h2g_sw_align: read length %u outside 1..%d", maxlen, H2G_SW_MAX_ROWS); return H2G_ERR_ARG; }
	HIPCHK(hipSetDevice(s->ix->device));
	const size_t ncolmax = maxlen + 4 * H2G_SW_MAXGAP, ndmax = maxlen + ncolmax - 1;
	SwWs ws;
	bool any_wide = false;                                        // 16-bit cells for the problems with minsc < -254 (aligner_sw.cpp:496)
	for(size_t i = 0; i < n && !any_wide; i++) any_wide = sw_wide_for(q[i].minsc);
	ws.mat_bytes = ((size_t)((maxlen + 63) / 64) * ndmax * 64) << (any_wide ? 1 : 0);
	ws.rf_bytes = (ncolmax + 255) & ~(size_t)255;
	ws.stride = 3 * ws.mat_bytes + ws.rf_bytes + (((size_t)maxlen * ncolmax * 2 + 255) & ~(size_t)255);
	// the batch is sized by BYTES, not by a problem count: a problem's stride runs from ~135 KB (101 rows, 8-bit cells: three H/E/F matrices + the
	// second walk's mask matrix) to ~1 MB (256 rows, 16-bit cells), and the workspace sits next to a human-size index
	const size_t ws_budget = (size_t)6 << 30;
	size_t batch = ws_budget / ws.stride;
	if(batch > 32768) batch = 32768;
	if(batch < 256) batch = 256;
	if(batch > n) batch = n;
	const size_t bt_threads = ((batch + 255) / 256) * 256;
	if(s->sw_ws_bytes < batch * ws.stride) {
		(void)hipFree(s->d_sw_ws); s->d_sw_ws = nullptr; s->sw_ws_bytes = 0;
		HIPCHK(hipMalloc((void**)&s->d_sw_ws, batch * ws.stride));
		s->sw_ws_bytes = batch * ws.stride;
	}
	if(s->sw_states < bt_threads) {
		(void)hipFree(s->d_sw_states); s->d_sw_states = nullptr; s->sw_states = 0;
		HIPCHK(hipMalloc((void**)&s->d_sw_states, bt_threads * sizeof(SwLaneState)));
		HIPCHK(hipMemset(s->d_sw_states, 0, bt_threads * sizeof(SwLaneState)));   // mask-table generations start at 0
		s->sw_states = bt_threads;
	}
	ws.base = s->d_sw_ws;
	void *dq, *dout;
	if((rc = tmp_buf(s, 0, n * sizeof *q, &dq)) || (rc = tmp_buf(s, 1, n * sizeof *out, &dout))) return rc;
	HIPCHK(hipMemcpyAsync(dq, q, n * sizeof *q, hipMemcpyHostToDevice, s->st));
	SwParams P;
	if(repeats < 1) repeats = 1;
	HIPCHK(hipEventRecord(s->ev[0], s->st));
	for(int r = 0; r < repeats; r++) {
		for(size_t first = 0; first < n; first += batch) {
			const size_t nb = n - first < batch ? n - first : batch;
			hipLaunchKernelGGL(k_sw_fill, dim3((unsigned)nb), dim3(64), 0, s->st, s->ix->dr, dreads(s), P, (const h2g_sw_query*)dq + first, nb, ws);
			hipLaunchKernelGGL(k_sw_backtrace, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s->st, s->ix->dr, dreads(s), P,
			                   (const h2g_sw_query*)dq + first, nb, ws, s->d_sw_states, (h2g_sw_result*)dout + first);
		}
	}
	HIPCHK(hipEventRecord(s->ev[1], s->st));
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(out, dout, n * sizeof *out, hipMemcpyDeviceToHost, s->st));
	HIPCHK(sync_all(s));
	float t = 0;
	HIPCHK(hipEventElapsedTime(&t, s->ev[0], s->ev[1]));
	if(kernel_ms) *kernel_ms = t / repeats;
	return H2G_OK;
}

extern "C" h2g_status h2g_sa_resolve(h2g_stream* s, const h2g_sa_query* q, size_t n, uint32_t cap, h2g_coord* coords,
                                     h2g_sa_result* res)
{
	if(!s || !q || !coords || !res || n == 0 || cap == 0) return H2G_ERR_ARG;
	int rc;
	if((rc = need_linear(s))) return rc;
	const uint32_t glen = s->ix->dg.gbwtLen;
	for(size_t i = 0; i < n; i++) if(q[i].top >= q[i].bot || q[i].bot > glen) return H2G_ERR_ARG;
	HIPCHK(hipSetDevice(s->ix->device));
	void *dq, *dco, *dres;
	if((rc = tmp_buf(s, 0, n * sizeof *q, &dq)) || (rc = tmp_buf(s, 1, n * cap * sizeof *coords, &dco)) ||
	   (rc = tmp_buf(s, 2, n * sizeof *res, &dres))) return rc;
	HIPCHK(hipMemcpyAsync(dq, q, n * sizeof *q, hipMemcpyHostToDevice, s->st));
	HIPCHK(hipMemsetAsync(dco, 0xff, n * cap * sizeof *coords, s->st));
	hipLaunchKernelGGL(k_sa_resolve, dim3(grid_for(n, 256)), dim3(256), 0, s->st, s->ix->dg, (const h2g_sa_query*)dq, n, cap, (h2g_coord*)dco, (h2g_sa_result*)dres);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(coords, dco, n * cap * sizeof *coords, hipMemcpyDeviceToHost, s->st));
	HIPCHK(hipMemcpyAsync(res, dres, n * sizeof *res, hipMemcpyDeviceToHost, s->st));
	HIPCHK(sync_all(s));
	return H2G_OK;
}

extern "C" h2g_status h2g_extend(h2g_stream* s, h2g_ghit* hits, const h2g_ext_args* args, size_t n, h2g_ext_result* res) {
	if(!s || !hits || !args || !res || n == 0) return H2G_ERR_ARG;
	int rc;
	if((rc = need_reads(s))) return rc;
	if(s->ix->synthetic) { snprintf(g_err, sizeof g_err, "synthetic index: rank only"); return H2G_ERR_ARG; }
	const bool graph = !s->ix->dg.linear;
	for(size_t i = 0; i < n; i++) {
		if(hits[i].read >= s->n_reads || hits[i].tidx >= s->ix->dr.nrefs || hits[i].nedits > H2G_MAX_EDITS) return H2G_ERR_ARG;
	}
	HIPCHK(hipSetDevice(s->ix->device));
	void *dh, *da, *dres;
	if((rc = tmp_buf(s, 0, n * sizeof *hits, &dh)) || (rc = tmp_buf(s, 1, n * sizeof *args, &da)) ||
	   (rc = tmp_buf(s, 2, n * sizeof *res, &dres))) return rc;
	HIPCHK(hipMemcpyAsync(dh, hits, n * sizeof *hits, hipMemcpyHostToDevice, s->st));
	HIPCHK(hipMemcpyAsync(da, args, n * sizeof *args, hipMemcpyHostToDevice, s->st));
	DScoring sc;
	if(graph) {
		const unsigned grid = grid_for(n, 256) > 256 ? 256 : grid_for(n, 256);
		void* dscr;
		if((rc = tmp_buf(s, 3, (size_t)grid * 256 * sizeof(AwaWS), &dscr))) return rc;
		hipLaunchKernelGGL(k_extend_alts, dim3(grid), dim3(256), 0, s->st, s->ix->dr, s->ix->dalts, dreads(s), sc, (h2g_ghit*)dh,
		                   (const h2g_ext_args*)da, n, (h2g_ext_result*)dres, (AwaWS*)dscr);
	} else
	hipLaunchKernelGGL(k_extend, dim3(grid_for(n, 256)), dim3(256), 0, s->st, s->ix->dr, dreads(s), sc, (h2g_ghit*)dh, (const h2g_ext_args*)da, n, (h2g_ext_result*)dres);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(hits, dh, n * sizeof *hits, hipMemcpyDeviceToHost, s->st));
	HIPCHK(hipMemcpyAsync(res, dres, n * sizeof *res, hipMemcpyDeviceToHost, s->st));
	HIPCHK(sync_all(s));
	return H2G_OK;
}

// ------------------------------------------------------------------------------------------ fused seed stage

// K1: every (read, strand) runs partialSearch from offset 0 (one lane per item)
__global__ __launch_bounds__(256) void k_seed_search(DGfm g, DReads rd, h2g_seed_params p, h2g_seed_result* out,
                                                     unsigned long long* counters)
{
	const size_t n = (size_t)rd.n * 2;
	size_t stride = (size_t)gridDim.x * blockDim.x;
	unsigned long long nrank = 0, nside = 0;
	for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
		SeqView sv = seq_view(rd, (uint32_t)(i >> 1), (i & 1) == 0);
		h2g_fm_hit h;
		partial_search_item(g, sv, 0, p.pseudogeneStop != 0, p.anchorStop != 0, p.khits, &h);
		out[i].hit = h;
		nrank += h.nrank; nside += h.nside;
	}
	wave_add(counters + 0, nrank);
	wave_add(counters + 1, nside);
}

// K2: coordinates + 0-mismatch extension for every partial hit
__global__ __launch_bounds__(256) void k_seed_resolve_extend(DGfm g, DRef ref, DReads rd, DScoring sc, h2g_seed_result* out,
                                                             unsigned long long* counters)
{
	const size_t n = (size_t)rd.n * 2;
	size_t stride = (size_t)gridDim.x * blockDim.x;
	unsigned long long nsteps = 0, next = 0;
	h2g_ghit scratch;
	for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
		SeqView sv = seq_view(rd, (uint32_t)(i >> 1), (i & 1) == 0);
		resolve_extend_item(g, ref, sc, sv, &out[i], &scratch);
		nsteps += out[i].nsteps; next += out[i].ncoords;
	}
	wave_add(counters + 2, nsteps);
	wave_add(counters + 3, next);
}

extern "C" void h2g_seed_params_init(h2g_seed_params* p, const h2g_index* ix, int no_spliced) {
	// nextBWT hi_aligner.h:4669-4670: pseudogeneStop = linearFM && !no_spliced_alignment; anchorStop = !repeat index
	p->pseudogeneStop = (ix && ix->dg.linear && !no_spliced) ? 1 : 0;
	p->anchorStop = 1;
	p->khits = (ix && !ix->dg.linear) ? 10 : 5;   // -k default hisat2.cpp:3903-3906
	p->search_variant = 0;
}

extern "C" h2g_status h2g_seed_extend_run(h2g_stream* s, const h2g_seed_params* p) {
	if(!s || !p) return H2G_ERR_ARG;
	int rc;
	if((rc = need_reads(s)) || (rc = need_linear(s))) return rc;
	HIPCHK(hipSetDevice(s->ix->device));
	const size_t n = s->n_reads * 2;
	HIPCHK(hipMemsetAsync(s->d_counters, 0, 4 * sizeof(unsigned long long), s->st));
	DScoring sc;
	HIPCHK(hipEventRecord(s->ev[2], s->st));
	hipLaunchKernelGGL(k_seed_search, dim3(grid_for(n, 256)), dim3(256), 0, s->st, s->ix->dg, dreads(s), *p, s->d_seed, s->d_counters);
	HIPCHK(hipEventRecord(s->ev[3], s->st));
	hipLaunchKernelGGL(k_seed_resolve_extend, dim3(grid_for(n, 256)), dim3(256), 0, s->st, s->ix->dg, s->ix->dr, dreads(s), sc, s->d_seed, s->d_counters);
	HIPCHK(hipEventRecord(s->ev[4], s->st));
	HIPCHK(hipGetLastError());
	s->ran_seed = true;
	return H2G_OK;
}

extern "C" h2g_status h2g_seed_extend_fetch(h2g_stream* s, h2g_seed_result* out, size_t first_read, size_t n_reads) {
	if(!s || !out || first_read + n_reads > s->n_reads) return H2G_ERR_ARG;
	HIPCHK(hipMemcpyAsync(out, s->d_seed + first_read * 2, n_reads * 2 * sizeof *out, hipMemcpyDeviceToHost, s->st));
	HIPCHK(sync_all(s));
	return H2G_OK;
}

// ------------------------------------------------------------------------------------------ go() for the batch
static_assert(sizeof(h2g_alnres) == sizeof(AlnRec), "h2g_alnres must mirror AlnRec");
static_assert(sizeof(h2g_read_result) == 40, "h2g_read_result layout");
static_assert(sizeof(h2g_pair_result) == sizeof(PairOut), "h2g_pair_result must mirror PairOut");
static_assert(H2G_PAIR_CAP == AL_MAX_PAIRS, "pair capacity");

extern "C" int h2g_device_count(void) { int n = 0; if(hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; } return n; }

extern "C" void h2g_align_params_init(h2g_align_params* p, const h2g_index* ix) { align_params_defaults(p, !ix || ix->dg.linear); }

// hisat2.cpp applies its presets AFTER every option was read, and the index type decides the default -k:
//   khits starts at 10 (:336); -k sets it and saw_k (:1316-1322); --sensitive: bowtie2_dp 0 -> 1, khits < 10 -> 10 (+ saw_k),
//   --score-min L,0,-0.5 (:1892-1901); --very-sensitive: bowtie2_dp 2, khits < 30 -> 30 (+ saw_k), L,0,-1 (:1902-1909);
//   without saw_k khits = 5 on a linear index, 10 on a graph (:3903-3906); --max-seeds 0 -> max(5, 2 khits) (:3174-3176).
// So `--sensitive` alone keeps -k 5 on a linear index, and a preset's --score-min wins over an explicit one.
extern "C" void h2g_align_params_presets(h2g_align_params* p, const h2g_index* ix, int saw_k, uint32_t k_arg, uint32_t max_seeds_arg,
                                         int sensitive, int very_sensitive)
{
	if(!p) return;
	uint32_t khits = saw_k ? k_arg : 10u;
	bool sawk = saw_k != 0;
	if(sensitive) {
		if(p->bowtie2_dp == 0) p->bowtie2_dp = 1;
		if(khits < 10) { khits = 10; sawk = true; }
		p->score_min_type = 2; p->score_min_const = 0.0; p->score_min_coeff = (double)(-0.5f);
	} else if(very_sensitive) {
		p->bowtie2_dp = 2;
		if(khits < 30) { khits = 30; sawk = true; }
		p->score_min_type = 2; p->score_min_const = 0.0; p->score_min_coeff = (double)(-1.0f);
	}
	if(!sawk) khits = (!ix || ix->dg.linear) ? 5u : 10u;
	p->khits = khits;
	p->kseeds = max_seeds_arg ? max_seeds_arg : (khits * 2 > 5 ? khits * 2 : 5);
}

extern "C" h2g_status h2g_set_read_names(h2g_stream* s, const char* bytes, const uint32_t* offs, size_t n) {
	if(s) HIPCHK(sync_all(s));        // (a machine pass of the previous batch may still read the buffers this call replaces)
	if(!s || !bytes || !offs || n != s->n_reads || n == 0) return H2G_ERR_ARG;
	HIPCHK(hipSetDevice(s->ix->device));
	size_t nb = offs[n];
	if(s->names_cap < nb || !s->d_name_offs) {
		(void)hipFree(s->d_names); (void)hipFree(s->d_name_offs);
		s->d_names = nullptr; s->d_name_offs = nullptr;
		HIPCHK(hipMalloc((void**)&s->d_names, nb + 64));
		HIPCHK(hipMalloc((void**)&s->d_name_offs, (s->max_reads + 1) * 4));
		s->names_cap = nb;
	}
	HIPCHK(hipMemcpyAsync(s->d_names, bytes, nb, hipMemcpyHostToDevice, s->st));
	HIPCHK(hipMemcpyAsync(s->d_name_offs, offs, (n + 1) * 4, hipMemcpyHostToDevice, s->st));
	HIPCHK(sync_all(s));
	s->has_names = true;
	return H2G_OK;
}

extern "C" h2g_status h2g_set_mates(h2g_stream* s, const uint8_t* codes2, const uint32_t* offs2, const char* quals2,
                                    const char* nb2, const uint32_t* noffs2, size_t n)
{
	if(s) HIPCHK(sync_all(s));
	if(!s || !codes2 || !offs2 || !nb2 || !noffs2 || n != s->n_reads || n == 0) return H2G_ERR_ARG;
	if(offs2[n] > s->max_bases) return H2G_ERR_ARG;
	for(size_t i = 0; i < n; i++) { const uint32_t l = offs2[i + 1] - offs2[i]; if(l > s->max_read_len) s->max_read_len = l; }
	HIPCHK(hipSetDevice(s->ix->device));
	if(!s->d_codes2) {
		HIPCHK(hipMalloc((void**)&s->d_codes2, s->max_bases + 64));
		HIPCHK(hipMalloc((void**)&s->d_quals2, s->max_bases + 64));
		HIPCHK(hipMalloc((void**)&s->d_offs2, (s->max_reads + 1) * 4));
		HIPCHK(hipMalloc((void**)&s->d_name_offs2, (s->max_reads + 1) * 4));
	}
	(void)hipFree(s->d_names2); s->d_names2 = nullptr;
	HIPCHK(hipMalloc((void**)&s->d_names2, noffs2[n] + 64));
	HIPCHK(hipMemcpyAsync(s->d_codes2, codes2, offs2[n], hipMemcpyHostToDevice, s->st));
	HIPCHK(hipMemcpyAsync(s->d_offs2, offs2, (n + 1) * 4, hipMemcpyHostToDevice, s->st));
	s->has_quals2 = quals2 != nullptr;
	if(quals2) HIPCHK(hipMemcpyAsync(s->d_quals2, quals2, offs2[n], hipMemcpyHostToDevice, s->st));
	HIPCHK(hipMemcpyAsync(s->d_names2, nb2, noffs2[n], hipMemcpyHostToDevice, s->st));
	HIPCHK(hipMemcpyAsync(s->d_name_offs2, noffs2, (n + 1) * 4, hipMemcpyHostToDevice, s->st));
	HIPCHK(sync_all(s));
	s->has_mates = true;
	return H2G_OK;
}

// go() on a graph index: the index must be a SNP graph (ALT database present)
// GenomeHit::combineWith as a primitive of its own: a[i] becomes the combination of a[i] (the left hit) and b[i] on the resident reads, ok[i] the
// function's return value.  Scoring and splice policy: `p` (nullptr: the defaults of the index).
extern "C" h2g_status h2g_combine_with(h2g_stream* s, const h2g_align_params* p, h2g_ghit* a, const h2g_ghit* b, const int64_t* minsc, size_t n, uint32_t* ok) {
	if(!s || !a || !b || !minsc || !ok || n == 0) return H2G_ERR_ARG;
	int rc;
	if((rc = need_reads(s))) return rc;
	if(s->ix->synthetic) { snprintf(g_err, sizeof g_err, "synthetic index: rank only"); return H2G_ERR_ARG; }
	const bool linear = s->ix->dg.linear != 0;
	for(size_t i = 0; i < n; i++) {
		if(a[i].read >= s->n_reads || b[i].read != a[i].read || a[i].tidx >= s->ix->dr.nrefs || b[i].tidx >= s->ix->dr.nrefs || a[i].nedits > H2G_MAX_EDITS || b[i].nedits > H2G_MAX_EDITS) return H2G_ERR_ARG;
	}
	if(s->max_read_len > H2G_COMBINE_MAXLEN) return H2G_ERR_ARG;      // (the joint's prefix / suffix score arrays)
	h2g_align_params hp;
	if(p) hp = *p; else align_params_defaults(&hp, linear);
	AlnParams P = aln_params_from(hp, hp.no_spliced_alignment != 0, linear);
	if(!hp.no_spliced_alignment) { P.sc.donor_sum = s->ix->d_spl[0]; P.sc.acc_sum1 = s->ix->d_spl[1]; P.sc.acc_sum2 = s->ix->d_spl[2]; }
	HIPCHK(hipSetDevice(s->ix->device));
	unsigned grid = grid_for(n, 256);
	if(grid > 64) grid = 64;
	void *da, *db, *dm, *dscr;
	const size_t ab = n * sizeof(h2g_ghit);
	if((rc = tmp_buf(s, 0, 2 * ab, &da)) || (rc = tmp_buf(s, 1, n * 8, &dm)) || (rc = tmp_buf(s, 2, n * 4, &db)) ||
	   (rc = tmp_buf(s, 3, (size_t)grid * 256 * 2 * H2G_COMBINE_MAXLEN * sizeof(int64_t), &dscr))) return rc;
	h2g_ghit* const d_a = (h2g_ghit*)da; h2g_ghit* const d_b = d_a + n;
	HIPCHK(hipMemcpyAsync(d_a, a, ab, hipMemcpyHostToDevice, s->st));
	HIPCHK(hipMemcpyAsync(d_b, b, ab, hipMemcpyHostToDevice, s->st));
	HIPCHK(hipMemcpyAsync(dm, minsc, n * 8, hipMemcpyHostToDevice, s->st));
	hipLaunchKernelGGL(k_combine, dim3(grid), dim3(256), 0, s->st, s->ix->dr, s->ix->dalts, dreads(s), P, d_a, (const h2g_ghit*)d_b, (const int64_t*)dm, n, (uint32_t*)db, (int64_t*)dscr);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(a, d_a, ab, hipMemcpyDeviceToHost, s->st));
	HIPCHK(hipMemcpyAsync(ok, db, n * 4, hipMemcpyDeviceToHost, s->st));
	HIPCHK(sync_all(s));
	return H2G_OK;
}

static int need_alignable(h2g_stream* s) {
	if(s->ix->synthetic) { snprintf(g_err, sizeof g_err, "synthetic index: rank only"); return H2G_ERR_ARG; }
	const DGfm& g = s->ix->dg;
	if(g.linear && g.lineRate == 6) return H2G_OK;
	if(!g.linear && g.lineRate == 7) return H2G_OK;
	snprintf(g_err, sizeof g_err, "unsupported side geometry (lineRate %u)", g.lineRate);
	return H2G_ERR_UNSUPPORTED;
}

// the go() units: [linear?][big?]
struct GoUnit {
	size_t (*ws_bytes)(); size_t (*gws_bytes)(); int (*waves)(); void (*caps)(uint32_t*); int (*launch)(const GoArgs*, unsigned, hipStream_t);
	size_t (*slot_off)(); size_t (*gsl_off)(); void (*geometry)(uint32_t*); size_t (*sw_bytes)(uint32_t, int);
};
static const GoUnit& go_unit(bool linear, bool big, bool spliced = false) {
	// spliced alignment: the units whose machine carries the splice-site database joins
#define H2G_UNIT_ROW(N) {h2g_go_ws_bytes_##N, h2g_go_gws_bytes_##N, h2g_go_waves_##N, h2g_go_caps_##N, h2g_go_launch_##N, h2g_go_slot_off_##N, h2g_go_gsl_off_##N, h2g_go_geometry_##N, h2g_go_sw_bytes_##N}
	static const GoUnit spl[2][2] = {{H2G_UNIT_ROW(graph_spl), H2G_UNIT_ROW(graph_spl_big)}, {H2G_UNIT_ROW(linear_spl), H2G_UNIT_ROW(linear_spl_big)}};
	if(spliced) return spl[linear ? 1 : 0][big ? 1 : 0];
	static const GoUnit u[2][2] = {
		{{h2g_go_ws_bytes_graph, h2g_go_gws_bytes_graph, h2g_go_waves_graph, h2g_go_caps_graph, h2g_go_launch_graph, h2g_go_slot_off_graph, h2g_go_gsl_off_graph, h2g_go_geometry_graph, h2g_go_sw_bytes_graph},
		 {h2g_go_ws_bytes_graph_big, h2g_go_gws_bytes_graph_big, h2g_go_waves_graph_big, h2g_go_caps_graph_big, h2g_go_launch_graph_big, h2g_go_slot_off_graph_big, h2g_go_gsl_off_graph_big, h2g_go_geometry_graph_big, h2g_go_sw_bytes_graph_big}},
		{{h2g_go_ws_bytes_linear, h2g_go_gws_bytes_linear, h2g_go_waves_linear, h2g_go_caps_linear, h2g_go_launch_linear, h2g_go_slot_off_linear, h2g_go_gsl_off_linear, h2g_go_geometry_linear, h2g_go_sw_bytes_linear},
		 {h2g_go_ws_bytes_linear_big, h2g_go_gws_bytes_linear_big, h2g_go_waves_linear_big, h2g_go_caps_linear_big, h2g_go_launch_linear_big, h2g_go_slot_off_linear_big, h2g_go_gsl_off_linear_big, h2g_go_geometry_linear_big, h2g_go_sw_bytes_linear_big}}};
	return u[linear ? 1 : 0][big ? 1 : 0];
}

// sizes the per-lane scratch of one pass and fills the pool fields of `a`
static int go_pool_for(h2g_stream* s, int which, const GoUnit& u, size_t slots, size_t lanes, uint32_t bowtie2_dp, GoArgs* a) {
	h2g_stream::GoPool& pl = s->pool[which];
	const size_t wsb = u.ws_bytes(), gwb = u.gws_bytes();
	if(pl.ws_bytes < slots * wsb) {
		(void)hipFree(pl.ws); pl.ws = nullptr; pl.ws_bytes = 0;
		HIPCHK(hipMalloc((void**)&pl.ws, slots * wsb));
		pl.ws_bytes = slots * wsb;
	}
	a->pool = pl.ws; a->ws_stride = wsb; a->slot_off = u.slot_off(); a->gsl_off = u.gsl_off();
	a->gws_base = nullptr; a->gws_stride = gwb;
	if(gwb) {
		if(pl.gws_bytes < lanes * gwb) {
			(void)hipFree(pl.gws); pl.gws = nullptr; pl.gws_bytes = 0;
			HIPCHK(hipMalloc((void**)&pl.gws, lanes * gwb));
			pl.gws_bytes = lanes * gwb;
		}
		a->gws_base = pl.gws;
	}
	if(pl.sc_lanes < lanes) {
		(void)hipFree(pl.sc); pl.sc = nullptr; pl.sc_lanes = 0;
		HIPCHK(hipMalloc((void**)&pl.sc, lanes * (size_t)(2 * H2G_COMBINE_MAXLEN * sizeof(int64_t))));
		pl.sc_lanes = lanes;
	}
	a->sc_base = pl.sc;
	a->sw_base = nullptr; a->sw_stride = 0;
	if(bowtie2_dp) {
		const uint32_t swlen = s->max_read_len > H2G_SW_MAX_ROWS ? (uint32_t)H2G_SW_MAX_ROWS : s->max_read_len;
		bool wide = false;                                        // some read length of the batch may put minsc below -254: 16-bit cells
		for(uint32_t l = 1; l <= swlen && !wide; l++) wide = sw_wide_for(min_score_for(a->P, l));
		const size_t stride = (u.sw_bytes(swlen, wide ? 1 : 0) + 255) & ~(size_t)255;   // the unit's own SwLaneState (the large-workspace units hold 192-edit records)
		if(pl.sw_stride < stride || pl.sw_lanes < lanes) {
			(void)hipFree(pl.sw); pl.sw = nullptr; pl.sw_stride = 0; pl.sw_lanes = 0;
			HIPCHK(hipMalloc((void**)&pl.sw, stride * lanes));
			HIPCHK(hipMemset(pl.sw, 0, stride * lanes));   // SwLaneState: mask-table generations start at 0
			pl.sw_stride = stride; pl.sw_lanes = lanes;
		}
		a->sw_base = pl.sw; a->sw_stride = pl.sw_stride;
	}
	return H2G_OK;
}

// (the reads whose main-pass workspace overflowed are listed by the pass itself — MachOut::defer_list — and their rows left alone until the second pass writes them)

// HI_Aligner::go for every read (pair) of the resident batch.  Two passes, both asynchronous on the stream: the main pass
// with the default workspace, then the reads it flagged (a list overflowed) once more through the large-workspace unit,
// whose results replace theirs.  Reads still flagged after that keep `overflow` set (n_overflow counts them).
static h2g_status go_run(h2g_stream* s, const h2g_align_params* p, bool paired) {
	if(!s || !p) return H2G_ERR_ARG;
	int rc;
	if((rc = need_reads(s)) || (rc = need_alignable(s))) return rc;
	if(!s->ix->has_local) { snprintf(g_err, sizeof g_err, "align: index loaded without local (.5/.6) indexes"); return H2G_ERR_ARG; }
	if(!s->has_names || (paired && !s->has_mates)) { snprintf(g_err, sizeof g_err, "align: read names (h2g_set_read_names)%s not set", paired ? " / mates (h2g_set_mates)" : ""); return H2G_ERR_ARG; }
	const bool linear = s->ix->dg.linear != 0;
	// --haplotype is compiled into the units that also carry the splice-site database (H2G_HAPLOTYPE, h2g_graph.h); they run either mode
	// ... and so are -I, --rf / --ff, --nofw / --norc (H2G_EXT_OPTS, h2g_align.h)
	const bool ext_opts = pe_flags_from(*p) != H2G_PE_DEFAULT || p->min_frag_len != 0;
	const bool spl = !p->no_spliced_alignment || (p->use_haplotype && !linear) || ext_opts;
	if(!p->no_spliced_alignment) {
		// spliced alignment: combineWith places introns (hi_aligner.h:1588-1739) and every read is independent when novel splice
		// sites are not shared (--no-temp-splicesite); the shared SpliceSiteDB of the default mode and graph indexes are not built
		// no_temp_splicesite == 0 (the reference's default) is the CALLER's wave protocol: batches of <= window reads, the junctions of
		// each batch's output merged into the database (h2g_index_set_splice_sites) before the next one; first_read_id carries the ids
		if(p->pen_canintronlen_type < 1 || p->pen_canintronlen_type > 4 || p->pen_noncanintronlen_type < 1 || p->pen_noncanintronlen_type > 4 ||
		   p->min_intronlen < 20 || p->max_intronlen < p->min_intronlen || p->pen_cansplice < 0 || p->pen_noncansplice < 0 ||
		   p->min_anchor_len < 1 || p->min_anchor_len_noncan < 1) {
			snprintf(g_err, sizeof g_err, "align: splice scoring outside its range (intron-length function type 1..4, 20 <= min_intronlen <= max_intronlen, penalties >= 0)");
			return H2G_ERR_ARG;
		}
		// a splice edit keeps its intron length in 20 bits (include/h2g.h): longer introns are refused by name, never truncated
		if(p->max_intronlen > H2G_SPL_MAXLEN) {
			snprintf(g_err, sizeof g_err, "align: --max-intronlen %u is beyond the %u bases a splice edit holds", p->max_intronlen, (unsigned)H2G_SPL_MAXLEN);
			return H2G_ERR_UNSUPPORTED;
		}
	}
	const uint32_t maxsz = p->khits > p->kseeds ? p->khits : p->kseeds;
	uint32_t caps[5], bcaps[5];
	go_unit(linear, false, spl).caps(caps); go_unit(linear, true, spl).caps(bcaps);
	if(p->khits == 0 || p->khits > H2G_SELECT_CAP || p->kseeds < p->khits || maxsz > bcaps[0]) {
		snprintf(g_err, sizeof g_err, "align: -k %u / --max-seeds %u outside the built range (-k 1..%u, --max-seeds <= %u)", p->khits, p->kseeds, (unsigned)H2G_SELECT_CAP, bcaps[0]);
		return H2G_ERR_ARG;
	}
	if(p->bowtie2_dp > 2 || p->pe_orientation > 2 || p->min_frag_len >= (1u << 24)) return H2G_ERR_ARG;
	if(p->max_alts_tried && p->max_alts_tried < 8) { snprintf(g_err, sizeof g_err, "align: --max-altstried arg must be at least 8"); return H2G_ERR_ARG; }
	if(p->bowtie2_dp) {
		if(s->max_read_len == 0) return H2G_ERR_ARG;
		// a read longer than H2G_SW_MAX_ROWS is flagged by the kernel (overflow bit 256) instead of run through the DP
	}
	HIPCHK(hipSetDevice(s->ix->device));
	// Runs queued back to back share the result arrays (rows per read, record stride): a machine pass still in flight may only meet a
	// run over the same reads with the same options.  Anything else waits for the machine streams first.
	if(s->st2_busy && (s->last_paired != (paired ? 1 : 0) || memcmp(&s->last_p, p, sizeof *p) != 0)) {
		for(int k_ = 0; k_ < H2G_MSTREAMS_MAX; k_++) HIPCHK(hipStreamSynchronize(s->mst[k_]));
		s->st2_busy = false;
	}
	s->last_p = *p; s->last_paired = paired ? 1 : 0;
	const bool big_main = maxsz > caps[0];
	const GoUnit& U = go_unit(linear, big_main, spl);
	// (graph indexes: h2g_k_go_fast_graph.hip, the same pass over the graph form of the compact state; --haplotype and the pair-policy options are `spl`)
	const bool fast = s->tune.fast && !spl && !big_main && p->no_spliced_alignment && !p->secondary && !p->bowtie2_dp;
	// geometry of the unit: workgroups of geo[0] threads own geo[1] reads in flight; resident workgroups per CU = what the
	// unit's waves per SIMD and the LDS (rings + one packed-read region per mate) allow
	uint32_t geo[4];
	U.geometry(geo);
	const unsigned block = geo[0];
	size_t per_cu = (size_t)U.waves() * 256 / block;
	const size_t lds_blocks = (160u * 1024u) / (geo[2] + (paired ? 2u : 1u) * geo[3]);
	if(per_cu > lds_blocks) per_cu = lds_blocks;
	if(per_cu < 1) per_cu = 1;
	if(s->tune.blocks_per_cu > 0) per_cu = (size_t)s->tune.blocks_per_cu;
	size_t want = (s->n_reads + geo[1] - 1) / geo[1];
	size_t maxblocks = 256 * per_cu;
	if(big_main && maxblocks > 32) maxblocks = 32;        // ~1.3 MB of workspace per read in flight
	const unsigned grid = (unsigned)(want < 1 ? 1 : (want > maxblocks ? maxblocks : want));
	GoArgs A;
	memset(&A, 0, sizeof A);
	A.g = s->ix->dg; A.ref = s->ix->dr; A.ls = s->ix->dls; A.alts = s->ix->dalts;
	if(p->max_alts_tried) A.alts.maxAltsTried = p->max_alts_tried;          // --max-altstried
	if(p->use_haplotype && !linear && A.alts.n) A.alts.has_splice |= 2u;   // --haplotype: the table behind the ALTs is read (h2g_graph.h haps_of)
	A.rd1 = dreads(s); A.rd2 = A.rd1;
	if(paired) { A.rd2.codes = s->d_codes2; A.rd2.offs = s->d_offs2; A.rd2.quals = s->has_quals2 ? s->d_quals2 : nullptr; }
	A.P = aln_params_from(*p, p->no_spliced_alignment != 0, linear);
	if(!p->no_spliced_alignment) { A.ssdb = s->ix->dssdb; A.rdid_base = p->first_read_id; }
	if(!p->no_spliced_alignment) { A.P.sc.donor_sum = s->ix->d_spl[0]; A.P.sc.acc_sum1 = s->ix->d_spl[1]; A.P.sc.acc_sum2 = s->ix->d_spl[2]; }
	A.names1 = s->d_names; A.noffs1 = s->d_name_offs; A.names2 = s->d_names2; A.noffs2 = s->d_name_offs2;
	A.paired = paired ? 1u : 0u;
	if(!fast && s->st2_busy) { for(int k_ = 0; k_ < H2G_MSTREAMS_MAX; k_++) HIPCHK(hipStreamSynchronize(s->mst[k_])); s->st2_busy = false; }   // (the machine streams' pools are about to be used on the first stream)
	const unsigned M = s->mstreams, NB = M + 1;                    // machine passes in flight, and the depth of the per-run buffers
	// workgroups of one machine pass behind a fast pass: mach_total / M.  The default policy follows the batch: when the fast pass hands on more than 1.5 % of a linear index's
	// batch (repeat-structured sequence: the machine's passes, not the fast pass, are the step) the passes get 192 workgroups in all instead of 128 — measured on the repeat-structured
	// leg 44.7 -> 40.3 ms per step, while the fast-pass-bound legs (0.8 % handed on) lose 3 % with it (profiles/r05_NOTES.md §6).  The pools are sized for the larger share from the start.
	const bool mt_auto = s->tune.mach_total_auto != 0 && linear;
	const size_t units_ = paired ? s->n_reads : (s->n_reads + 1) / 2;
	const unsigned mach_total_now = mt_auto && (size_t)s->last_bails * 1000 > units_ * 15 ? 192u : s->tune.mach_total;
	unsigned mach_cap = mach_total_now / M;
	if(mach_cap > H2G_MACH_MAXGRID) mach_cap = H2G_MACH_MAXGRID;
	if(mach_cap < 1) mach_cap = 1;
	unsigned pool_cap = (mt_auto ? 192u : s->tune.mach_total) / M;      // what the pools are sized for
	if(pool_cap > H2G_MACH_MAXGRID) pool_cap = H2G_MACH_MAXGRID;
	if(pool_cap < mach_cap) pool_cap = mach_cap;
	const unsigned bgrid = fast ? 2u : 4u;                        // workgroups of a second pass (large workspace: ~5 MB per read in flight)
	{	// behind a fast pass the machine works on stream gen % M with that stream's pools, on at most mach_cap workgroups
		const size_t pgrid = fast && grid > pool_cap ? (size_t)pool_cap : (size_t)grid;
		if(fast && s->n_reads >= 200000) {
			// large batches (a streaming caller's): every machine stream's pools exist before the first pass that could need them — an allocation
			// in the middle of a queue of runs (gigabytes, synchronous) would stall all of them.  The stream's SECOND run pays for it, once (round 6: a caller with one
			// batch — a command line over a million pairs — never needs streams 1 .. M - 1: their 30 GB of pools were a second of its three).
			// (Small batches keep allocating a machine stream's pools when it is first used.)
			const GoUnit& Bp = go_unit(linear, true, spl);
			uint32_t bgeo_[4];
			Bp.geometry(bgeo_);
			GoArgs scratch = A;
			for(unsigned m_ = 0; m_ < M && s->gen >= 1; m_++) {
				if((rc = go_pool_for(s, 2 * (int)m_, U, pgrid * geo[1], pgrid * block, p->bowtie2_dp, &scratch))) return rc;
				if(!big_main && !s->tune.no_second_pass && (rc = go_pool_for(s, 2 * (int)m_ + 1, Bp, (size_t)bgrid * bgeo_[1], (size_t)bgrid * bgeo_[0], p->bowtie2_dp, &scratch))) return rc;
				// ... and its overflow list, and the stream itself (a HIP stream gets its hardware queue when it is first used)
				if(!s->d_ovf_list[m_]) { HIPCHK(hipMalloc((void**)&s->d_ovf_list[m_], (s->max_reads + 4) * 4)); HIPCHK(hipMemsetAsync(s->d_ovf_list[m_] + s->max_reads, 0, 16, s->mst[m_])); }
			}
			for(unsigned b_ = 0; b_ < M + 1; b_++) {      // the per-run buffers of every generation
				if(!s->d_bail_list[b_]) HIPCHK(hipMalloc((void**)&s->d_bail_list[b_], (s->max_reads + 4) * 4));
				if(!s->d_fast_args[b_]) HIPCHK(hipMalloc((void**)&s->d_fast_args[b_], sizeof(FastArgs)));
				if(!s->d_fast_args[H2G_NBUF + b_]) HIPCHK(hipMalloc((void**)&s->d_fast_args[H2G_NBUF + b_], sizeof(FastArgs)));
			}
		}
		if((rc = go_pool_for(s, fast ? 2 * (int)(s->gen % M) : 0, U, pgrid * geo[1], pgrid * block, p->bowtie2_dp, &A))) return rc;
	}
	memset(&A.O, 0, sizeof A.O);
	if(!paired) {
		if(!s->d_rout) HIPCHK(hipMalloc((void**)&s->d_rout, s->max_reads * sizeof(ReadOut)));
		const uint32_t slots = p->khits;
		if(s->aln_alloc < s->max_reads * (size_t)slots) {
			(void)hipFree(s->d_aln); s->d_aln = nullptr; s->aln_alloc = 0;
			HIPCHK(hipMalloc((void**)&s->d_aln, s->max_reads * (size_t)slots * sizeof(h2g_alnres)));
			s->aln_alloc = s->max_reads * (size_t)slots;
		}
		s->aln_slots = slots;
		A.O.rout = s->d_rout; A.O.aln = s->d_aln; A.O.aln_slots = slots;
	} else {
		if(!s->d_pout) HIPCHK(hipMalloc((void**)&s->d_pout, s->max_reads * sizeof(PairOut)));
		// a mate can report more alignments than -k before the pair is settled: 2 k + 4 slots, at least H2G_PAIR_RES_CAP
		uint32_t pslots = p->khits * 2 + 4;
		if(pslots < H2G_PAIR_RES_CAP) pslots = H2G_PAIR_RES_CAP;
		if(s->tune.pair_slots > 0) pslots = (uint32_t)s->tune.pair_slots;   // test knob: tiny rows push pairs through the overflow area (dense fetch only)
		if(s->paln_alloc < s->max_reads * (size_t)pslots) {
			for(int m = 0; m < 2; m++) { (void)hipFree(s->d_paln[m]); s->d_paln[m] = nullptr; }
			s->paln_alloc = 0;
			for(int m = 0; m < 2; m++) HIPCHK(hipMalloc((void**)&s->d_paln[m], s->max_reads * (size_t)pslots * sizeof(h2g_alnres)));
			s->paln_alloc = s->max_reads * (size_t)pslots;
		}
		s->pair_slots = pslots;
		A.O.pout = s->d_pout; A.O.paln[0] = s->d_paln[0]; A.O.paln[1] = s->d_paln[1]; A.O.pair_slots = pslots;
		// One part per machine stream: the machine passes of M queued runs may be in flight together and each takes its blocks
		// from its own cursor.  Block offsets (PairOut::pad) are relative to the whole area, so whichever pass wrote a pair last, its block is found.
		const size_t ovf_cap = s->max_reads / 4 > 65536 ? s->max_reads / 4 : 65536;   // records per part
		if(s->paln_ovf_cap < ovf_cap || s->paln_ovf_parts < M) {
			if(s->st2_busy) { for(int k_ = 0; k_ < H2G_MSTREAMS_MAX; k_++) HIPCHK(hipStreamSynchronize(s->mst[k_])); s->st2_busy = false; }
			HIPCHK(hipStreamSynchronize(s->st));
			(void)hipFree(s->d_paln_ovf); s->d_paln_ovf = nullptr; s->paln_ovf_cap = 0; s->paln_ovf_parts = 0;
			HIPCHK(hipMalloc((void**)&s->d_paln_ovf, M * ovf_cap * sizeof(h2g_alnres)));
			s->paln_ovf_cap = ovf_cap; s->paln_ovf_parts = M;
		}
		A.O.ovf = s->d_paln_ovf;
	}
	(void)hipGetLastError();
	// the fast pass's hand-on list, the counters and its argument block are buffered H2G_NBUF deep: the general machine's pass over
	// run k's hand-ons goes to machine stream k & 1 and may still be under way while the fast passes of runs k + 1 and k + 2 run
	const unsigned gsel = s->gen % NB, msel = s->gen % M;
	unsigned long long* const cblk = s->d_counters + H2G_CNT_BLOCK * gsel;
	HIPCHK(hipStreamWaitEvent(s->st, s->ev_mach[gsel], 0));          // run k - NB's machine pass: done with this set of buffers
	HIPCHK(hipMemsetAsync(cblk, 0, H2G_CNT_BLOCK * sizeof(unsigned long long), s->st));
	A.counters = cblk;
	A.work = reinterpret_cast<uint32_t*>(cblk + 14);
	if(paired) {   // the cursor starts at this machine stream's half of the area
		const uint32_t half = fast ? msel : 0u;
		A.O.ovf_cursor = reinterpret_cast<uint32_t*>(cblk + 124);
		A.O.ovf_cap = (uint32_t)((half + 1) * s->paln_ovf_cap);
		if(half) HIPCHK(hipMemsetD32Async((hipDeviceptr_t)A.O.ovf_cursor, (int)(half * s->paln_ovf_cap), 1, s->st));
	}
	A.list = nullptr; A.nlist = nullptr;
	if(s->tune.dbg_read >= 0) {
		if(!s->dbg_buf) HIPCHK(hipMalloc((void**)&s->dbg_buf, (1u << 20) * 4));      // (the stream's own: a device's memory, not the process's)
		HIPCHK(hipMemsetAsync(s->dbg_buf, 0, (1u << 20) * 4, s->st));
		A.dbg_buf = s->dbg_buf; A.dbg_read = (uint32_t)s->tune.dbg_read;
	}
	const int no_second = s->tune.no_second_pass;   // measurement / debugging knob
	const bool second = !big_main && !no_second;
	A.defer_overflow = second ? 1u : 0u;
	HIPCHK(hipEventRecord(s->ev[5], s->st));
	// ---- the fast pass (h2g_fast.h): the dominant traces with the per-read state on chip.  What it completes is final; the reads
	// it hands on (a device-side list, no host sync) are the general machine's batch.  Built for unspliced alignment on a linear
	// index with the default pair policy; every other option set goes to the machine whole.
	s->ran_fast = fast;
	unsigned fast_mgrid = 0, fast_orphan = 0, fast_dgrid = 0, fast_pool = 0; bool fast_mate_ho = false;
	if(fast) {
		uint32_t fgeo[5] = {0, 0, 0, 0, 0};
		const bool use_am = linear && paired && s->tune.align_mate != 0;
		if(!linear) h2g_go_fast_graph_geometry(fgeo); else if(use_am) h2g_go_fast_am_geometry(fgeo); else h2g_go_fast_geometry(fgeo);
		// CUs: one persistent fast workgroup each (LDS-bound), minus the few the machine pass of the PREVIOUS run may still hold
		// (the machine takes ~150 hand-ons per workgroup in half the time of a fast pass; the count is the last finished fast pass's)
		for(unsigned back = 1; back < NB && back <= s->gen; back++) {                           // the latest fast pass that is over
			const unsigned b = (s->gen - back) % NB;
			if(hipEventQuery(s->ev_bails[b]) == hipSuccess) { s->last_bails = s->h_bails[b]; break; }
		}
		(void)hipGetLastError();
		// hand-ons per machine workgroup (latency chains, two passes in flight).  Linear index: 400 — the fast pass is short and the machine's pass bounds the
		// step.  Graph index: the fast pass bounds the step and every machine workgroup costs it two of its 256 (one per CU), while the machine's pass only
		// has to end within two steps — so large batches give the machine less: 400 per workgroup up to 500 k batch units (pairs, or two unpaired reads), rising
		// to 1600 at 1 M and beyond.  Measured steady steps, SNP graphs over 8-256 Mbp (profiles/r04_graph_scale.jsonl, r04_graph_scale2.jsonl):
		//   500 k pairs: machine alone 55.3 ms | 400: 51.0 | 800: 55.2 | 1600: 70.6        1 M pairs (32 Mbp): alone 103.1 | 200: 110.0 | 400: 106.1 | 1600: 94.7 | 3200: 129.4
		//   2 M pairs: alone 194.6 | 1600: 171.1 | 3200: 176.4                               1 M pairs, 128 / 256 Mbp: alone 110.6 / 119.6 | 400: 113.7 / 122.7 | 1600: 101.5 / 109.2
		unsigned mach_div = s->tune.mach_div;
		if(mach_div == 0) {
			mach_div = 400;
			if(!linear) {
				const size_t units = paired ? (size_t)s->n_reads : (size_t)s->n_reads / 2;
				if(units >= 1000000) mach_div = 1600;
				else if(units > 500000) mach_div = 400 + (unsigned)((units - 500000) * 1200 / 500000);
			}
		}
		const unsigned mach_min = s->tune.mach_min;
		unsigned mgrid = (unsigned)((s->last_bails + mach_div - 1) / (mach_div ? mach_div : 1u));
		if(mgrid < mach_min) mgrid = mach_min;
		if(mgrid > mach_cap) mgrid = mach_cap;
		// (Tried on repeat-rich sequence, 27 000 hand-ons: a share that follows the measured work of the two passes, m = 128 M / (F + M) — the machine's
		// pass stayed a latency chain, 78 -> 66 ms on twice the workgroups, while the fast pass went 14.5 -> 49 ms on what was left; and the second pass
		// on a stream of its own — the extra queues cost the common case 13 -> 20 ms per run.  Neither ships: profiles/r04_NOTES.md §6.)
		size_t fwant = (s->n_reads + 127) / 128;                                                // small batches spread over the chip
		unsigned reserve = s->tune.fast_reserve >= 0 ? (unsigned)s->tune.fast_reserve : M * mgrid;  // (M machine passes may be in flight)
		if(reserve > 192) reserve = 192;
		const unsigned fmax = 256 - reserve;
		const unsigned fgrid = (unsigned)(fwant < 1 ? 1 : (fwant > fmax ? fmax : fwant));
		const size_t slot_bytes = (size_t)256 * fgeo[2] * fgeo[3];
		// the end of the batch goes to a drain launch (h2g_k_go_fast.hip) when the batch is large enough to have one
		const size_t units_f = paired ? (size_t)s->n_reads : ((size_t)s->n_reads + 1) / 2;
		uint32_t orphan_T = s->tune.orphan >= 0 ? (uint32_t)s->tune.orphan : (units_f >= H2G_ORPHAN_MIN_UNITS ? (uint32_t)H2G_ORPHAN_T : 0u);
		if(orphan_T > fgeo[2]) orphan_T = fgeo[2];
		if(!linear && fgeo[4] != 0) orphan_T = 0;                     // (a graph unit with GraphWS in global memory: its scratch is sized for one launch)
		const unsigned dgrid = (unsigned)(s->tune.drain_grid < 1 ? 1 : s->tune.drain_grid);
		// pairs that need alignMate (hi_aligner.h:5579): with a drain launch there, the fast launch parks them and the drain launch is the alignMate build of the pass (k_go_fast_am_drain)
		// — the hot loop keeps the lighter build, the machine sees neither them nor the tail (h2g_stream_tune "mate_handover": -1 this policy, 0 off, 1 on)
		const bool mate_ho = orphan_T != 0 && linear && paired && !use_am && s->tune.mate_handover != 0;
		uint32_t dgeo[5] = {fgeo[0], fgeo[1], fgeo[2], fgeo[3], fgeo[4]};
		if(mate_ho) h2g_go_fast_am_geometry(dgeo);
		const unsigned psel_f = orphan_T ? s->gen % H2G_FAST_POOLS : 0u;
		{	// every buffer of the two launches exists (and has been written once: fresh device memory costs its first writer the mapping) before the first run that could need it
			const size_t ocap = (size_t)256 * fgeo[2], dbytes = (size_t)dgrid * dgeo[2] * dgeo[3];
			const size_t dsc = linear ? 0 : (size_t)dgrid * fgeo[0] * (size_t)(2 * H2G_COMBINE_MAXLEN * sizeof(int64_t));
			bool need = s->fast_slot_bytes[psel_f] < slot_bytes;
			if(orphan_T) {
				for(unsigned p_ = 0; p_ < H2G_FAST_POOLS; p_++) need = need || s->fast_slot_bytes[p_] < slot_bytes;
				need = need || s->drain_slot_bytes < dbytes || s->drain_sc_bytes < dsc;
				for(unsigned b_ = 0; b_ < NB; b_++) need = need || !s->d_orphans[b_] || !s->d_fast_args[H2G_NBUF + b_];
				need = need || s->orphan_cap < ocap;
			}
			if(need) {
				HIPCHK(hipStreamSynchronize(s->st)); HIPCHK(hipStreamSynchronize(s->dst)); for(int k_ = 0; k_ < H2G_MSTREAMS_MAX; k_++) HIPCHK(hipStreamSynchronize(s->mst[k_]));
				auto grow = [&](void** ptr, size_t* have, size_t want_) -> int {
					if(*have >= want_) return H2G_OK;
					(void)hipFree(*ptr); *ptr = nullptr; *have = 0;
					if(want_) { HIPCHK(hipMalloc(ptr, want_)); HIPCHK(hipMemsetAsync(*ptr, 0, want_, s->st)); }
					*have = want_;
					return H2G_OK;
				};
				for(unsigned p_ = 0; p_ < (orphan_T ? (unsigned)H2G_FAST_POOLS : 1u); p_++) if((rc = grow((void**)&s->d_fast_slots[orphan_T ? p_ : psel_f], &s->fast_slot_bytes[orphan_T ? p_ : psel_f], slot_bytes))) return rc;
				if(orphan_T) {
					if(s->orphan_cap < ocap) { for(int b_ = 0; b_ < H2G_NBUF; b_++) { (void)hipFree(s->d_orphans[b_]); s->d_orphans[b_] = nullptr; } s->orphan_cap = ocap; }
					for(unsigned b_ = 0; b_ < NB; b_++) {
						if(!s->d_orphans[b_]) { HIPCHK(hipMalloc((void**)&s->d_orphans[b_], (s->orphan_cap + 4) * 4)); HIPCHK(hipMemsetAsync(s->d_orphans[b_], 0, (s->orphan_cap + 4) * 4, s->st)); }
						if(!s->d_fast_args[H2G_NBUF + b_]) HIPCHK(hipMalloc((void**)&s->d_fast_args[H2G_NBUF + b_], sizeof(FastArgs)));
					}
					if((rc = grow((void**)&s->d_drain_slots, &s->drain_slot_bytes, dbytes))) return rc;
					if((rc = grow((void**)&s->d_drain_sc, &s->drain_sc_bytes, dsc))) return rc;
				}
				HIPCHK(hipStreamSynchronize(s->st));
			}
		}
		// run k - 3's drain launch is done with this pool — and a run WITHOUT a drain launch (a small resident batch queued behind a large one) uses pool 0, which the drain launch
		// of an earlier run may still be reading: every fast launch waits for the last reader of its pool
		HIPCHK(hipStreamWaitEvent(s->st, s->ev_pool[psel_f], 0));
		if(!s->d_bail_list[gsel]) HIPCHK(hipMalloc((void**)&s->d_bail_list[gsel], (s->max_reads + 4) * 4));
		uint32_t* const bl = s->d_bail_list[gsel];
		HIPCHK(hipMemsetAsync(bl + s->max_reads, 0, 16, s->st));
		FastArgs F;
		memset(&F, 0, sizeof F);
		F.g = A.g; F.ref = A.ref; F.ls = A.ls; F.rd1 = A.rd1; F.rd2 = A.rd2; F.P = A.P;
		F.names1 = A.names1; F.noffs1 = A.noffs1; F.names2 = A.names2; F.noffs2 = A.noffs2;
		F.slots = s->d_fast_slots[psel_f];
		F.O.rout = A.O.rout; F.O.aln = A.O.aln; F.O.aln_slots = A.O.aln_slots; F.O.pout = A.O.pout; F.O.paln[0] = A.O.paln[0]; F.O.paln[1] = A.O.paln[1]; F.O.pair_slots = A.O.pair_slots;
		F.counters = cblk; F.work = reinterpret_cast<uint32_t*>(cblk + 12);
		F.bail_list = bl; F.bail_count = bl + s->max_reads;
		F.total = (uint32_t)s->n_reads; F.paired = paired ? 1u : 0u;
		F.alts = A.alts; F.gws_base = nullptr; F.gws_stride = 0; F.sc_base = nullptr;
		F.dbg_read = A.dbg_read; F.dbg_buf = A.dbg_buf;
		// (single-end batches and graph indexes: there the machine's pass is the longer of the two already — NOTES §1, §3: 47 -> 53 ms per 500 k graph pairs with it)
		F.tail = linear && paired && s->tune.tail > 0 ? (uint32_t)s->tune.tail : 0u;
		if(!linear) {   // per-lane scratch of the graph primitives (every CU may hold a workgroup)
			const size_t lanes = (size_t)256 * fgeo[0];
			const size_t gws_bytes = lanes * fgeo[4], sc_bytes = lanes * (size_t)(2 * H2G_COMBINE_MAXLEN * sizeof(int64_t));
			if(s->fast_gws_bytes < gws_bytes) {
				(void)hipFree(s->d_fast_gws); s->d_fast_gws = nullptr; s->fast_gws_bytes = 0;
				HIPCHK(hipMalloc((void**)&s->d_fast_gws, gws_bytes));
				s->fast_gws_bytes = gws_bytes;
			}
			if(s->fast_sc_bytes < sc_bytes) {
				(void)hipFree(s->d_fast_sc); s->d_fast_sc = nullptr; s->fast_sc_bytes = 0;
				HIPCHK(hipMalloc((void**)&s->d_fast_sc, sc_bytes));
				s->fast_sc_bytes = sc_bytes;
			}
			F.gws_base = s->d_fast_gws; F.gws_stride = fgeo[4]; F.sc_base = s->d_fast_sc;
		}
		if(!s->d_fast_args[gsel]) HIPCHK(hipMalloc((void**)&s->d_fast_args[gsel], sizeof(FastArgs)));
		FastArgs D = F;
		if(orphan_T) {
			uint32_t* const ol = s->d_orphans[gsel];
			HIPCHK(hipMemsetAsync(ol + s->orphan_cap, 0, 16, s->st));
			F.orphan_T = orphan_T; F.orphan_list = ol; F.orphan_count = ol + s->orphan_cap; F.mate_handover = mate_ho ? 1u : 0u;
			D.adopt_slot_words = fgeo[3] / 4;
			// The drain launch's own tail (the last reads of each of its workgroups go to the machine).  With few hand-ons the machine's passes are short and the tail is a third of
			// what they get: the drain launch finishes its reads itself (lease ZE, random genome: 11.48 -> 11.16 ms per step, hand-ons 2 830 -> 1 880); where the machine's passes are
			// the step (more than 1.5 % handed on: the policy of mach_total) they absorb the tail for nothing and a drain launch that runs to its last read would be the longer one
			// (repeat-structured: 37.8 -> 39.0 ms with it): the tail stays.  A caller's own setting (H2G_FAST_TAIL, "tail") is kept as it is.
			if(s->tune.tail_auto) D.tail = linear && paired && (size_t)s->last_bails * 1000 > units_f * 15 ? (uint32_t)H2G_DEFAULT_TAIL : 0u;
			D.slots = s->d_drain_slots;
			D.adopt_list = ol; D.adopt_count = ol + s->orphan_cap; D.adopt_slots = F.slots;
			D.work = reinterpret_cast<uint32_t*>(cblk + 13);
			D.cnt_off = 120;                                             // its own rank / side / step / aligned counters: [240..243]
			D.total = 0;
			if(!linear) D.sc_base = s->d_drain_sc;
			s->orph_cur = ol + s->orphan_cap;
		}
		// through pinned memory: a pageable source would make this call wait for everything queued on the stream (the previous run's fast
		// pass), and the chip would idle while the host queues this run.  The staging block of this buffer set was last read by the upload
		// of run k - H2G_NBUF, which is over once that run's fast pass is
		if(s->gen >= NB && hipEventQuery(s->ev_fast[gsel]) != hipSuccess) { (void)hipGetLastError(); HIPCHK(hipEventSynchronize(s->ev_fast[gsel])); }
		FastArgs* const hF = reinterpret_cast<FastArgs*>(s->h_fast_args) + gsel;
		*hF = F;
		HIPCHK(hipMemcpyAsync(s->d_fast_args[gsel], hF, sizeof F, hipMemcpyHostToDevice, s->st));
		if(orphan_T) {
			FastArgs* const hD = reinterpret_cast<FastArgs*>(s->h_fast_args) + H2G_NBUF + gsel;
			*hD = D;
			HIPCHK(hipMemcpyAsync(s->d_fast_args[H2G_NBUF + gsel], hD, sizeof D, hipMemcpyHostToDevice, s->st));
		}
		if((!linear ? h2g_go_fast_graph_launch : use_am ? h2g_go_fast_am_launch : h2g_go_fast_launch)(reinterpret_cast<const FastArgs*>(s->d_fast_args[gsel]), fgrid, s->st) != 0) return set_err("go() fast pass launch", hipGetLastError());
		A.list = bl; A.nlist = bl + s->max_reads;
		fast_mgrid = mgrid;
		fast_orphan = orphan_T; fast_dgrid = dgrid; fast_pool = psel_f; fast_mate_ho = mate_ho;
	}
	HIPCHK(hipEventRecord(s->ev[10], s->st));
	// behind a fast pass the machine works on the second stream (a short list on few workgroups: the next run's fast pass does not wait for it)
	hipStream_t ms = s->st;
	unsigned mach_grid = grid;
	s->ran_drain = false;
	if(fast) {
		HIPCHK(hipEventRecord(s->ev_fast[gsel], s->st));
		ms = s->mst[msel]; s->st2_busy = true;
		if(fast_mgrid < mach_grid) mach_grid = fast_mgrid;
		if(fast_orphan) {
			// the drain launch: the reads the fast launch's workgroups left in flight, on the drain stream next to the following run's fast launch; the machine's pass needs its hand-ons too
			const bool use_am = (linear && paired && s->tune.align_mate != 0) || fast_mate_ho;
			HIPCHK(hipStreamWaitEvent(s->dst, s->ev_fast[gsel], 0));
			HIPCHK(hipEventRecord(s->ev_dr[0], s->dst));
			if((!linear ? h2g_go_fast_graph_launch_drain : use_am ? h2g_go_fast_am_launch_drain : h2g_go_fast_launch_drain)(reinterpret_cast<const FastArgs*>(s->d_fast_args[H2G_NBUF + gsel]), fast_dgrid, s->dst) != 0) return set_err("go() drain launch", hipGetLastError());
			HIPCHK(hipEventRecord(s->ev_dr[1], s->dst));
			HIPCHK(hipEventRecord(s->ev_pool[fast_pool], s->dst));
			HIPCHK(hipMemcpyAsync(&s->h_bails[gsel], s->d_bail_list[gsel] + s->max_reads, 4, hipMemcpyDeviceToHost, s->dst));
			HIPCHK(hipEventRecord(s->ev_drain[gsel], s->dst));
			HIPCHK(hipEventRecord(s->ev_bails[gsel], s->dst));
			HIPCHK(hipStreamWaitEvent(ms, s->ev_drain[gsel], 0));
			s->ran_drain = true; s->dst_busy = true;
		} else {
			HIPCHK(hipMemcpyAsync(&s->h_bails[gsel], s->d_bail_list[gsel] + s->max_reads, 4, hipMemcpyDeviceToHost, s->st));
			HIPCHK(hipEventRecord(s->ev_bails[gsel], s->st));
			HIPCHK(hipStreamWaitEvent(ms, s->ev_fast[gsel], 0));
		}
	}
	const unsigned psel = fast ? msel : 0u;          // workspace pools and the overflow list of this machine stream
	s->ovf_cur = psel;
	if(!s->d_ovf_list[psel]) HIPCHK(hipMalloc((void**)&s->d_ovf_list[psel], (s->max_reads + 4) * 4));
	uint32_t* const ovl = s->d_ovf_list[psel];
	HIPCHK(hipMemsetAsync(ovl + s->max_reads, 0, 16, ms));
	{	// the long-edit area: part psel and its cursor belong to this machine stream (the pass that used them last is over: same stream)
		const size_t lcap = s->max_reads > (1u << 20) ? s->max_reads : (size_t)(1u << 20);
		if(s->ledits_cap < lcap || s->ledits_parts < M) {
			if(s->st2_busy) { for(int k_ = 0; k_ < H2G_MSTREAMS_MAX; k_++) HIPCHK(hipStreamSynchronize(s->mst[k_])); }
			HIPCHK(hipStreamSynchronize(s->st));
			(void)hipFree(s->d_ledits); s->d_ledits = nullptr; s->ledits_cap = 0; s->ledits_parts = 0;
			HIPCHK(hipMalloc((void**)&s->d_ledits, M * lcap * sizeof(h2g_edit)));
			if(!s->d_ledits_cur) HIPCHK(hipMalloc((void**)&s->d_ledits_cur, H2G_MSTREAMS_MAX * 4));
			s->ledits_cap = lcap; s->ledits_parts = M;
			for(unsigned m_ = 0; m_ < H2G_MSTREAMS_MAX; m_++) HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(s->d_ledits_cur + m_), (int)(m_ < M ? m_ * lcap : 0), 1, s->st));
			HIPCHK(hipStreamSynchronize(s->st));
		}
		A.O.ledits = s->d_ledits; A.O.ledits_cursor = s->d_ledits_cur + psel; A.O.ledits_cap = (uint32_t)((psel + 1) * s->ledits_cap);
		HIPCHK(hipMemsetD32Async((hipDeviceptr_t)A.O.ledits_cursor, (int)(psel * s->ledits_cap), 1, ms));
		s->ledits_touched |= 1u << psel;
	}
	if(second) { A.O.defer_list = ovl; A.O.defer_count = ovl + s->max_reads; }     // overflowed reads: listed for the second pass, their rows untouched
	if(fast && s->n_reads >= 200000 && s->gen >= 1 && !s->mstreams_warm) {
		// A machine stream's FIRST kernels cost it ~64 ms on top of their own time (lease I of round 6, `profiles/r06_i_batches_first.jsonl`: runs 1-7 of a stream 94 ms, every
		// later one 30; the kernels' own events read 10 + 20 ms throughout — the queue's scratch memory and code objects are set up when a queue first needs them).  A streaming
		// caller's SECOND run therefore runs both machine kernels once on every other machine stream (the first run may be the only one), over an empty list: with the pools, lists and per-run buffers above this is
		// everything a stream does for the first time (a bench that warms up for 5 steps then timed the first use of streams 5-7: 18.5 ms per step where the steady state is 12-13).
		if(!s->d_warm_cnt) { HIPCHK(hipMalloc((void**)&s->d_warm_cnt, H2G_CNT_BLOCK * sizeof(unsigned long long))); HIPCHK(hipMemset(s->d_warm_cnt, 0, H2G_CNT_BLOCK * sizeof(unsigned long long))); }
		const GoUnit& Bw = go_unit(linear, true, spl);
		uint32_t wgeo[4];
		Bw.geometry(wgeo);
		for(unsigned m_ = 0; m_ < M; m_++) {
			if(m_ == msel || !s->d_ovf_list[m_]) continue;
			GoArgs W1 = A;
			if((rc = go_pool_for(s, 2 * (int)m_, U, (size_t)geo[1], (size_t)block, p->bowtie2_dp, &W1))) return rc;      // (the stream's pools exist: nothing is allocated here)
			W1.counters = s->d_warm_cnt; W1.work = reinterpret_cast<uint32_t*>(s->d_warm_cnt + 14);
			W1.list = s->d_ovf_list[m_]; W1.nlist = s->d_ovf_list[m_] + s->max_reads;                                       // a count of zero (memset on this stream above)
			W1.O.ovf_cursor = reinterpret_cast<uint32_t*>(s->d_warm_cnt + 124); W1.O.ledits_cursor = reinterpret_cast<uint32_t*>(s->d_warm_cnt + 126);
			W1.defer_overflow = 0; W1.O.defer_list = nullptr; W1.O.defer_count = nullptr;
			if(U.launch(&W1, 1, s->mst[m_]) != 0) return set_err("go() warm-up launch", hipGetLastError());
			if(second) {
				GoArgs W2 = W1;
				if((rc = go_pool_for(s, 2 * (int)m_ + 1, Bw, (size_t)wgeo[1], (size_t)wgeo[0], p->bowtie2_dp, &W2))) return rc;
				W2.work = reinterpret_cast<uint32_t*>(s->d_warm_cnt + 15);
				if(Bw.launch(&W2, 1, s->mst[m_]) != 0) return set_err("go() warm-up launch", hipGetLastError());
			}
		}
		s->st2_busy = true;
		s->mstreams_warm = true;
	}
	HIPCHK(hipEventRecord(s->ev[7], ms));
	if(U.launch(&A, mach_grid, ms) != 0) return set_err("go() launch", hipGetLastError());
	HIPCHK(hipEventRecord(s->ev[6], ms));
	if(second) {
		const GoUnit& B = go_unit(linear, true, spl);
		uint32_t* cnt = ovl + s->max_reads;       // (filled by the main pass itself: MachOut::defer_list)
		uint32_t bgeo[4];
		B.geometry(bgeo);
		GoArgs A2 = A;
		if((rc = go_pool_for(s, 2 * psel + 1, B, (size_t)bgrid * bgeo[1], (size_t)bgrid * bgeo[0], p->bowtie2_dp, &A2))) return rc;
		A2.counters = cblk + 256;           // (a region of its own: in H2G_GO_PROF builds a pass writes up to 96 words behind its counters)
		A2.work = reinterpret_cast<uint32_t*>(cblk + 15);
		A2.list = ovl; A2.nlist = cnt;
		A2.defer_overflow = 0; A2.O.defer_list = nullptr; A2.O.defer_count = nullptr;
		if(B.launch(&A2, bgrid, ms) != 0) return set_err("go() second pass launch", hipGetLastError());
	}
	HIPCHK(hipEventRecord(s->ev[8], ms));
	if(fast) HIPCHK(hipEventRecord(s->ev_mach[gsel], ms));
	s->cnt_cur = cblk; s->gen++;
	HIPCHK(hipGetLastError());
	s->ran_align = true;
	return H2G_OK;
}

extern "C" h2g_status h2g_align_run(h2g_stream* s, const h2g_align_params* p) { return go_run(s, p, false); }
extern "C" h2g_status h2g_align_pairs_run(h2g_stream* s, const h2g_align_params* p) { return go_run(s, p, true); }

extern "C" h2g_status h2g_align_fetch(h2g_stream* s, h2g_read_result* res, h2g_alnres* aln, size_t first, size_t n) {
	if(s && s->st2_busy) { for(int k_ = 0; k_ < H2G_MSTREAMS_MAX; k_++) HIPCHK(hipStreamSynchronize(s->mst[k_])); s->st2_busy = false; }   // (results of the machine pass on the second stream)
	if(!s || !res || first + n > s->n_reads || !s->d_rout) return H2G_ERR_ARG;
	std::vector<ReadOut> tmp(n);
	HIPCHK(hipMemcpyAsync(tmp.data(), s->d_rout + first, n * sizeof(ReadOut), hipMemcpyDeviceToHost, s->st));
	if(aln && n) {   // device rows hold aln_slots (= -k of the run) records, the caller's rows H2G_ALN_CAP: copy what fits in both
		const uint32_t w = s->aln_slots < H2G_ALN_CAP ? s->aln_slots : H2G_ALN_CAP;
		HIPCHK(hipMemcpy2DAsync(aln, (size_t)H2G_ALN_CAP * sizeof(h2g_alnres), s->d_aln + first * s->aln_slots, (size_t)s->aln_slots * sizeof(h2g_alnres),
		                        (size_t)w * sizeof(h2g_alnres), n, hipMemcpyDeviceToHost, s->st));
	}
	HIPCHK(sync_all(s));
	for(size_t i = 0; i < n; i++) {
		res[i].nres = tmp[i].nres; res[i].nselect = tmp[i].nselect; res[i].overflow = tmp[i].overflow;
		res[i].nrank = tmp[i].nrank; res[i].nsteps = tmp[i].nsteps; res[i].depth = tmp[i].depth;
		res[i].best = tmp[i].best; res[i].secbest = tmp[i].secbest; res[i].best_h2 = tmp[i].best_h2; res[i].secbest_h2 = tmp[i].secbest_h2;
	}
	return H2G_OK;
}

// ------------------------------------------------------------------------------------------ paired go(): fetch
static h2g_status pairs_fetch_rows(h2g_stream* s, h2g_pair_result* res, h2g_alnres* aln1, h2g_alnres* aln2, size_t first, size_t n, bool callers_view) {
	if(s && s->st2_busy) { for(int k_ = 0; k_ < H2G_MSTREAMS_MAX; k_++) HIPCHK(hipStreamSynchronize(s->mst[k_])); s->st2_busy = false; }   // (results of the machine pass on the second stream)
	if(!s || !res || first + n > s->n_reads || !s->d_pout) return H2G_ERR_ARG;
	if((aln1 || aln2) && s->pair_slots < H2G_PAIR_RES_CAP) return H2G_ERR_ARG;
	HIPCHK(hipMemcpyAsync(res, s->d_pout + first, n * sizeof(PairOut), hipMemcpyDeviceToHost, s->st));
	// device rows hold pair_slots records, the caller's rows H2G_PAIR_RES_CAP (the dense variant returns all of them)
	for(int m = 0; m < 2 && n; m++) {
		h2g_alnres* dst = m == 0 ? aln1 : aln2;
		if(dst) HIPCHK(hipMemcpy2DAsync(dst, (size_t)H2G_PAIR_RES_CAP * sizeof(h2g_alnres), s->d_paln[m] + first * s->pair_slots, (size_t)s->pair_slots * sizeof(h2g_alnres),
		                                (size_t)H2G_PAIR_RES_CAP * sizeof(h2g_alnres), n, hipMemcpyDeviceToHost, s->st));
	}
	HIPCHK(sync_all(s));
	// a pair kept in the overflow area has more records than these fixed rows return: flagged in the returned copy (the dense variant
	// returns every record)
	if(callers_view) for(size_t i = 0; i < n; i++) if(res[i].pad) { res[i].overflow |= 4; res[i].pad = 0; }      // (the block offset itself is the device's business: allocation order, i.e. timing)
	return H2G_OK;
}
extern "C" h2g_status h2g_align_pairs_fetch(h2g_stream* s, h2g_pair_result* res, h2g_alnres* aln1, h2g_alnres* aln2, size_t first, size_t n) { return pairs_fetch_rows(s, res, aln1, aln2, first, n, true); }

// ------------------------------------------------------------------------------------------ dense result fetch
// The slot layout of h2g_align_fetch moves H2G_ALN_CAP x 424 B per read over PCIe whatever was found; these variants gather
// only the records that exist into one dense array on the device (one lane per read) and copy that.
__global__ __launch_bounds__(256) void k_gather_aln(const h2g_alnres* src, uint32_t slots, const uint32_t* cnt, uint32_t cnt_stride,
                                                    const unsigned long long* offs, size_t n, h2g_alnres* dst,
                                                    const PairOut* pout = nullptr, const h2g_alnres* ovf = nullptr, int mate = 0)
{
	const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	if(i >= n) return;
	uint32_t c = cnt[i * cnt_stride];
	const h2g_alnres* a = src + i * slots;
	if(pout && pout[i].pad) a = ovf + (pout[i].pad - 1u) + (mate ? pout[i].nres[0] : 0u);      // every record of the pair lives in its overflow block
	else if(c > slots) c = slots;
	h2g_alnres* d = dst + offs[i];
	for(uint32_t k = 0; k < c; k++) {
		d[k].fw = a[k].fw; d[k].tidx = a[k].tidx; d[k].toff = a[k].toff; d[k].len = a[k].len; d[k].trim5 = a[k].trim5; d[k].trim3 = a[k].trim3;
		d[k].nedits = a[k].nedits; d[k].splicescore = a[k].splicescore; d[k].score = a[k].score;
		for(uint32_t e = 0; e < a[k].nedits && e < H2G_MAX_EDITS; e++) d[k].edits[e] = a[k].edits[e];
	}
}
// dense offsets of [first, first+n) of a slot array from the per-read record counts already on the host
static uint64_t dense_offsets(const uint32_t* h_cnt, uint32_t h_stride, uint32_t slots, size_t n, uint64_t* offs, const uint32_t* h_pad = nullptr) {
	uint64_t tot = 0;
	for(size_t i = 0; i < n; i++) { offs[i] = tot; const uint32_t c = h_cnt[i * h_stride]; tot += (c < slots || (h_pad && h_pad[i * h_stride])) ? c : slots; }
	offs[n] = tot;
	return tot;
}
// gathers the records into `out` (host) at the offsets computed by dense_offsets
static int gather_dense(h2g_stream* s, const h2g_alnres* d_src, uint32_t slots, const uint32_t* d_cnt, uint32_t cnt_stride, size_t n, h2g_alnres* out,
                        const uint64_t* offs, int tmp_slot, const PairOut* d_pout = nullptr, int mate = 0)
{
	const uint64_t tot = offs[n];
	if(tot == 0) return H2G_OK;
	void *d_offs = nullptr, *d_dense = nullptr;
	int rc;
	if((rc = tmp_buf(s, tmp_slot, (n + 1) * 8, &d_offs)) || (rc = tmp_buf(s, tmp_slot + 1, tot * sizeof(h2g_alnres), &d_dense))) return rc;
	HIPCHK(hipMemcpyAsync(d_offs, offs, (n + 1) * 8, hipMemcpyHostToDevice, s->st));
	hipLaunchKernelGGL(k_gather_aln, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->st, d_src, slots, d_cnt, cnt_stride,
	                   (const unsigned long long*)d_offs, n, (h2g_alnres*)d_dense, d_pout, (const h2g_alnres*)s->d_paln_ovf, mate);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(out, d_dense, tot * sizeof(h2g_alnres), hipMemcpyDeviceToHost, s->st));
	HIPCHK(sync_all(s));
	return H2G_OK;
}

extern "C" h2g_status h2g_align_fetch_dense(h2g_stream* s, h2g_read_result* res, h2g_alnres* aln, size_t aln_cap, uint64_t* aln_offs, size_t first, size_t n) {
	if(s && s->st2_busy) { for(int k_ = 0; k_ < H2G_MSTREAMS_MAX; k_++) HIPCHK(hipStreamSynchronize(s->mst[k_])); s->st2_busy = false; }   // (results of the machine pass on the second stream)
	if(!s || !res || !aln || !aln_offs || first + n > s->n_reads || !s->d_rout) return H2G_ERR_ARG;
	const h2g_status rc = h2g_align_fetch(s, res, nullptr, first, n);
	if(rc != H2G_OK) return rc;
	static_assert(offsetof(ReadOut, nselect) == 4, "ReadOut layout");
	if(dense_offsets(&res[0].nselect, sizeof(h2g_read_result) / 4, s->aln_slots, n, aln_offs) > aln_cap) return H2G_ERR_ARG;
	return gather_dense(s, s->d_aln + first * s->aln_slots, s->aln_slots, reinterpret_cast<const uint32_t*>(s->d_rout + first) + 1, sizeof(ReadOut) / 4,
	                    n, aln, aln_offs, 0);
}

extern "C" h2g_status h2g_align_pairs_fetch_dense(h2g_stream* s, h2g_pair_result* res, h2g_alnres* aln1, size_t cap1, uint64_t* offs1,
                                                  h2g_alnres* aln2, size_t cap2, uint64_t* offs2, size_t first, size_t n)
{
	if(s && s->st2_busy) { for(int k_ = 0; k_ < H2G_MSTREAMS_MAX; k_++) HIPCHK(hipStreamSynchronize(s->mst[k_])); s->st2_busy = false; }   // (results of the machine pass on the second stream)
	if(!s || !res || !aln1 || !aln2 || !offs1 || !offs2 || first + n > s->n_reads || !s->d_pout) return H2G_ERR_ARG;
	const h2g_status rc = pairs_fetch_rows(s, res, nullptr, nullptr, first, n, false);       // (the headers as the device holds them: `pad` says where a pair's records are)
	if(rc != H2G_OK) return rc;
	static_assert(offsetof(PairOut, nres) == 0, "PairOut layout");
	// both totals are known before either capacity is judged, so a caller that has to grow its buffers learns both needs at once
	static_assert(offsetof(h2g_pair_result, pad) == offsetof(PairOut, pad) && sizeof(h2g_pair_result) == sizeof(PairOut), "PairOut layout");
	const uint64_t t1 = dense_offsets(&res[0].nres[0], sizeof(h2g_pair_result) / 4, s->pair_slots, n, offs1, &res[0].pad);
	const uint64_t t2 = dense_offsets(&res[0].nres[1], sizeof(h2g_pair_result) / 4, s->pair_slots, n, offs2, &res[0].pad);
	if(t1 > cap1 || t2 > cap2) return H2G_ERR_ARG;
	int r;
	if((r = gather_dense(s, s->d_paln[0] + first * s->pair_slots, s->pair_slots, reinterpret_cast<const uint32_t*>(s->d_pout + first), sizeof(PairOut) / 4,
	                     n, aln1, offs1, 0, s->d_pout + first, 0))) return r;
	r = gather_dense(s, s->d_paln[1] + first * s->pair_slots, s->pair_slots, reinterpret_cast<const uint32_t*>(s->d_pout + first) + 1, sizeof(PairOut) / 4,
	                 n, aln2, offs2, 2, s->d_pout + first, 1);
	// `pad` is where the device kept the pair's records (a block offset in the overflow area: allocation order, i.e. timing) — nothing a caller may see
	for(size_t i = 0; i < n; i++) res[i].pad = 0;
	return r;
}

// ------------------------------------------------------------------------------------------ compact result fetch
// The dense fetch still moves 424 B per record (40 B of fields + 32 edit entries of which an alignment of a 101 bp read uses one or two) and needs the
// result headers on the host before it can size anything (two round trips).  The compact fetch sizes, scans and gathers on the device: a record travels as its
// 40 B of fields + 12 B per edit it holds (a long record: its marker entry), 8-byte aligned — a tenth of the bytes — and the host gets byte offsets per read.
// A compact record is a PREFIX of an h2g_alnres: read it through a `const h2g_alnres*`, never copy the struct.
__device__ __forceinline__ uint32_t compact_rec_bytes(uint32_t nedits) { const uint32_t e = nedits > H2G_MAX_EDITS ? 1u : nedits; return (40u + 12u * e + 7u) & ~7u; }
__device__ __forceinline__ const h2g_alnres* compact_src(const h2g_alnres* src, uint32_t slots, size_t i, uint32_t* c, const PairOut* pout, const h2g_alnres* ovf, int mate) {
	const h2g_alnres* a = src + i * slots;
	if(pout && pout[i].pad) a = ovf + (pout[i].pad - 1u) + (mate ? pout[i].nres[0] : 0u);      // every record of the pair lives in its overflow block
	else if(*c > slots) *c = slots;
	return a;
}
__global__ __launch_bounds__(256) void k_compact_sizes(const h2g_alnres* src, uint32_t slots, const uint32_t* cnt, uint32_t cnt_stride, size_t n, unsigned long long* sizes,
                                                       const PairOut* pout, const h2g_alnres* ovf, int mate)
{
	const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	if(i > n) return;
	if(i == n) { sizes[n] = 0; return; }
	uint32_t c = cnt[i * cnt_stride];
	const h2g_alnres* a = compact_src(src, slots, i, &c, pout, ovf, mate);
	unsigned long long b = 0;
	for(uint32_t k = 0; k < c; k++) b += compact_rec_bytes(a[k].nedits);
	sizes[i] = b;
}
// exclusive prefix sum of v[0 .. n1) in place, three launches: tiles of 2048 entries scanned by a workgroup each (8 consecutive entries per thread), their sums scanned by
// one workgroup, then added back.  (The first version — one workgroup walking 1 M entries with a stride per thread — took 3 ms of a 48 ms round trip.)
#define H2G_SCAN_TILE 2048u
__global__ __launch_bounds__(256) void k_scan_tiles(unsigned long long* v, size_t n1, unsigned long long* tile_sum) {
	__shared__ unsigned long long part[256];
	const size_t base = (size_t)blockIdx.x * H2G_SCAN_TILE + (size_t)threadIdx.x * 8;
	unsigned long long x[8], sum = 0;
#pragma unroll
	for(int k = 0; k < 8; k++) { x[k] = base + k < n1 ? v[base + k] : 0; sum += x[k]; }
	part[threadIdx.x] = sum;
	__syncthreads();
	for(unsigned d = 1; d < 256; d <<= 1) {
		const unsigned long long t = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
		__syncthreads();
		part[threadIdx.x] += t;
		__syncthreads();
	}
	unsigned long long run = threadIdx.x ? part[threadIdx.x - 1] : 0;
#pragma unroll
	for(int k = 0; k < 8; k++) { if(base + k < n1) v[base + k] = run; run += x[k]; }
	if(threadIdx.x == 255) tile_sum[blockIdx.x] = part[255];
}
__global__ __launch_bounds__(1024) void k_scan_tile_sums(unsigned long long* tile_sum, size_t ntiles) {      // in place, exclusive; one workgroup
	__shared__ unsigned long long part[1024];
	const size_t per = (ntiles + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < ntiles ? lo + per : ntiles;
	unsigned long long sum = 0;
	for(size_t k = lo; k < hi; k++) sum += tile_sum[k];
	part[threadIdx.x] = sum;
	__syncthreads();
	for(unsigned d = 1; d < 1024; d <<= 1) {
		const unsigned long long t = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
		__syncthreads();
		part[threadIdx.x] += t;
		__syncthreads();
	}
	unsigned long long run = threadIdx.x ? part[threadIdx.x - 1] : 0;
	for(size_t k = lo; k < hi; k++) { const unsigned long long x = tile_sum[k]; tile_sum[k] = run; run += x; }
}
__global__ __launch_bounds__(256) void k_scan_add(unsigned long long* v, size_t n1, const unsigned long long* tile_sum) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(i < n1) v[i] += tile_sum[i / H2G_SCAN_TILE];
}
__global__ __launch_bounds__(256) void k_compact_gather(const h2g_alnres* src, uint32_t slots, const uint32_t* cnt, uint32_t cnt_stride, const unsigned long long* offs, size_t n,
                                                        uint8_t* dst, const PairOut* pout, const h2g_alnres* ovf, int mate)
{
	const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	if(i >= n) return;
	uint32_t c = cnt[i * cnt_stride];
	const h2g_alnres* a = compact_src(src, slots, i, &c, pout, ovf, mate);
	uint8_t* d = dst + offs[i];
	for(uint32_t k = 0; k < c; k++) {
		const uint32_t ne = a[k].nedits, e = ne > H2G_MAX_EDITS ? 1u : ne, bytes = compact_rec_bytes(ne);
		const unsigned long long* sw = reinterpret_cast<const unsigned long long*>(&a[k]);
		unsigned long long* dw = reinterpret_cast<unsigned long long*>(d);
		for(uint32_t w = 0; w < 5; w++) dw[w] = sw[w];                       // fw .. score: 40 bytes
		const uint32_t* se = reinterpret_cast<const uint32_t*>(a[k].edits);
		uint32_t* de = reinterpret_cast<uint32_t*>(d + 40);
		for(uint32_t w = 0; w < 3 * e; w++) de[w] = se[w];
		if(e & 1u) de[3 * e] = 0;                                            // the alignment pad: never stale bytes
		d += bytes;
	}
}
static_assert(offsetof(h2g_alnres, edits) == 40 && sizeof(h2g_edit) == 12, "compact records are prefixes of h2g_alnres");

// sizes + scan for one mate into tmp slot `slot` (-> device offsets [n + 1], exclusive, in bytes)
static int compact_offsets(h2g_stream* s, const h2g_alnres* d_src, uint32_t slots, const uint32_t* d_cnt, uint32_t cnt_stride, size_t n, int slot, const PairOut* d_pout, int mate, void** d_offs) {
	int rc;
	if((rc = tmp_buf(s, slot, (n + 1) * 8 + ((n + 1) / H2G_SCAN_TILE + 2) * 8, d_offs))) return rc;      // offsets [n + 1], then the scan's tile sums
	hipLaunchKernelGGL(k_compact_sizes, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, s->st, d_src, slots, d_cnt, cnt_stride, n, (unsigned long long*)*d_offs, d_pout, (const h2g_alnres*)s->d_paln_ovf, mate);
	{	// exclusive scan of the n + 1 sizes (the last one is 0: its slot becomes the total); the tile sums live behind the offsets
		const size_t n1 = n + 1, ntiles = (n1 + H2G_SCAN_TILE - 1) / H2G_SCAN_TILE;
		unsigned long long* v = (unsigned long long*)*d_offs;
		unsigned long long* ts = v + n1;
		hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)ntiles), dim3(256), 0, s->st, v, n1, ts);
		hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(1024), 0, s->st, ts, ntiles);
		hipLaunchKernelGGL(k_scan_add, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s->st, v, n1, (const unsigned long long*)ts);
	}
	HIPCHK(hipGetLastError());
	return H2G_OK;
}

extern "C" h2g_status h2g_align_pairs_fetch_compact(h2g_stream* s, h2g_pair_result* res, uint8_t* rec1, size_t cap1, uint64_t* boffs1, uint8_t* rec2, size_t cap2, uint64_t* boffs2,
                                                    size_t first, size_t n)
{
	if(s && s->st2_busy) { for(int k_ = 0; k_ < H2G_MSTREAMS_MAX; k_++) HIPCHK(hipStreamSynchronize(s->mst[k_])); s->st2_busy = false; }
	if(!s || !res || !rec1 || !rec2 || !boffs1 || !boffs2 || first + n > s->n_reads || !s->d_pout) return H2G_ERR_ARG;
	if(n == 0) { boffs1[0] = boffs2[0] = 0; return H2G_OK; }
	HIPCHK(hipSetDevice(s->ix->device));
	static_assert(offsetof(PairOut, nres) == 0 && sizeof(h2g_pair_result) == sizeof(PairOut), "PairOut layout");
	const uint32_t* cnt = reinterpret_cast<const uint32_t*>(s->d_pout + first);
	void *d_o1 = nullptr, *d_o2 = nullptr, *d_b1 = nullptr, *d_b2 = nullptr;
	int rc;
	if((rc = compact_offsets(s, s->d_paln[0] + first * s->pair_slots, s->pair_slots, cnt, sizeof(PairOut) / 4, n, 0, s->d_pout + first, 0, &d_o1)) ||
	   (rc = compact_offsets(s, s->d_paln[1] + first * s->pair_slots, s->pair_slots, cnt + 1, sizeof(PairOut) / 4, n, 2, s->d_pout + first, 1, &d_o2))) return (h2g_status)rc;
	HIPCHK(hipMemcpyAsync(boffs1, d_o1, (n + 1) * 8, hipMemcpyDeviceToHost, s->st));
	HIPCHK(hipMemcpyAsync(boffs2, d_o2, (n + 1) * 8, hipMemcpyDeviceToHost, s->st));
	HIPCHK(hipMemcpyAsync(res, s->d_pout + first, n * sizeof(PairOut), hipMemcpyDeviceToHost, s->st));
	HIPCHK(hipStreamSynchronize(s->st));
	for(size_t i = 0; i < n; i++) if(res[i].pad) { res[i].overflow &= ~4u; res[i].pad = 0; }      // (every record is returned; the block offset is the device's business)
	if(boffs1[n] > cap1 || boffs2[n] > cap2) return H2G_ERR_ARG;              // (boffs[n] = the bytes needed)
	if((rc = tmp_buf(s, 1, boffs1[n] + 8, &d_b1)) || (rc = tmp_buf(s, 3, boffs2[n] + 8, &d_b2))) return (h2g_status)rc;
	const unsigned g = (unsigned)((n + 255) / 256);
	if(boffs1[n]) hipLaunchKernelGGL(k_compact_gather, dim3(g), dim3(256), 0, s->st, s->d_paln[0] + first * s->pair_slots, s->pair_slots, cnt, (uint32_t)(sizeof(PairOut) / 4),
	                                 (const unsigned long long*)d_o1, n, (uint8_t*)d_b1, s->d_pout + first, (const h2g_alnres*)s->d_paln_ovf, 0);
	if(boffs2[n]) hipLaunchKernelGGL(k_compact_gather, dim3(g), dim3(256), 0, s->st, s->d_paln[1] + first * s->pair_slots, s->pair_slots, cnt + 1, (uint32_t)(sizeof(PairOut) / 4),
	                                 (const unsigned long long*)d_o2, n, (uint8_t*)d_b2, s->d_pout + first, (const h2g_alnres*)s->d_paln_ovf, 1);
	HIPCHK(hipGetLastError());
	if(boffs1[n]) HIPCHK(hipMemcpyAsync(rec1, d_b1, boffs1[n], hipMemcpyDeviceToHost, s->st));
	if(boffs2[n]) HIPCHK(hipMemcpyAsync(rec2, d_b2, boffs2[n], hipMemcpyDeviceToHost, s->st));
	HIPCHK(hipStreamSynchronize(s->st));
	return H2G_OK;
}

extern "C" h2g_status h2g_align_fetch_compact(h2g_stream* s, h2g_read_result* res, uint8_t* rec, size_t cap, uint64_t* boffs, size_t first, size_t n) {
	if(s && s->st2_busy) { for(int k_ = 0; k_ < H2G_MSTREAMS_MAX; k_++) HIPCHK(hipStreamSynchronize(s->mst[k_])); s->st2_busy = false; }
	if(!s || !res || !rec || !boffs || first + n > s->n_reads || !s->d_rout) return H2G_ERR_ARG;
	if(n == 0) { boffs[0] = 0; return H2G_OK; }
	HIPCHK(hipSetDevice(s->ix->device));
	static_assert(offsetof(ReadOut, nselect) == 4, "ReadOut layout");
	const uint32_t* cnt = reinterpret_cast<const uint32_t*>(s->d_rout + first) + 1;
	void *d_o = nullptr, *d_b = nullptr;
	int rc;
	if((rc = compact_offsets(s, s->d_aln + first * s->aln_slots, s->aln_slots, cnt, sizeof(ReadOut) / 4, n, 0, nullptr, 0, &d_o))) return (h2g_status)rc;
	HIPCHK(hipMemcpyAsync(boffs, d_o, (n + 1) * 8, hipMemcpyDeviceToHost, s->st));
	const h2g_status hr = h2g_align_fetch(s, res, nullptr, first, n);           // (the headers; syncs the stream)
	if(hr != H2G_OK) return hr;
	if(boffs[n] > cap) return H2G_ERR_ARG;
	if(boffs[n] == 0) return H2G_OK;
	if((rc = tmp_buf(s, 1, boffs[n] + 8, &d_b))) return (h2g_status)rc;
	hipLaunchKernelGGL(k_compact_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->st, s->d_aln + first * s->aln_slots, s->aln_slots, cnt, (uint32_t)(sizeof(ReadOut) / 4),
	                   (const unsigned long long*)d_o, n, (uint8_t*)d_b, (const PairOut*)nullptr, (const h2g_alnres*)nullptr, 0);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(rec, d_b, boffs[n], hipMemcpyDeviceToHost, s->st));
	HIPCHK(hipStreamSynchronize(s->st));
	return H2G_OK;
}

// page-locked host memory for the buffers a caller hands to the set_* / fetch_* entry points: copies from / to it run at the link's rate and asynchronously
extern "C" void* h2g_host_alloc(size_t bytes) { void* p = nullptr; if(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; } return p; }
extern "C" void h2g_host_free(void* p) { if(p) (void)hipHostFree(p); }

// The edit lists of the records with more than H2G_MAX_EDITS edits of the resident batch (include/h2g.h): the used prefix of the stream's long-edit
// area, offsets as the records carry them in edits[0].pos.  *n = edits the prefix spans (0: no such record); H2G_ERR_ARG when cap is smaller.
extern "C" h2g_status h2g_align_fetch_long_edits(h2g_stream* s, h2g_edit* out, size_t cap, size_t* n) {
	if(!s || !n) return H2G_ERR_ARG;
	*n = 0;
	if(!s->ran_align || !s->d_ledits || !s->d_ledits_cur) return H2G_OK;
	HIPCHK(hipSetDevice(s->ix->device));
	if(s->st2_busy) { for(int k_ = 0; k_ < H2G_MSTREAMS_MAX; k_++) HIPCHK(hipStreamSynchronize(s->mst[k_])); s->st2_busy = false; }
	HIPCHK(hipStreamSynchronize(s->st));
	uint32_t cur[H2G_MSTREAMS_MAX];
	HIPCHK(hipMemcpy(cur, s->d_ledits_cur, sizeof cur, hipMemcpyDeviceToHost));
	size_t hi = 0;
	// (only the parts the runs since the last upload wrote to: a part left over from an earlier batch keeps its cursor until its machine stream is used again — ADVICE r5)
	for(unsigned m = 0; m < s->ledits_parts; m++) if(!((s->ledits_touched >> m) & 1u)) cur[m] = (uint32_t)(m * s->ledits_cap);
	for(unsigned m = 0; m < s->ledits_parts; m++) {
		size_t end = cur[m]; const size_t lo = m * s->ledits_cap, top = (m + 1) * s->ledits_cap;
		if(end > top) end = top;                       // (a full part: the reads that found no room are flagged)
		if(end > lo) hi = end;
	}
	*n = hi;
	if(hi == 0) return H2G_OK;
	if(!out || cap < hi) return H2G_ERR_ARG;
	for(unsigned m = 0; m < s->ledits_parts; m++) {
		size_t end = cur[m]; const size_t lo = m * s->ledits_cap, top = (m + 1) * s->ledits_cap;
		if(end > top) end = top;
		if(end > lo) HIPCHK(hipMemcpy(out + lo, s->d_ledits + lo, (end - lo) * sizeof(h2g_edit), hipMemcpyDeviceToHost));
	}
	return H2G_OK;
}

// development hook: measurement / debugging knobs of go_run by name.  Everything in flight is waited for first, so a change never meets a
// queued run.  "fast" 0/1, "blocks_per_cu", "pair_slots", "no_second_pass", "mach_div", "mach_min", "dbg_read" (-1 = off), "tail", "align_mate",
// "mstreams" (machine passes kept in flight, 1..H2G_MSTREAMS_MAX), "mach_total" (workgroups of all of them together), "fast_reserve" (CUs the fast pass leaves
// free: -1 = as many as the machine passes in flight may hold)
extern "C" __attribute__((visibility("default"))) int h2g_stream_tune(h2g_stream* s, const char* key, long v) {
	if(!s || !key) return H2G_ERR_ARG;
	HIPCHK(sync_all(s));
	for(int k_ = 0; k_ < H2G_MSTREAMS_MAX; k_++) HIPCHK(hipStreamSynchronize(s->mst[k_]));
	const std::string k(key);
	if(k == "fast") s->tune.fast = (int)v; else if(k == "blocks_per_cu") s->tune.blocks_per_cu = (int)v; else if(k == "pair_slots") s->tune.pair_slots = (int)v;
	else if(k == "no_second_pass") s->tune.no_second_pass = (int)v; else if(k == "mach_div") s->tune.mach_div = (unsigned)v; else if(k == "mach_min") s->tune.mach_min = (unsigned)v;
	else if(k == "dbg_read") s->tune.dbg_read = v; else if(k == "tail") { s->tune.tail = (int)v; s->tune.tail_auto = 0; } else if(k == "align_mate") s->tune.align_mate = (int)v;
	else if(k == "mach_total") { s->tune.mach_total_auto = v <= 0; s->tune.mach_total = v <= 0 ? H2G_MACH_TOTAL : (unsigned)v; }      // (0 = the default policy)
	else if(k == "fast_reserve") s->tune.fast_reserve = (int)v;
	else if(k == "mate_handover") s->tune.mate_handover = (int)v;
	else if(k == "orphan") s->tune.orphan = (int)v; else if(k == "drain_grid") s->tune.drain_grid = (int)(v < 1 ? 1 : v > 128 ? 128 : v);
	else if(k == "mstreams") { s->mstreams = (unsigned)(v < 1 ? 1 : v > H2G_MSTREAMS_MAX ? H2G_MSTREAMS_MAX : v); s->gen = 0; }   // (nothing is in flight: every buffer set is free)
	else return H2G_ERR_ARG;
	return H2G_OK;
}

// development hook (h2g_stream_tune "dbg_read" / env H2G_GO_DBG_READ=<read id>): the primitive requests of that read in the last go() launch, 8 words each
extern "C" __attribute__((visibility("default"))) int h2g_go_debug_trace(h2g_stream* s, uint32_t* out, uint32_t cap_words) {
	if(s && s->st2_busy) { for(int k_ = 0; k_ < H2G_MSTREAMS_MAX; k_++) HIPCHK(hipStreamSynchronize(s->mst[k_])); s->st2_busy = false; }   // (results of the machine pass on the second stream)
	if(!s || !out || !s->dbg_buf) return H2G_ERR_ARG;
	HIPCHK(sync_all(s));
	HIPCHK(hipMemcpy(out, s->dbg_buf, (size_t)cap_words * 4, hipMemcpyDeviceToHost));
	return H2G_OK;
}

// development hook (builds with -DH2G_GO_PROF): the wave-level time split of the last go() launch, 48 slots (h2g_go_kernels.h)
extern "C" __attribute__((visibility("default"))) int h2g_go_prof(h2g_stream* s, unsigned long long* out48) {
	if(!s || !out48) return H2G_ERR_ARG;
	HIPCHK(sync_all(s));
	HIPCHK(hipMemcpy(out48, s->cnt_cur + 16, 80 * sizeof(unsigned long long), hipMemcpyDeviceToHost));   // [0..47] split, [64..79] control by source ring
	return H2G_OK;
}

// development hook (builds with -DH2G_GO_PROF): the wave-level time split of the last fast pass, 48 slots + 24 bail reasons (h2g_k_go_fast.hip)
extern "C" __attribute__((visibility("default"))) int h2g_go_fast_prof(h2g_stream* s, unsigned long long* out72 /* [136] */) {
	if(!s || !out72) return H2G_ERR_ARG;
	HIPCHK(sync_all(s));
	HIPCHK(hipMemcpy(out72, s->cnt_cur + 128, 48 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(out72 + 48, s->cnt_cur + 96, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(out72 + 72, s->cnt_cur + 176, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));   // control time / trips by site
	return H2G_OK;
}
#ifdef H2G_GO_PROF
extern "C" __attribute__((visibility("default"))) int h2g_go_fast_prof_bins(h2g_stream* s, unsigned long long* out256) {
	if(!s || !out256) return H2G_ERR_ARG;
	HIPCHK(sync_all(s));
	HIPCHK(hipMemcpy(out256, s->cnt_cur + 512, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	return H2G_OK;
}
#endif

extern "C" h2g_status h2g_get_counters(h2g_stream* s, h2g_counters* c) {
	if(!s || !c) return H2G_ERR_ARG;
	HIPCHK(sync_all(s));
	unsigned long long v[16];
	unsigned long long* const cb = s->ran_align ? s->cnt_cur : s->d_counters;
	HIPCHK(hipMemcpy(v, cb, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(v + 8, cb + (s->ran_align ? 256 : 64), 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	// [0..7] the main pass, [8..15] <- slots 256..263: the second pass over its overflowed reads (go_run)
	s->last.n_rank = v[0] + v[8]; s->last.n_side = v[1] + v[9]; s->last.n_sa_steps = v[2] + v[10]; s->last.n_ext = v[3];
	s->last.n_aligned = v[4] + v[12];
	uint32_t nsecond = 0;
	if(s->d_ovf_list[s->ovf_cur] && s->ran_align) HIPCHK(hipMemcpy(&nsecond, s->d_ovf_list[s->ovf_cur] + s->max_reads, 4, hipMemcpyDeviceToHost));
	s->last.n_second_pass = nsecond;
	s->last.n_overflow = nsecond ? v[13] : v[5];   // reads still flagged after the last pass that saw them
	s->last.n_queries = s->n_reads * 2;
	float t = 0;
	if(s->ran_seed) {
		if(hipEventElapsedTime(&t, s->ev[2], s->ev[3]) == hipSuccess) s->last.ms_search = t;
		if(hipEventElapsedTime(&t, s->ev[3], s->ev[4]) == hipSuccess) s->last.ms_resolve_extend = t;
	}
	if(s->ran_align && hipEventElapsedTime(&t, s->ev[7], s->ev[6]) == hipSuccess) s->last.ms_align_kernel = t;
	if(s->ran_align && hipEventElapsedTime(&t, s->ev[5], s->ev[8]) == hipSuccess) s->last.ms_align = t;   // every pass
	s->last.n_fast = 0; s->last.n_fast_bail = 0; s->last.ms_fast_kernel = 0; s->last.pad_ = 0; s->last.n_fast_side = 0; s->last.n_fast_sa_steps = 0;
	s->last.ms_drain_kernel = 0; s->last.pad2_ = 0; s->last.n_drain_side = 0; s->last.n_drain_sa_steps = 0; s->last.n_adopted = 0;
	if(s->ran_align && s->ran_fast) {
		unsigned long long f[2], fc[4];
		HIPCHK(hipMemcpy(f, s->cnt_cur + 6, sizeof f, hipMemcpyDeviceToHost));
		HIPCHK(hipMemcpy(fc, s->cnt_cur + 120, sizeof fc, hipMemcpyDeviceToHost));
		s->last_bails = (uint32_t)f[1];
		s->last.n_fast = f[0]; s->last.n_fast_bail = f[1];
		s->last.n_fast_side = fc[1]; s->last.n_fast_sa_steps = fc[2];
		s->last.n_rank += fc[0]; s->last.n_side += fc[1]; s->last.n_sa_steps += fc[2]; s->last.n_aligned += fc[3];
		if(hipEventElapsedTime(&t, s->ev[5], s->ev[10]) == hipSuccess) s->last.ms_fast_kernel = t;
		if(s->ran_drain) {
			unsigned long long dc[4]; uint32_t nad = 0;
			HIPCHK(hipMemcpy(dc, s->cnt_cur + 240, sizeof dc, hipMemcpyDeviceToHost));
			HIPCHK(hipMemcpy(&nad, s->orph_cur, 4, hipMemcpyDeviceToHost));
			s->last.n_drain_side = dc[1]; s->last.n_drain_sa_steps = dc[2]; s->last.n_adopted = nad;
			s->last.n_rank += dc[0]; s->last.n_side += dc[1]; s->last.n_sa_steps += dc[2]; s->last.n_aligned += dc[3];
			if(hipEventElapsedTime(&t, s->ev_dr[0], s->ev_dr[1]) == hipSuccess) s->last.ms_drain_kernel = t;
		}
	}
	(void)hipGetLastError();
	*c = s->last;
	return H2G_OK;
}
