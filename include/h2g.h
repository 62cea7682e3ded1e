/*
 * h2g.h — C ABI of the MI355X-native HISAT2 seed-and-extend hot path (libh2g.so).
 *
 * The reference (HISAT2 2.2.3) has no FFI for this path: the hot path is header-template
 * code called from the per-thread worker loop (hisat2.cpp:3276-3644 -> HI_Aligner::go,
 * hi_aligner.h:4048).  This header is the boundary a maintainer binds instead; each entry
 * point names the reference function whose semantics it reproduces bit-exactly.  Conventions
 * follow the reference's only C ABI, hisat2lib/ht2.h:30-150: opaque handles, `int` status
 * (0 = OK, <0 = error), option structs initialised by a function, no exceptions across the
 * boundary, caller-owned buffers with explicit capacities.
 *
 * Plain pointers and sizes only; no torch / C++ types.  All kernels are hand-written HIP for
 * gfx950; there is NO CPU fallback — every entry point returns H2G_ERR_DEVICE without a GPU.
 */
#ifndef H2G_H_
#define H2G_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define H2G_EXPORT __attribute__((visibility("default")))   /* cf. hisat2lib/ht2_handle.h:23-27 */

typedef struct h2g_index  h2g_index;    /* immutable device-resident .ht2 index; shareable across streams      */
typedef struct h2g_stream h2g_stream;   /* per-host-thread batch context: device scratch + one hipStream_t     */
typedef int h2g_status;

enum {
	H2G_OK = 0,
	H2G_ERR_IO = -1,          /* index file missing / short read                                        */
	H2G_ERR_FORMAT = -2,      /* not a little-endian .ht2 (32-bit offsets) index                         */
	H2G_ERR_DEVICE = -3,      /* no usable HIP device / HIP runtime error (see h2g_last_error)           */
	H2G_ERR_ARG = -4,         /* bad argument / capacity exceeded                                        */
	H2G_ERR_UNSUPPORTED = -5, /* feature of the reference not built yet (e.g. graph index in a kernel)   */
	H2G_ERR_NOMEM = -6
};

#define H2G_MAX 0xffffffffu     /* INDEX_MAX of the 32-bit (-s) binaries, btypes.h:24-46 */

/* ---- index ------------------------------------------------------------------------------------------ */
typedef struct {
	int32_t device;            /* HIP device ordinal */
	int32_t load_local;        /* also upload the local (.5/.6) indexes */
} h2g_load_opts;

/* GFMParams gfm.h:115-199 + hi_aligner.h:3979 (_minK) */
typedef struct {
	uint32_t len, gbwtLen, numNodes;
	int32_t  lineRate, offRate, ftabChars;
	uint32_t eftabLen, linear;
	uint32_t sideSz, sideGbwtSz, sideGbwtLen, numSides, offsLen, ftabLen;
	uint32_t nPat, nFrag, nZ, minK, nLocal, nRefRecs;
	uint64_t device_bytes;     /* bytes resident in HBM */
} h2g_index_info;

H2G_EXPORT void       h2g_load_opts_init(h2g_load_opts*);
/* Parses <base>.1-.6.ht2 exactly like GFM::readIntoMemory (gfm.h:5823-6453), HGFM::readIntoMemory
 * (hgfm.h:2453-2651) and BitPairReference (reference.cpp:30-380) and uploads them to HBM. */
H2G_EXPORT h2g_status h2g_index_load(const char* ht2_base, const h2g_load_opts*, h2g_index** out);
H2G_EXPORT h2g_status h2g_index_get_info(const h2g_index*, h2g_index_info* out);
/* Roofline helper: a device-only index whose sides hold `num_sides` random linear sides with
 * consistent Occ checkpoints (GRCh38 scale = 15.3 M sides ~ 0.98 GB); only rank queries are valid. */
H2G_EXPORT h2g_status h2g_index_synth_sides(uint64_t num_sides, uint64_t seed, int device, h2g_index** out);
H2G_EXPORT void       h2g_index_free(h2g_index*);
/* Splice sites for spliced alignment (SpliceSiteDB, splice_site.h:470): what --known-splicesite-infile / --novel-splicesite-infile
 * hand the reference (SpliceSiteDB::read splice_site.cpp:727: text name, left = last base of the upstream exon, right = first
 * base of the downstream exon, both 0-based, strand).  go() joins reads through them (spliced_aligner.h:409-676, 685-811,
 * 1365-1496) and the SAM formatter leaves their introns out of TLEN (aligner_result.h:1669-1689).  Replaces the previous set;
 * n == 0 empties the database.  readid / fromfile: a site with fromfile == 0 is one found by read `readid` and is visible only to
 * reads readid + window and later (the reference's -p window, hisat2.cpp:3687). */
typedef struct { uint32_t tidx, left, right, readid; uint8_t dir /* 2 = '+', 3 = '-' (SPL_FW / SPL_RC) */, fromfile, known, editdist /* out of h2g_sam_take_novel_sites: mismatches + gaps of the line that crossed the site (<= 255); ignored on input */; } h2g_splice_site;
H2G_EXPORT h2g_status h2g_index_set_splice_sites(h2g_index*, const h2g_splice_site* sites, size_t n, uint32_t window);
/* the sites met since (new ones, or known ones whose smallest read id went down) join the database: O(sites + n log n), no re-allocation;
 * the call waits for the device first (a caller with shards of a wave still in flight pays that wait; the command line merges at wave
 * boundaries only).  Without a h2g_index_set_splice_sites call before it the database starts from the index's own splice-site ALTs and
 * the window is 0 (SpliceSiteDB::addSpliceSite, splice_site.cpp:243; the temporary-splice-site waves of the command line).
 * Queued runs: h2g_align(_pairs)_run calls queued without a fetch between them share the result rows; a run whose options differ
 * from the previous one's first waits for everything still in flight (go_run). */
H2G_EXPORT h2g_status h2g_index_add_splice_sites(h2g_index*, const h2g_splice_site* delta, size_t n);

H2G_EXPORT const char* h2g_last_error(void);

/* ---- stream / reads ---------------------------------------------------------------------------------- */
H2G_EXPORT h2g_status h2g_stream_create(h2g_index*, size_t max_reads, size_t max_bases, h2g_stream** out);
H2G_EXPORT void       h2g_stream_free(h2g_stream*);
H2G_EXPORT void*      h2g_stream_hip(h2g_stream*);   /* the hipStream_t every kernel of this context runs on */
H2G_EXPORT h2g_status h2g_stream_sync(h2g_stream*);
/* Resident batches: a stream holds up to 16 read sets, each with result rows of its own.  Batch 0 is selected when the stream is created.  After a selection the
 * h2g_set_* calls fill that batch, h2g_align_*run works on it and the fetches read its rows; runs queued over other batches stay in flight (their machine passes next
 * to this batch's fast pass), which is a streaming caller's steady state — the reference's worker threads likewise hold different reads at any moment
 * (hisat2.cpp:3276-3644).  The call itself never waits. */
H2G_EXPORT h2g_status h2g_stream_select_batch(h2g_stream*, unsigned batch);

/* Reads as the worker loop holds them after parsing (read.h:325, Read::patFw): one byte per base,
 * codes A,C,G,T,N = 0..4; read i = codes[offs[i] .. offs[i+1]).  quals = ASCII (phred+33) or NULL for
 * FASTA input (all 'I', pat.cpp).  Copied to HBM; the reverse complement (patRc) is derived on device. */
H2G_EXPORT h2g_status h2g_set_reads(h2g_stream*, const uint8_t* codes, const uint32_t* offs, const char* quals,
                                    size_t n_reads);

/* ---- primitives (batched; semantics == the reference function named) ---------------------------------- */

/* GFM::mapLF(SideLocus(row), c) -> countBt2Side (gfm.h:3712, :2958): the Occ-rank micro-kernel.
 * variant: 0 = one lane per side, 1 = 4 lanes per side (16 B each + DPP reduce), 2 = 8 lanes per side.
 * On a graph index (128 B sides): 0 = one lane per side (8 x dwordx4), 1 = 8 lanes per side.
 * rows/cs/out are HOST arrays unless device_ptrs != 0.  *kernel_ms = HIP-event time of the kernel. */
H2G_EXPORT h2g_status h2g_rank_bench(h2g_stream*, const uint32_t* rows, const uint8_t* cs, size_t n, uint32_t* out,
                                     int variant, int device_ptrs, int repeats, float* kernel_ms);
/* generates the SURVEY §8(d) query set on device: row ~ U[0,gbwtLen) from splitmix64(seed), c = hash&3 */
H2G_EXPORT h2g_status h2g_rank_bench_synth(h2g_stream*, size_t n, uint64_t seed, int variant, int repeats,
                                           float* kernel_ms, uint64_t* checksum);
/* out[j] = result j * stride of this stream's last h2g_rank_bench_synth run: the sampled comparison with GFM::mapLF on the CPU */
H2G_EXPORT h2g_status h2g_rank_bench_synth_sample(h2g_stream*, size_t stride, size_t nsample, uint32_t* out);

enum { H2G_FM_PARTIAL = 0, H2G_FM_GLOBAL = 1, H2G_FM_LOCAL = 2 };
enum { H2G_CANDIDATE_HIT = 1, H2G_PSEUDOGENE_HIT = 2, H2G_ANCHOR_HIT = 3 };   /* hi_aligner.h:96-100 */

typedef struct {
	uint32_t read;             /* index into the batch set by h2g_set_reads */
	uint32_t offset;           /* ReadBWTHit::_cur: bases already consumed from the 3' end */
	uint8_t  fw;               /* 1: patFw, 0: patRc */
	uint8_t  mode;             /* H2G_FM_* */
	uint8_t  pseudogeneStop;   /* hi_aligner.h:4669 */
	uint8_t  anchorStop;       /* hi_aligner.h:4670 */
} h2g_fm_query;

/* the BWTHit appended by partialSearch (hi_aligner.h:6361-6600) + the ReadBWTHit counters it updates */
typedef struct {
	uint32_t top, bot, node_top, node_bot, bwoff, len, hit_type;
	uint32_t cur, done, numPartialSearch, numUniqueSearch, pseudogeneStop, anchorStop;
	uint32_t nrank;            /* rank calls (HI_Aligner::bwops_) */
	uint32_t nside;            /* unique sides visited = algorithmic bytes / sideSz */
} h2g_fm_hit;

H2G_EXPORT h2g_status h2g_fm_search(h2g_stream*, const h2g_fm_query* q, size_t n, uint32_t khits, h2g_fm_hit* out);

/* ---- the searches of hybridSearch_recur: globalGFMSearch (hi_aligner.h:6606) and localGFMSearch (:6751) -----------------
 * Backward search of the read bases [.., rdoff] of the searched strand (rdoff counted from its 5' end, inclusive) in the global
 * index (lidx == H2G_MAX) or in local index `lidx` (HGFM::getLocalGFM hgfm.h:1713: lidx = h2g_local_index_of(tidx, toff)).
 * Local queries are bucketed by lidx; a bucket with at least `stage_min` queries is served by one workgroup that first copies
 * the local index's sides (16-33 KB) and ftab (8 KB) into LDS — every rank of those queries is an LDS read; the other queries
 * (and graph-index locals) read their sides from HBM.  stage_min == 0 never stages.  Linear and graph indexes. */
typedef struct { uint32_t read, rdoff, lidx, maxHitLen; uint8_t fw, uniqueStop, pad[2]; } h2g_ext_search_query;
typedef struct { uint32_t nelt, hitlen, top, bot, uniqueStop, nrank, nside, staged; } h2g_ext_search_hit;
typedef struct { uint64_t n_local, n_staged, n_buckets, n_buckets_staged, lds_bytes_staged; float ms_staged, ms_hbm; } h2g_ext_search_stats;
H2G_EXPORT h2g_status h2g_ext_search(h2g_stream*, const h2g_ext_search_query* q, size_t n, uint32_t stage_min, h2g_ext_search_hit* out,
                                     h2g_ext_search_stats* stats /* nullable */);
H2G_EXPORT uint32_t   h2g_local_index_of(const h2g_index*, uint32_t tidx, uint32_t toff);   /* H2G_MAX: none */

/* ---- graph (GFM) index primitives: 128 B sides with F/M bit vectors (gfm.h:160-176) ------------------- */
/* BWTHit::_node_iedge_count (hi_aligner.h:199): nodes of a range with more than one incoming edge */
#ifndef H2G_IEDGE_CAP       /* (a per-translation-unit capacity like H2G_GHIT_EDITS: the graph fast pass keeps lists of two entries and sets 4) */
#define H2G_IEDGE_CAP 24
#endif
typedef struct {
	uint32_t n;                        /* true count; only the first H2G_IEDGE_CAP entries are stored */
	uint32_t e[H2G_IEDGE_CAP][2];      /* {node index relative to node_top, extra in-edges} */
} h2g_iedges;
/* single == 0: GFM::mapGLF(tloc(top), bloc(bot), c, &node_range, &node_iedges, k)   (gfm.h:3759-3837)
 * single != 0: GFM::mapGLF1(top, loc(top), c, &node_range)                          (gfm.h:3957-4021) */
typedef struct { uint32_t top, bot; uint8_t c, single, pad[2]; } h2g_glf_query;
typedef struct { uint32_t ok, top, bot, node_top, node_bot; } h2g_glf_result;   /* ok == 0: empty (0,0) */
H2G_EXPORT h2g_status h2g_graph_lf(h2g_stream*, const h2g_glf_query* q, size_t n, uint32_t k, h2g_glf_result* res,
                                   h2g_iedges* iedges /* [n] or NULL */);
/* partialSearch on a graph index (h2g_fm_search forwards here with kseeds = max(5, 2 khits), hisat2.cpp:3177);
 * iedges[n] (or NULL) receives the in-edge list the BWTHit is initialised with (hi_aligner.h:6570-6578) */
H2G_EXPORT h2g_status h2g_fm_search_graph(h2g_stream*, const h2g_fm_query* q, size_t n, uint32_t khits, uint32_t kseeds,
                                          h2g_fm_hit* out, h2g_iedges* iedges);
/* Roofline helper, graph flavour of h2g_index_synth_sides: random 128 B graph sides with consistent Occ
 * checkpoints (F/M bits random, headers zero); only rank queries (h2g_rank_bench*) are valid on it. */
H2G_EXPORT h2g_status h2g_index_synth_graph_sides(uint64_t num_sides, uint64_t seed, int device, h2g_index** out);

/* HI_Aligner::getGenomeCoords (hi_aligner.h:5774-5855): GroupWalk2S::advanceElement + GFM::joinedToTextOff */
typedef struct { uint32_t top, bot, maxelt, len; uint32_t rejectStraddle; } h2g_sa_query;
typedef struct { uint32_t tidx, toff, joinedOff; } h2g_coord;          /* tidx == H2G_MAX: straddles */
typedef struct { uint32_t ok, ncoords, straddled, nsteps; } h2g_sa_result;
H2G_EXPORT h2g_status h2g_sa_resolve(h2g_stream*, const h2g_sa_query* q, size_t n, uint32_t cap_per_query,
                                     h2g_coord* coords /* [n*cap] */, h2g_sa_result* res /* [n] */);
/* getGenomeCoords on a graph index: the node-based group walk (GroupWalk2S / GWState, group_walk.h:464-1545).  Query i
 * walks the nodes [node_top, node_bot) of rows [top, bot) with in-edge list iedges[i] (as h2g_fm_search_graph returns
 * them; NULL = none).  Results as h2g_sa_resolve; res[i].nsteps == H2G_MAX flags a capacity overflow (ok = 0). */
typedef struct { uint32_t top, bot, node_top, node_bot, maxelt, len, rejectStraddle; } h2g_gsa_query;
H2G_EXPORT h2g_status h2g_sa_resolve_graph(h2g_stream*, const h2g_gsa_query* q, const h2g_iedges* iedges, size_t n,
                                           uint32_t cap_per_query, h2g_coord* coords /* [n*cap] */, h2g_sa_result* res /* [n] */);

/* GenomeHit::extend (hi_aligner.h:2031-2232) incl. alignWithALTs (:683) and calculateScore (:3711) */
#define H2G_MAX_EDITS 32
enum { H2G_EDIT_READ_GAP = 1, H2G_EDIT_REF_GAP = 2, H2G_EDIT_MM = 3, H2G_EDIT_SPL = 5 };  /* edit.h:36-42 */
/* An H2G_EDIT_SPL edit (an intron, CIGAR N) keeps Edit::splLen / splDir / knownSpl in the three bytes chr, qchr, pad:
 * splLen = chr | qchr << 8 | (pad & 15) << 16 (introns up to 2^20), splDir = (pad >> 4) & 7 (splice_site.h:37-43: 1 unknown, 2 +, 3 -,
 * 4 semi +, 5 semi -), knownSpl = pad >> 7; `snp` holds the float bits of SpliceSiteDB::probscore(donor_seq, acceptor_seq). */
typedef struct { uint32_t pos; uint8_t chr, qchr, type, pad; uint32_t snp; /* Edit::snpID: index into the ALT list, H2G_MAX = none */ } h2g_edit;   /* Edit, edit.h */
/* h2g_ghit is also the WORKING hit of the go() kernels.  The go() units with the large workspace (the second pass, h2g_go_big.h) are compiled with
 * longer edit lists in it (H2G_GHIT_EDITS, set before this header is read): the reference's lists are unbounded (hi_aligner.h:421), and a deletion of
 * n bases is n edits.  Every entry point of this header sees the default, H2G_MAX_EDITS. */
#ifndef H2G_GHIT_EDITS
#define H2G_GHIT_EDITS H2G_MAX_EDITS
#endif
typedef struct {
	uint32_t read;
	uint32_t fw, rdoff, len, trim5, trim3, tidx, toff, joinedOff;
	uint32_t splicescore;      /* GenomeHit::_splicescore (mean intron length of the short-anchored splices), truncated */
	int64_t  score;
	uint32_t nedits;
	uint32_t overflow;         /* edit list exceeded H2G_MAX_EDITS: caller must take its own path */
	h2g_edit edits[H2G_GHIT_EDITS];
} h2g_ghit;
typedef struct { uint32_t mm, max_leftext, max_rightext; } h2g_ext_args;
typedef struct { uint32_t extended, leftext, rightext; } h2g_ext_result;
H2G_EXPORT h2g_status h2g_extend(h2g_stream*, h2g_ghit* hits /* in/out */, const h2g_ext_args* args, size_t n,
                                 h2g_ext_result* res);

/* static GenomeHit::adjustWithALT (hi_aligner.h:2239-2390) as getAnchorHits calls it on a graph index (:5175): an anchor
 * (read, strand, rdoff, len) placed at (tidx, toff, joinedOff) by the SA walk -> the GenomeHits it yields once offsets are
 * corrected for indel ALTs (findOffDiffs :2545) and known variants are written as edits.  hits[i * cap ...], nhits[i]. */
typedef struct { uint32_t read, fw, rdoff, len, tidx, toff, joinedOff; } h2g_adjust_query;
H2G_EXPORT h2g_status h2g_adjust_with_alt(h2g_stream*, const h2g_adjust_query* q, size_t n, uint32_t cap, h2g_ghit* hits /* [n*cap] */,
                                          uint32_t* nhits /* [n] */);

/* ---- Smith-Waterman extension (opt-in in the reference: --bowtie2-dp / --sensitive) ---------------------------- */
/* One problem = the SwAligner call site of hybridSearch (spliced_aligner.h:209-262) for one seed hit of one read:
 * DynProgFramer::frameSeedExtensionRect (dp_framer.cpp:81) around refoff = hit.refoff - hit.rdoff, SwAligner::initRef
 * (aligner_sw.cpp:137), the end-to-end fill — 8-bit cells, alignNucleotidesEnd2EndSseU8 (aligner_swsse_ee_u8.cpp:791), or for
 * minsc < -254 16-bit cells, alignNucleotidesEnd2EndSseI16 (aligner_swsse_ee_i16.cpp:793): SwAligner::align's rule, aligner_sw.cpp:496 —,
 * gatherCells (:1202), and the first SwAligner::nextAlignment (aligner_sw.cpp:709) with its backtrace (:1309) and PRNG
 * reseeding.  Edits are in the coordinates of the aligned strand (fw: patFw, !fw: patRc), ascending. */
typedef struct {
	uint32_t read;             /* index into the batch set by h2g_set_reads */
	uint32_t fw;               /* 1: align patFw, 0: patRc */
	uint32_t tidx, refoff;     /* reference id; ref offset implied by the seed hit assuming no gaps */
	int32_t  minsc;            /* _minsc[rdi] (scoreMin.f(len), hisat2.cpp:3470) */
	uint32_t rnd;              /* RandomSource::last on entry */
} h2g_sw_query;
typedef struct {
	int32_t  found_align;      /* SwAligner::align(): at least one candidate cell */
	int32_t  found;            /* SwAligner::nextAlignment() */
	int32_t  best;             /* bestCell: best last-row score (lrmax - 0xff) */
	int32_t  score;            /* AlnRes::score().score() */
	int64_t  off;              /* AlnRes::refcoord().off() */
	uint32_t nedits, gaps;
	uint32_t overflow;         /* edit / branch-stack capacity exceeded: caller must take its own path */
	uint32_t rnd;              /* RandomSource::last on exit */
	int64_t  refl, refr;       /* DPRect::refl / refr (inclusive) */
	h2g_edit edits[H2G_MAX_EDITS];
} h2g_sw_result;
/* *kernel_ms (nullable) = HIP-event time of the SW kernel; repeats > 1 re-runs it for timing */
H2G_EXPORT h2g_status h2g_sw_align(h2g_stream*, const h2g_sw_query* q, size_t n, h2g_sw_result* out, int repeats,
                                   float* kernel_ms);

/* ---- fused seed-and-extend stage over the resident read batch ------------------------------------------ */
/* For every read and both strands: partialSearch from offset 0 (nextBWT hi_aligner.h:4644-4760) ->
 * getAnchorHits coordinate resolution (:5007, ranges up to H2G_SEED_CAP rows) -> 0-mismatch extend
 * (hybridSearch spliced_aligner.h:139-163).  Results stay in HBM until h2g_seed_extend_fetch. */
#define H2G_SEED_CAP 5
typedef struct {
	h2g_fm_hit hit;
	uint32_t   ncoords, straddled, nsteps, pad;
	struct { uint32_t tidx, toff, joinedOff, rdoff, len; int32_t score; } ext[H2G_SEED_CAP];
} h2g_seed_result;                         /* one per (read, strand): index 2*read + (fw ? 0 : 1) */

typedef struct {
	uint32_t pseudogeneStop, anchorStop, khits;
	uint32_t search_variant;   /* 0 = lane-per-query search kernel, 1 = cooperative 8-lane kernel */
} h2g_seed_params;

H2G_EXPORT void       h2g_seed_params_init(h2g_seed_params*, const h2g_index*, int no_spliced_alignment);
H2G_EXPORT h2g_status h2g_seed_extend_run(h2g_stream*, const h2g_seed_params*);     /* async on the stream */
H2G_EXPORT h2g_status h2g_seed_extend_fetch(h2g_stream*, h2g_seed_result* out, size_t first_read, size_t n_reads);

/* ---- the coarse entry: HI_Aligner::go for every read of the resident batch ---------------------------------- */
/* Semantics == one iteration of the worker loop body (hisat2.cpp:3380-3640) for an unpaired read that passed the
 * filters: rnd.init(genRandSeed(read)) (pat.h:55), splicedAligner.go(...) (hi_aligner.h:4048), and the selection
 * half of AlnSinkWrap::finishRead (aln_sink.h:1939 -> selectByScore :2680).  Built so far: linear (HFM) and SNP-graph
 * (GFM + ALT database) indexes, --no-spliced-alignment, default scoring, --bowtie2-dp 0/1/2. */
#define H2G_ALN_CAP 10             /* alignments returned per read (>= -k: 5 on linear, 10 on graph indexes) */
typedef struct {                   /* == the arguments reportHit (hi_aligner.h:6064-6166) passes to AlnRes::init */
	uint32_t fw, tidx, toff, len, trim5, trim3, nedits, splicescore;   /* splicescore: AlnScore::splicescore_ (aligner_result.h:322) */
	int64_t  score;                /* AS:i */
	h2g_edit edits[H2G_MAX_EDITS]; /* as stored in the AlnRes (aligner_result.cpp:110-118): positions along the original read
	                                * 5'->3', relative to its first aligned (non-soft-clipped) base */
} h2g_alnres;
typedef struct {
	uint32_t nres;                 /* alignments reported to the sink (rs1u_) */
	uint32_t nselect;              /* alignments to print, best first; [0] is the primary */
	uint32_t overflow;             /* !=0: a fixed-capacity list overflowed -> caller runs this read through its own go() */
	uint32_t nrank, nsteps, depth; /* work counters: rank calls, SA-walk steps, deepest recursion frame */
	int32_t  best, secbest;        /* AlnSetSumm over all nres alignments (aligner_result.cpp:1209): best / second-best AS:i, */
	uint32_t best_h2, secbest_h2;  /* and the low 32 bits of their AlnScore::hisat2_score_ (aligner_result.h:322: repeat, transcript,
	                                * splice-score and trimmed-base fields — the AlnScore tie-break); best == INT32_MIN = none.  MAPQ, ZS:i */
} h2g_read_result;
typedef struct {
	uint32_t khits, kseeds;        /* -k, --max-seeds */
	uint32_t no_spliced_alignment; /* 1: --no-spliced-alignment; 0: introns are placed by combineWith (needs no_temp_splicesite, linear index) */
	uint32_t secondary;
	uint32_t bowtie2_dp;           /* --bowtie2-dp: 0 off (default), 1 SwAligner when no alignment reached minsc, 2 always
	                                * (spliced_aligner.h:209).  Inside go() the DP is run by the read's own lane over
	                                * ~75 KB of HBM scratch per lane; the batched LDS kernel is h2g_sw_align. */
	/* scoring scheme (Scoring scoring.h:29-87; option -> SeedAlignmentPolicy::parseString aligner_seed_policy.cpp:294-620) */
	int32_t  mm_max, mm_min;       /* --mp MX,MN      6,2  (quality-aware mismatch penalty, COST_MODEL_QUAL) */
	int32_t  n_pen;                /* --np            1 */
	int32_t  rdg_const, rdg_linear;/* --rdg           5,3 */
	int32_t  rfg_const, rfg_linear;/* --rfg           5,3 */
	int32_t  sc_max, sc_min;       /* --sp MX,MN      2,1  (soft-clip penalty); --no-softclip = INT32_MAX,INT32_MAX */
	uint32_t score_min_type;       /* --score-min <type>,<const>,<coeff>: 1 = C, 2 = L, 3 = S (sqrt), 4 = G (log)   (simple_func.h:30-33) */
	double   score_min_const, score_min_coeff;   /* default L,0,-0.2 (hisat2.cpp:440) */
	uint32_t no_temp_splicesite;   /* --no-temp-splicesite: novel splice sites are not shared between reads.  Spliced alignment
	                                * (no_spliced_alignment == 0) is built for this setting on linear indexes: every read is independent */
	/* splice scoring (Scoring::canSpl / noncanSpl scoring.h:473-487, TranscriptomePolicy tp.h; hisat2.cpp:493-499, :1631-1688) */
	uint32_t min_intronlen, max_intronlen;           /* --min-intronlen 20, --max-intronlen 500000 */
	int32_t  pen_cansplice, pen_noncansplice;        /* --pen-cansplice 0, --pen-noncansplice 12 */
	uint32_t pen_canintronlen_type, pen_noncanintronlen_type;   /* --pen-canintronlen / --pen-noncanintronlen <type>,<const>,<coeff>: */
	uint32_t first_read_id;                                     /* type as score_min_type; default G,-8,1.  first_read_id: Read::rdid of
	                                                             * read 0 of the batch — the splice-site window compares read ids */
	double   pen_canintronlen_const, pen_canintronlen_coeff, pen_noncanintronlen_const, pen_noncanintronlen_coeff;
	/* TranscriptomePolicy (hisat2.cpp:4076-4084): anchor minima 7 / 14, with --dta (transcript assemblers) 15 / 20 and
	 * --pen-noncanintronlen G,-8,2; xs_only (--dta-cufflinks): spliced alignments of unknown strand are not reported (hi_aligner.h:6101) */
	uint32_t min_anchor_len, min_anchor_len_noncan, xs_only;
	/* --haplotype (graph indexes): an ALT is only tried when a haplotype of the index carries it together with the ALTs already taken
	 * (GraphPolicy::useHaplotype gp.h:71, alignWithALTs_recur hi_aligner.h:2898-2996, :3251-3331) */
	uint32_t use_haplotype;
	/* --max-altstried (GraphPolicy::maxAltsTried, default 16, at least 8: hisat2.cpp:521, :1745): ALTs one extension may walk through
	 * (hi_aligner.h:2794) and, / 4, the offset combinations adjustWithALT tries (:2313, :2423) */
	uint32_t max_alts_tried;
	/* -X / --maxins (PairedEndPolicy::maxfrag, default 1000 hisat2.cpp:345): the longest fragment of a concordant pair (pe.cpp:38; checked
	 * under --no-spliced-alignment, hi_aligner.h:6018) and half the window alignMate searches the other mate in (:5688) */
	uint32_t max_frag_len;
	/* -I / --minins (PairedEndPolicy::minfrag, default 0); --fr / --rf / --ff as pe_orientation 0 / 1 / 2 (gMate1fw, gMate2fw hisat2.cpp:1166-1168:
	 * the strand pair a concordant pair has, hi_aligner.h:5605, :6003, pe.cpp:59-83); --nofw / --norc: the strand of the READ (of mate 1's
	 * fragment strand for pairs, hisat2.cpp:3449-3452) that is not searched (hi_aligner.h:4875) */
	uint32_t min_frag_len, pe_orientation, nofw, norc;
} h2g_align_params;
/* number of visible HIP devices (0 without a GPU: the library has no CPU path) */
H2G_EXPORT int        h2g_device_count(void);
H2G_EXPORT void       h2g_align_params_init(h2g_align_params*, const h2g_index*);
/* The reference applies its presets after ALL options were read (hisat2.cpp:1882-1909) and lets the index type decide the
 * default -k (:3903-3906): khits = saw_k ? k_arg : 10; --sensitive: bowtie2_dp 0 -> 1, khits < 10 -> 10 (counts as saw_k),
 * --score-min L,0,-0.5; --very-sensitive: bowtie2_dp 2, khits < 30 -> 30, L,0,-1; without saw_k khits = 5 (linear) / 10 (graph);
 * max_seeds_arg 0 -> max(5, 2 khits) (:3174).  Call it last, after every other field of *p was set from the options. */
H2G_EXPORT void       h2g_align_params_presets(h2g_align_params* p, const h2g_index* ix, int saw_k, uint32_t k_arg, uint32_t max_seeds_arg,
                                               int sensitive, int very_sensitive);
/* read names (needed by genRandSeed): name i = bytes[offs[i] .. offs[i+1]) */
H2G_EXPORT h2g_status h2g_set_read_names(h2g_stream*, const char* bytes, const uint32_t* offs, size_t n_reads);
/* GenomeHit::combineWith (hi_aligner.h:1420-2025) as a primitive of its own: a[i] (the left hit) absorbs b[i] — concatenation, the mismatch rescan of the joint,
 * an insertion or deletion, or (spliced alignment) an intron placed by the donor / acceptor scan — over the resident reads; ok[i] = the function's return value.  `p` carries
 * scoring and splice policy (NULL: h2g_align_params_init's defaults for this index).  Inside go() the same device function runs as OP_COMBINE. */
H2G_EXPORT h2g_status h2g_combine_with(h2g_stream*, const h2g_align_params* p, h2g_ghit* a /* in/out */, const h2g_ghit* b, const int64_t* minsc, size_t n, uint32_t* ok);
H2G_EXPORT h2g_status h2g_align_run(h2g_stream*, const h2g_align_params*);           /* async on the stream */
H2G_EXPORT h2g_status h2g_align_fetch(h2g_stream*, h2g_read_result* res /* [n] */, h2g_alnres* aln /* [n*H2G_ALN_CAP] or NULL */,
                                      size_t first_read, size_t n_reads);

/* Dense variant for callers that move the results over PCIe: only the nselect printed alignments of every read, back to back
 * (gathered on the device); read i's records are aln[aln_offs[i] .. aln_offs[i+1]).  H2G_ERR_ARG if aln_cap is too small
 * (aln_offs[n] then holds the count needed). */
H2G_EXPORT h2g_status h2g_align_fetch_dense(h2g_stream*, h2g_read_result* res /* [n] */, h2g_alnres* aln, size_t aln_cap,
                                            uint64_t* aln_offs /* [n+1] */, size_t first_read, size_t n_reads);

/* ---- paired-end: HI_Aligner::go with both mates (initReads hi_aligner.h:4019; pairReads :5948; alignMate :5579) -- */
/* Mate 2 of every read of the batch (same count as h2g_set_reads); names2 feed genRandSeed of mate 2. */
H2G_EXPORT h2g_status h2g_set_mates(h2g_stream*, const uint8_t* codes2, const uint32_t* offs2, const char* quals2,
                                    const char* name_bytes2, const uint32_t* name_offs2, size_t n_reads);
#define H2G_PAIR_RES_CAP 16       /* unpaired alignments returned per mate */
#define H2G_PAIR_CAP 32           /* concordant pairs returned per read pair */
typedef struct {
	uint32_t nres[2];              /* sink.report(mate) events per mate, in report order (rs1u_/rs2u_) */
	uint32_t npairs;               /* sink.report(r1, r2) events, in report order (rs1_/rs2_) */
	uint32_t overflow, nrank, nsteps, depth, nside;
	uint32_t rnd_state;            /* RandomSource::last after go(): the sink's selectByScore continues from it */
	uint32_t pad;                  /* device-side: != 0 when the pair's records live in the stream's growable area (a mate with more reports than its
	                                * fixed rows); h2g_align_pairs_fetch_dense returns all of them, h2g_align_pairs_fetch sets overflow bit 4 for such a pair */
	uint8_t  pair_i[H2G_PAIR_CAP], pair_j[H2G_PAIR_CAP];   /* indexes into the two per-mate lists */
} h2g_pair_result;
/* The concordant / discordant / unpaired decision, -k selection, MAPQ and SAM stay in the caller's AlnSinkWrap
 * (aln_sink.h:1939): it replays these events through msinkwrap.report() and calls finishRead(). */
H2G_EXPORT h2g_status h2g_align_pairs_run(h2g_stream*, const h2g_align_params*);
H2G_EXPORT h2g_status h2g_align_pairs_fetch(h2g_stream*, h2g_pair_result* res /* [n] */, h2g_alnres* aln1 /* [n*H2G_PAIR_RES_CAP] */,
                                            h2g_alnres* aln2 /* [n*H2G_PAIR_RES_CAP] */, size_t first_read, size_t n_reads);

/* Dense variant: the nres[m] report events of each mate back to back (see h2g_align_fetch_dense) */
H2G_EXPORT h2g_status h2g_align_pairs_fetch_dense(h2g_stream*, h2g_pair_result* res /* [n] */, h2g_alnres* aln1, size_t cap1, uint64_t* aln_offs1 /* [n+1] */,
                                                  h2g_alnres* aln2, size_t cap2, uint64_t* aln_offs2 /* [n+1] */, size_t first_read, size_t n_reads);

/* Compact variants — what a caller on the other side of PCIe wants: sized, scanned and gathered on the device in one go.  A record travels as its 40 bytes
 * of fields + 12 bytes per edit it holds (a record beyond H2G_MAX_EDITS: its one marker entry), rounded up to 8: read i's records lie back to back in
 * rec[boffs[i] .. boffs[i + 1]) (byte offsets).  A compact record is a PREFIX of an h2g_alnres: read it through a `const h2g_alnres*` up to edits[nedits), never
 * copy the struct; the next record starts H2G_COMPACT_BYTES(nedits) further.  H2G_ERR_ARG when a capacity is too small (boffs[n] then holds the bytes needed).
 * include/h2g_sam.h formats this layout directly (h2g_sam_format_*_compact). */
#define H2G_COMPACT_BYTES(nedits) ((40u + 12u * ((nedits) > H2G_MAX_EDITS ? 1u : (uint32_t)(nedits)) + 7u) & ~7u)
H2G_EXPORT h2g_status h2g_align_fetch_compact(h2g_stream*, h2g_read_result* res /* [n] */, uint8_t* rec, size_t cap, uint64_t* boffs /* [n+1] */, size_t first_read, size_t n_reads);
H2G_EXPORT h2g_status h2g_align_pairs_fetch_compact(h2g_stream*, h2g_pair_result* res /* [n] */, uint8_t* rec1, size_t cap1, uint64_t* boffs1 /* [n+1] */,
                                                    uint8_t* rec2, size_t cap2, uint64_t* boffs2 /* [n+1] */, size_t first_read, size_t n_reads);
/* Page-locked host memory for the buffers handed to h2g_set_* / h2g_*_fetch_*: copies from / to it run at the link's rate (pageable memory is staged by the
 * runtime at a fraction of it).  NULL when the allocation fails. */
H2G_EXPORT void*      h2g_host_alloc(size_t bytes);
H2G_EXPORT void       h2g_host_free(void*);

/* Records beyond H2G_MAX_EDITS edits.  The reference's edit lists are unbounded (hi_aligner.h:421, reportHit :6129-6166) and a deletion of n bases
 * is n edits (edit.h).  A record whose list does not fit its H2G_MAX_EDITS inline entries says so with nedits > H2G_MAX_EDITS: its edits live in the
 * stream's long-edit area, at offset edits[0].pos (edits[0].snp == 0x4c4f4e47 marks it; the other inline entries are unspecified).  This call returns
 * the used prefix of that area for the resident batch: *n = edits it spans (0 when the batch has no such record); H2G_ERR_ARG with *n set when cap is
 * smaller.  include/h2g_sam.h: h2g_sam_set_long_edits hands it to the formatter.  Capacity of the working lists: 192 edits per alignment (the units with
 * the large workspace, which the flagged reads of the default units are re-run by); beyond that a read keeps overflow bit 1. */
H2G_EXPORT h2g_status h2g_align_fetch_long_edits(h2g_stream*, h2g_edit* out, size_t cap, size_t* n);

/* ---- counters (roofline numerators) ------------------------------------------------- */
typedef struct {
	uint64_t n_rank, n_side, n_sa_steps, n_ext, n_ref_bytes, n_queries, n_aligned, n_overflow;
	float    ms_search, ms_resolve_extend, ms_rank, ms_align;   /* HIP-event durations of the last launches */
	float    ms_align_kernel;                                   /* the main go() pass alone (ms_align includes the second pass) */
	uint64_t n_second_pass;                                     /* reads whose default workspace overflowed and that were re-run with the
	                                                             * large one; n_overflow = reads still flagged after that */
	uint64_t n_fast, n_fast_bail;                               /* reads / pairs completed by the fast pass; handed on to the general machine */
	float    ms_fast_kernel, pad_;                              /* the fast pass alone (ms_align covers every pass) */
	uint64_t n_fast_side, n_fast_sa_steps;                      /* sides / SA-walk steps of the fast pass alone (n_side, n_sa_steps cover every pass) */
	float    ms_drain_kernel, pad2_;                            /* the drain launch of the fast pass: the reads its workgroups still held when the batch was exhausted ... */
	uint64_t n_drain_side, n_drain_sa_steps, n_adopted;         /* ... its sides / SA-walk steps (NOT in n_fast_side / n_fast_sa_steps) and how many reads it took up */
} h2g_counters;
H2G_EXPORT h2g_status h2g_get_counters(h2g_stream*, h2g_counters*);

#ifdef __cplusplus
}
#endif
#endif /* H2G_H_ */
