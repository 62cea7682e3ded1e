/*
 * h2o — CPU ORACLE for the HISAT2 seed-and-extend hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference algorithm (HISAT2 2.2.3), each function citing
 * the reference file:line it follows.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product (hisat2_amd/libh2g.so)
 * never links, calls or falls back to it.
 *
 * Parity pinning: every function here is checked against vectors emitted by the *real*
 * reference classes (oracle/ref_probe.cpp -> oracle/_ref/ref_probe) committed under
 * tests/golden/, and the end-to-end path against SAM written by oracle/_ref/hisat2-align-s.
 */
#ifndef H2O_H_
#define H2O_H_
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define H2O_MAX 0xffffffffu

/* GFMParams (gfm.h:115-199), for index_t = u32 (wsz 4) or local_index_t = u16 (wsz 2) */
typedef struct {
	uint32_t len, gbwtLen, numNodes;
	int32_t  lineRate, offRate, ftabChars;
	uint32_t eftabLen;
	int      linear;
	uint32_t offMask, ftabLen, offsLen, sideSz, sideGbwtSz, sideGbwtLen, numSides, gbwtTotLen;
	int      wsz;
} h2o_params;

typedef struct {
	h2o_params p;
	uint32_t  nPat, nFrag;
	uint32_t *plen, *rstarts;     /* rstarts[3*nFrag] = {joinedOff, textId, textOff} */
	uint8_t  *gfm;                /* sides, raw on-disk bytes */
	uint32_t  nZ, *zOffs;
	uint32_t  fchr[5];
	uint32_t *ftab, *eftab, *offs;
	/* local indexes only (hgfm.h:1130-1140) */
	uint32_t  tidx, localOffset, joinedOffset;
} h2o_gfm;

/* BitPairReference (reference.h:58, reference.cpp:30-380) */
typedef struct {
	uint32_t  nrecs, nrefs;
	uint32_t *rec_off, *rec_len;  /* RefRecord.off / .len */
	uint8_t  *rec_first;
	uint32_t *refRecOffs;         /* [nrefs+1] first record of each ref */
	uint32_t *refOffs;            /* [nrefs+1] unambiguous chars preceding each ref */
	uint32_t *refLens;            /* [nrefs] total (ambiguous+unambiguous) length */
	uint8_t  *buf;                /* 2-bit packed unambiguous bases */
	uint64_t  bufSz;
} h2o_ref;

/* ALT (alt.h:41-120) as GFM::GFM loads them (gfm.h:728-905): a reversed copy of every deletion is appended
 * (pos = last deleted base, reversed = 1) and the list is sorted by ALT::operator< (alt.h:88-102) */
enum { H2O_ALT_SNP_SGL = 1, H2O_ALT_SNP_INS = 2, H2O_ALT_SNP_DEL = 3, H2O_ALT_SNP_ALT = 4, H2O_ALT_SPLICESITE = 5, H2O_ALT_EXON = 6 };
typedef struct { uint32_t pos, type, len; uint64_t seq; } h2o_alt;   /* seq & 0xff = `reversed` for deletions */

typedef struct {
	h2o_gfm   g;                  /* global index (.1/.2) */
	h2o_ref   r;                  /* reference (.3/.4) */
	uint32_t  nlocal;             /* local indexes (.5/.6) */
	h2o_gfm  *local;
	uint32_t *local_first;        /* [nPat+1] first local index of each text */
	uint32_t  minK;               /* hi_aligner.h:3979-3984 */
	char    **names;
	uint32_t  nalts;              /* ALTDB (.7.ht2), graph indexes only */
	h2o_alt  *alts;
} h2o_index;

/* Scoring subset (scoring.h:100-546, defaults hisat2.cpp:425-441) */
typedef struct {
	int mmpMax, mmpMin, nPen, rdGapConst, rdGapLinear, rfGapConst, rfGapLinear, scMax, scMin;
	int matchBonus;
} h2o_scoring;

/* Edit (edit.h) — only what the linear path uses */
enum { H2O_EDIT_READ_GAP = 1, H2O_EDIT_REF_GAP = 2, H2O_EDIT_MM = 3 }; /* edit.h:37-39 */
typedef struct { uint32_t pos; uint8_t chr, qchr, type, pad; uint32_t snp; /* Edit::snpID, H2O_MAX = none */ } h2o_edit;


enum { H2O_CANDIDATE_HIT = 1, H2O_PSEUDOGENE_HIT = 2, H2O_ANCHOR_HIT = 3 }; /* hi_aligner.h:98 */

/* Result of one partialSearch call: the appended BWTHit + the ReadBWTHit counters */
typedef struct {
	uint32_t top, bot, node_top, node_bot, bwoff, len, hit_type;
	uint32_t cur, done, numPartialSearch, numUniqueSearch;
	uint32_t pseudogeneStop, anchorStop;
	uint32_t nrank;               /* bwops_ (hi_aligner.h:6467,6474): rank calls issued */
	uint32_t nside;               /* unique sides touched (SURVEY §8(d) algorithmic bytes) */
} h2o_bwthit;

typedef struct {
	uint32_t tidx;                /* H2O_MAX when the hit straddles a boundary */
	uint32_t toff, joinedOff;
} h2o_coord;

#define H2O_MAX_EDITS 64
typedef struct {
	uint32_t fw, rdoff, len, trim5, trim3, tidx, toff, joinedOff;
	int64_t  score;
	uint32_t nedits;
	h2o_edit edits[H2O_MAX_EDITS];
} h2o_ghit;

int  h2o_index_load(const char* base, h2o_index** out);
void h2o_index_free(h2o_index*);
void h2o_scoring_default(h2o_scoring*);

uint32_t h2o_rank(const h2o_gfm*, uint32_t row, int c);              /* countBt2Side gfm.h:2958 */
int      h2o_rowL(const h2o_gfm*, uint32_t row);                     /* rowL gfm.h:3615 */
int      h2o_ftab_lohi(const h2o_gfm*, const uint8_t* seq, uint32_t off, uint32_t* top, uint32_t* bot); /* gfm.h:2670 */
uint32_t h2o_get_offset(const h2o_gfm*, uint32_t row, uint32_t* steps); /* gfm.h:5682 / group_walk.h */
int      h2o_joined_to_text(const h2o_gfm*, uint32_t qlen, uint32_t off, uint32_t* tidx, uint32_t* toff,
                            uint32_t* tlen, int rejectStraddle, int* straddled); /* gfm.h:5527 */
void     h2o_get_stretch(const h2o_ref*, uint32_t tidx, int64_t toff, uint32_t count, uint8_t* dest); /* reference.cpp:486 */

/* partialSearch hi_aligner.h:6361.  seq = read as 0..4 codes in the searched orientation. */
void h2o_partial_search(const h2o_index*, const uint8_t* seq, uint32_t len, uint32_t cur,
                        int pseudogeneStop, int anchorStop, uint32_t khits, h2o_bwthit* out);
/* same for linear AND graph indexes; iedges[2*cap] <- BWTHit::_node_iedge_count (hi_aligner.h:199) */
void h2o_partial_search_graph(const h2o_index*, const uint8_t* seq, uint32_t len, uint32_t cur,
                              int pseudogeneStop, int anchorStop, uint32_t khits, uint32_t kseeds,
                              h2o_bwthit* out, uint32_t* iedges, uint32_t cap, uint32_t* niedges);
/* graph index primitives (gfm.h): rank over the M bit-vector, select over F, in-edge counts, graph LF */
uint32_t h2o_rank_M(const h2o_gfm*, uint32_t row);                              /* gfm.h:4100 */
uint32_t h2o_select_F(const h2o_gfm*, uint32_t row, uint32_t count);            /* gfm.h:4113 */
uint32_t h2o_in_edge_count(const h2o_gfm*, uint32_t top, uint32_t bot, uint32_t* iedges, uint32_t cap); /* gfm.h:4172 */
int h2o_map_glf(const h2o_gfm*, uint32_t top, uint32_t bot, int c, uint32_t k, uint32_t* otop, uint32_t* obot,
                uint32_t* ontop, uint32_t* onbot, uint32_t* iedges, uint32_t cap, uint32_t* niedges); /* gfm.h:3759 */
int h2o_map_glf1(const h2o_gfm*, uint32_t row, int c, uint32_t* otop, uint32_t* obot,
                 uint32_t* ontop, uint32_t* onbot);                             /* gfm.h:3957 */
/* getGenomeCoords hi_aligner.h:5774 (linear index): coords for rows [top, top+nelt) */
int  h2o_genome_coords(const h2o_index*, uint32_t top, uint32_t bot, uint32_t maxelt, uint32_t rdlen,
                       int rejectStraddle, h2o_coord* coords, uint32_t* ncoords, int* straddled, uint32_t* nsteps);
/* getGenomeCoords on a GRAPH index: the group walk over NODES (group_walk.h:464-1545); iedges = BWTHit::_node_iedge_count */
int  h2o_genome_coords_graph(const h2o_index*, uint32_t top, uint32_t bot, uint32_t node_top, uint32_t node_bot,
                             const uint32_t* iedges, uint32_t niedges, uint32_t maxelt, uint32_t rdlen, int rejectStraddle,
                             h2o_coord* coords, uint32_t* ncoords, int* straddled, uint32_t* nsteps);
/* GenomeHit::extend hi_aligner.h:2031 (+alignWithALTs :683, _recur :2763 without ALTs,
 * calculateScore :3711).  seq/qual in the hit's orientation. */
int  h2o_extend(const h2o_index*, const h2o_scoring*, const uint8_t* seq, const char* qual, uint32_t rdlen,
                h2o_ghit* hit, uint32_t* leftext, uint32_t* rightext, uint32_t mm);
/* static GenomeHit::adjustWithALT hi_aligner.h:2239 (as getAnchorHits calls it, :5175): appends the hits an anchor coordinate
 * yields once indel ALTs are accounted for (findOffDiffs :2545) and known variants are written as edits */
int  h2o_adjust_with_alt(const h2o_index*, const uint8_t* seq, int fw, uint32_t rdoff, uint32_t len, uint32_t tidx, uint32_t toff,
                         uint32_t joinedOff, h2o_ghit* hits, uint32_t* nhits, uint32_t cap);
int64_t h2o_calculate_score(const h2o_scoring*, const char* qual, h2o_ghit* hit);

/* SwAligner as called from hybridSearch (spliced_aligner.h:209-262): frame + end-to-end fill (8-bit cells; 16-bit when minsc < -254) + gather + the first
 * nextAlignment.  seq/qual in the aligned orientation; refoff = hit.refoff - hit.rdoff (or 0); *rnd = RandomSource::last */
typedef struct {
	int64_t  refl, refr, refl_pretrim, refr_pretrim, corel, corer;   /* DPRect dp_framer.h */
	int32_t  found_align;         /* SwAligner::align() */
	int64_t  best;                /* bestCell (lrmax - 0xff) */
	int32_t  found;               /* nextAlignment() */
	int64_t  score, off;          /* AlnRes score, refcoord().off() */
	uint32_t nedits, gaps, overflow;
	h2o_edit edits[H2O_MAX_EDITS];
} h2o_sw_result;
int h2o_sw_align(const h2o_index*, const h2o_scoring*, const uint8_t* seq, const char* qual, uint32_t rdlen,
                 uint32_t tidx, uint32_t refoff, int64_t minsc, int nceil, int gapbar, uint32_t* rnd, h2o_sw_result* out);
/* the same DP over columns refl..refr of a plain reference string (codes 0..4; N outside it), core diagonals corel..corer: the setting of the
 * reference's own known-answer cases (aligner_sw.cpp:1470-2727) */
int h2o_sw_align_window(const h2o_scoring*, const uint8_t* seq, const char* qual, uint32_t rdlen, const uint8_t* ref, uint32_t reflen,
                        int64_t refl, int64_t refr, int64_t corel, int64_t corer, int64_t minsc, int nceil, int gapbar, uint32_t* rnd,
                        h2o_sw_result* out, int* ns_out);

/* whole-batch CPU baseline of the stage timed by bench.py: both strands' partialSearch +
 * coordinate resolution + 0-mismatch extension, returns a checksum */
uint64_t h2o_seed_extend_batch(const h2o_index*, const uint8_t* seqs, const uint32_t* offs, uint32_t nreads,
                               int pseudogeneStop, uint32_t khits, uint64_t* counters);

#ifdef __cplusplus
}
#endif
#endif
