/* h2g_sam.h — SAM emission for the alignments the device path produces.
 *
 * The go() kernels return what HI_Aligner::reportHit hands to the sink (h2g_alnres == the arguments of AlnRes::init,
 * hi_aligner.h:6143-6167).  These entry points are the host half that the reference runs afterwards for every read:
 *   AlnSinkWrap::finishRead          aln_sink.h:1939   (which lists are printed, primary / secondary, NH:i)
 *   AlnSetSumm::init                 aligner_result.cpp:1167   (best / second-best scores)
 *   BowtieMapq2::mapq                unique.h:187      (MAPQ, `--mapq-v 2`, the default hisat2.cpp:480)
 *   AlnSinkSam::appendMate           aln_sink.h:3024   (the eleven mandatory fields)
 *   StackedAln::init/leftAlign/buildCigar/buildMdz   aligner_result.cpp:660-1000 (CIGAR, MD:Z)
 *   SamConfig::printAlignedOptFlags / printEmptyOptFlags   sam.h:525 / :1033 (AS ZS XN XM XO XG NM MD YS YT YF NH Zs)
 *   AlnRes::setMateParams / setFragmentLength        aligner_result.h:1594-1697 (TLEN)
 * with the reference's default print options (hisat2.cpp:371-409).  Pure host code: no device, no index traffic; it
 * needs only the reference names (.1.ht2) and, for Zs:Z, the ALT names (.7/.8.ht2).  Every line it writes is byte-identical
 * to the line `hisat2 --no-spliced-alignment` prints for the same read (tests/test_sam_lines.py).
 */
#ifndef H2G_SAM_H_
#define H2G_SAM_H_
#include "h2g.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct h2g_sam h2g_sam;

/* reference names + lengths, ALT types/lengths/names of the index `base` (no device needed) */
H2G_EXPORT h2g_status h2g_sam_open(const char* index_base, h2g_sam** out);
H2G_EXPORT void       h2g_sam_close(h2g_sam*);
/* host threads used by the format calls (contiguous read ranges, output concatenated in read order); default 1 */
H2G_EXPORT void       h2g_sam_set_threads(h2g_sam*, int threads);
/* the alignment summary the reference prints on stderr at the end of a run (AlnSink::printAlSumm aln_sink.h:1637), over all
 * reads formatted so far by this handle.  Returns bytes needed; writes at most cap. */
H2G_EXPORT size_t     h2g_sam_summary(const h2g_sam*, char* out, size_t cap);
/* --no-unal: lines of reads / mates that failed to align are not printed (aln_sink.h:3040) */
H2G_EXPORT void       h2g_sam_set_no_unal(h2g_sam*, int on);
/* --remove-chrname (1) / --add-chrname (2): the reference names lose / gain a leading "chr" (hisat2.cpp:3962-3976); call right after open */
H2G_EXPORT void       h2g_sam_set_chrname_mode(h2g_sam*, int mode);
/* --new-summary: h2g_sam_summary returns the "HISAT2 summary stats:" text instead (aln_sink.h:1659-1679) */
H2G_EXPORT void       h2g_sam_set_new_summary(h2g_sam*, int on);
/* --rg-id <text> (id) / --rg <text> (field; "ID:x" acts like --rg-id x): the @RG header line, printed when an id is set, and RG:Z:<id> on
 * every record (hisat2.cpp:1389-1407, sam.h:456, :780, :1102).  Either argument may be NULL. */
H2G_EXPORT void       h2g_sam_add_read_group(h2g_sam*, const char* id, const char* field);
/* --no-sq: the header without @SQ lines; --omit-sec-seq: SEQ and QUAL of secondary lines are '*' (aln_sink.h:3190) */
H2G_EXPORT void       h2g_sam_set_header_options(h2g_sam*, int no_sq, int omit_sec_seq);
/* --no-discordant (discordant = 0): a pair with one alignment per mate that is not concordant is not converted into a discordant pair
 * (ReportingState::nextRead aln_sink.cpp:38); --no-mixed (mixed = 0): mates of a pair without a paired alignment are reported unaligned
 * (ReportingState::getReport aln_sink.cpp:280).  Both default to 1 (hisat2.cpp:353-354). */
H2G_EXPORT void       h2g_sam_set_report_policy(h2g_sam*, int discordant, int mixed);
/* --secondary: the sink's -k selection for pairs keeps lower-scoring alignments (aln_sink.h:2733-2745) */
H2G_EXPORT void       h2g_sam_set_secondary(h2g_sam*, int on);
/* reads a --known-splicesite-infile / --novel-splicesite-infile (SpliceSiteDB::read splice_site.cpp:727): returns the number of
 * records of texts the index holds ((size_t)-1: cannot open), the first `cap` of them in out */
H2G_EXPORT size_t     h2g_sam_read_splice_site_file(const h2g_sam*, const char* path, int known, h2g_splice_site* out, size_t cap);
/* the splice sites given to h2g_index_set_splice_sites: TLEN of a concordant pair leaves the longest database intron lying between
 * its mates out (AlnRes::setFragmentLength aligner_result.h:1669-1689) */
H2G_EXPORT void       h2g_sam_set_splice_sites(h2g_sam*, const h2g_splice_site* sites, size_t n, uint32_t window);
H2G_EXPORT void       h2g_sam_add_splice_sites(h2g_sam*, const h2g_splice_site* delta, size_t n);   /* as h2g_index_add_splice_sites */
/* --no-templatelen-adjustment (on = 0): setMateParams without the database, TLEN keeps every intron (aln_sink.h:2070-2076) */
H2G_EXPORT void       h2g_sam_set_templatelen_adjustment(h2g_sam*, int on);
/* --rna-strandness: 0 unknown (XS:A from the splice directions), 1 F, 2 R, 3 FR, 4 RF (XS:A on every aligned line, sam.h:940-966) */
H2G_EXPORT void       h2g_sam_set_rna_strandness(h2g_sam*, int code);
/* Temporary splice sites (the reference's default mode; SpliceSiteDB::addSpliceSite splice_site.cpp:190, called for every line
 * written, aln_sink.h:1570): with collection on, the format calls record the junctions of the alignments they print, tagged with
 * the read's id = first_read_id + its index in the call.  h2g_sam_take_novel_sites hands them over in read order (returns the
 * count; nothing is consumed unless cap holds them all).  The caller merges them into its database — per site the smallest
 * read id — and passes the result to h2g_index_set_splice_sites / h2g_sam_set_splice_sites with the window of its wave scheme. */
H2G_EXPORT void       h2g_sam_collect_novel_sites(h2g_sam*, int on);
H2G_EXPORT void       h2g_sam_set_first_read_id(h2g_sam*, uint64_t id);
H2G_EXPORT size_t     h2g_sam_take_novel_sites(h2g_sam*, h2g_splice_site* out, size_t cap);
/* --novel-splicesite-outfile (SpliceSiteDB::print splice_site.cpp:565, written at the end of a run hisat2.cpp:4189): the text of the
 * file over every site this handle has seen — the index's and the files' (h2g_sam_set_splice_sites entries with fromfile) and those
 * of the lines formatted while collection was on, with the reference's read-count cut-off and its merging of near-identical sites.
 * Returns bytes needed; writes at most cap. */
H2G_EXPORT size_t     h2g_sam_novel_splice_sites_text(const h2g_sam*, char* out, size_t cap);
/* --score-min as given to the aligner (h2g_align_params.score_min_*): MAPQ is relative to it (unique.h:214-222) */
H2G_EXPORT void       h2g_sam_set_score_min(h2g_sam*, uint32_t type, double constant, double coeff);

/* Records with more than H2G_MAX_EDITS edits (nedits > H2G_MAX_EDITS: a long deletion is one edit per base, edit.h) keep their edit list in the
 * long-edit area of their batch (h2g_align_fetch_long_edits, include/h2g.h).  Hand the area of the batch about to be formatted to the handle (not
 * copied: it must stay valid through the format call; n = 0 clears it).  A dense format call over such a record without its area is H2G_ERR_ARG. */
H2G_EXPORT void       h2g_sam_set_long_edits(h2g_sam*, const h2g_edit* area, size_t n);

/* "@HD / @SQ / @PG" header as the reference prints it (sam.h printHeader: VN:1.0 SO:unsorted, one @SQ per reference,
 * @PG ID:hisat2 PN:hisat2 VN:<version> CL:"<cmdline>").  Returns bytes needed; writes at most cap. */
H2G_EXPORT size_t     h2g_sam_header(const h2g_sam*, const char* cmdline, char* out, size_t cap);

/* Unpaired reads: one h2g_read_result + H2G_ALN_CAP h2g_alnres slots per read, exactly as h2g_align_fetch returns them.
 * codes = base codes 0..4 of the forward strand, offs[n+1]; quals = ASCII qualities with the same offsets or NULL (FASTA:
 * 'I').  *used = bytes needed for the whole batch; H2G_ERR_ARG (nothing truncated mid-line is ever left) if > cap. */
H2G_EXPORT h2g_status h2g_sam_format_unpaired(const h2g_sam*, const uint8_t* codes, const uint32_t* offs, const char* quals,
                                              const char* name_bytes, const uint32_t* name_offs, size_t n_reads,
                                              const h2g_read_result* res, const h2g_alnres* aln /* [n*H2G_ALN_CAP] */,
                                              char* out, size_t cap, size_t* used);

/* Same, over the dense layout of h2g_align_fetch_dense: read i's records start at aln[aln_offs[i]] */
H2G_EXPORT h2g_status h2g_sam_format_unpaired_dense(const h2g_sam*, const uint8_t* codes, const uint32_t* offs, const char* quals,
                                                    const char* name_bytes, const uint32_t* name_offs, size_t n_reads,
                                                    const h2g_read_result* res, const h2g_alnres* aln, const uint64_t* aln_offs /* [n+1] */,
                                                    char* out, size_t cap, size_t* used);

/* Read pairs: the report events of h2g_align_pairs_fetch (per-mate lists + concordant pair list + PRNG state).  Runs the
 * sink's decision (concordant / discordant / unpaired, -k selection continuing the per-pair PRNG) and prints both mates. */
H2G_EXPORT h2g_status h2g_sam_format_paired(const h2g_sam*, const uint8_t* codes1, const uint32_t* offs1, const char* quals1,
                                            const char* name_bytes1, const uint32_t* name_offs1, const uint8_t* codes2,
                                            const uint32_t* offs2, const char* quals2, const char* name_bytes2,
                                            const uint32_t* name_offs2, size_t n_pairs, const h2g_pair_result* res,
                                            const h2g_alnres* aln1 /* [n*H2G_PAIR_RES_CAP] */, const h2g_alnres* aln2,
                                            uint32_t khits, char* out, size_t cap, size_t* used);
/* Same, over the dense layout of h2g_align_pairs_fetch_dense */
H2G_EXPORT h2g_status h2g_sam_format_paired_dense(const h2g_sam*, const uint8_t* codes1, const uint32_t* offs1, const char* quals1,
                                                  const char* name_bytes1, const uint32_t* name_offs1, const uint8_t* codes2,
                                                  const uint32_t* offs2, const char* quals2, const char* name_bytes2,
                                                  const uint32_t* name_offs2, size_t n_pairs, const h2g_pair_result* res,
                                                  const h2g_alnres* aln1, const uint64_t* aln_offs1, const h2g_alnres* aln2,
                                                  const uint64_t* aln_offs2, uint32_t khits, char* out, size_t cap, size_t* used);
/* Same, over the compact layout of h2g_align_fetch_compact / h2g_align_pairs_fetch_compact (byte offsets; records are prefixes of h2g_alnres).  A record
 * beyond H2G_MAX_EDITS edits needs its batch's long-edit area (h2g_sam_set_long_edits), as in every layout. */
H2G_EXPORT h2g_status h2g_sam_format_unpaired_compact(const h2g_sam*, const uint8_t* codes, const uint32_t* offs, const char* quals,
                                                      const char* name_bytes, const uint32_t* name_offs, size_t n_reads,
                                                      const h2g_read_result* res, const uint8_t* rec, const uint64_t* boffs /* [n+1] */,
                                                      char* out, size_t cap, size_t* used);
H2G_EXPORT h2g_status h2g_sam_format_paired_compact(const h2g_sam*, const uint8_t* codes1, const uint32_t* offs1, const char* quals1,
                                                    const char* name_bytes1, const uint32_t* name_offs1, const uint8_t* codes2,
                                                    const uint32_t* offs2, const char* quals2, const char* name_bytes2,
                                                    const uint32_t* name_offs2, size_t n_pairs, const h2g_pair_result* res,
                                                    const uint8_t* rec1, const uint64_t* boffs1, const uint8_t* rec2, const uint64_t* boffs2,
                                                    uint32_t khits, char* out, size_t cap, size_t* used);
#ifdef __cplusplus
}
#endif
#endif
