"""SURVEY §8(f) N1 — the C++ SAM emitter (include/h2g_sam.h, hisat2_amd/csrc/h2g_sam.cpp) against the reference's own SAM
text: every non-header line byte for byte (QNAME … QUAL, AS ZS XN XM XO XG NM MD YS YT YF NH Zs, MAPQ, TLEN).  The
alignment records come from the host instantiation of the go() state machine here and from the GPU in test_gpu_sam.py."""
import ctypes as C
import os

import numpy as np
import pytest

import sam_lines as SL
import sam_util as SU
from hisat2_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")


def read_fa(path, n=None):
    code = {"A": 0, "C": 1, "G": 2, "T": 3, "N": 4}
    names, reads = [], []
    for ln in open(path):
        if ln[0] == ">":
            names.append(ln[1:].strip())
        else:
            reads.append(np.array([code[c] for c in ln.strip()], dtype=np.uint8))
    return names[:n], reads[:n]


def diff_lines(got, want, show=4):
    bad = 0
    for i, (a, b) in enumerate(zip(got, want)):
        if a != b:
            bad += 1
            if bad <= show:
                print("line", i, "\n  got ", a, "\n  want", b)
    return bad + abs(len(got) - len(want))


@needs_ref
@pytest.mark.parametrize("case", [
    dict(seed=301, nreads=4000, rdlen=101, sub=0.02, indel=0.003, nrate=0.004),                                   # Ns, YF:Z:NS, soft clips
    dict(seed=302, nreads=3000, rdlen=101, sub=0.004, indel=0.0, nrate=0.0, lens=(120000,), repeats=300, gaps=0),  # multi-mappers: MAPQ 0/1, ZS, NH, 256
    dict(seed=303, nreads=4000, rdlen=101, sub=0.02, indel=0.003, nrate=0.0, snps=80),                             # graph index: Zs:Z, XM/NM without variants
    dict(seed=304, nreads=2000, rdlen=150, sub=0.01, indel=0.004, nrate=0.001, snps=40),
    dict(seed=305, nreads=4000, rdlen=101, sub=0.02, indel=0.003, nrate=0.003, fastq=True),                        # FASTQ: quality-dependent penalties, QUAL column
    # scoring / reporting options: MAPQ is relative to --score-min, NH / secondary lines follow -k and --secondary
    dict(seed=306, nreads=3000, rdlen=101, sub=0.03, indel=0.004, nrate=0.001, fastq=True, extra=("--score-min", "L,0,-0.4", "--mp", "4,2", "-k", "3")),
    dict(seed=307, nreads=3000, rdlen=101, sub=0.004, indel=0.0, nrate=0.0, lens=(120000,), repeats=300, gaps=0, extra=("--secondary", "--rdg", "4,2", "--no-softclip")),
])
def test_unpaired_lines_identical(case):
    import fuzz_align as F
    from h2gemu_align import emu_align
    assert C.sizeof(SU.AlnRec) == C.sizeof(api.AlnRes)
    bad, tmp = F.run_case(verbose=2, **case)
    assert bad == 0
    names, reads = read_fa(os.path.join(tmp, "r.fa"))
    quals = None
    if case.get("fastq"):
        lines = open(os.path.join(tmp, "r.fq"), "rb").read().split(b"\n")
        quals = np.frombuffer(b"".join(lines[3::4]), dtype=np.uint8)
    opts = case.get("extra", ())
    outs, recs = emu_align(os.path.join(tmp, "g"), reads, names, quals=quals, options=opts)
    res, aln = SL.emu_to_abi(outs, recs)
    got = SL.format_unpaired(SL.load_sam_lib(), os.path.join(tmp, "g"), reads, names, res, aln, quals=quals, options=opts)
    want = SL.body_lines(os.path.join(tmp, "ref.sam"))
    assert len(want) >= case["nreads"]
    assert diff_lines(got, want) == 0
    assert SL.LAST_SUMMARY == open(os.path.join(tmp, "ref.err")).read()      # the alignment summary on stderr


@needs_ref
@pytest.mark.parametrize("snps,case", [
    (0, dict(seed=311, npairs=2500, rdlen=101, sub=0.02)),
    (0, dict(seed=312, npairs=1500, rdlen=101, sub=0.06, frag_mean=250, frag_sd=80)),     # many lone mates / discordant / unaligned
    (60, dict(seed=313, npairs=2000, rdlen=101, sub=0.02)),
    (0, dict(seed=314, npairs=1500, rdlen=101, sub=0.01, mutate="flip")),                 # every 4th mate 2 reverse-complemented: YT:Z:DP
    (0, dict(seed=315, npairs=3000, rdlen=101, sub=0.03, repeats=60, mutate="nmask")),    # N-filtered mates: YF:Z:NS + lone-mate paths
    (0, dict(seed=316, npairs=2500, rdlen=101, sub=0.025, opts=("-k", "2", "--mp", "5,3", "--score-min", "L,0,-0.35"))),
    (0, dict(seed=317, npairs=2500, rdlen=101, sub=0.025, repeats=80, mutate="nmask", opts=("--secondary",))),
])
def test_paired_lines_identical(monkeypatch, snps, case):
    import fuzz_pairs as F
    monkeypatch.setattr(F, "SNPS", snps)
    case = dict(case)
    opts = case.pop("opts", ())
    monkeypatch.setattr(F, "OPTS", opts)
    bad, tmp = F.run_case(verbose=2, **case)
    assert bad == 0
    n1, m1 = read_fa(os.path.join(tmp, "r1.fa"))
    n2, m2 = read_fa(os.path.join(tmp, "r2.fa"))
    outs, r1, r2 = F.emu_pairs(os.path.join(tmp, "g"), np.stack(m1), np.stack(m2), n1, n2)
    n = len(m1)
    res = (api.PairResult * n)()
    a1 = (api.AlnRes * (n * api.PAIR_RES_CAP))()
    a2 = (api.AlnRes * (n * api.PAIR_RES_CAP))()
    assert C.sizeof(api.PairResult) == C.sizeof(outs[0])
    C.memmove(res, outs, C.sizeof(res))
    for i in range(n):
        for m, (src, dst) in enumerate(((r1, a1), (r2, a2))):
            for k in range(min(outs[i].nres[m], api.PAIR_RES_CAP)):
                C.memmove(C.byref(dst[i * api.PAIR_RES_CAP + k]), C.byref(src[i * SU.AL_MAX_RESULTS + k]), C.sizeof(api.AlnRes))
    khits = int(opts[opts.index("-k") + 1]) if "-k" in opts else (10 if snps else 5)
    got = SL.format_paired(SL.load_sam_lib(), os.path.join(tmp, "g"), m1, m2, n1, n2, res, a1, a2, khits, options=opts)
    want = SL.body_lines(os.path.join(tmp, "ref.sam"))
    assert diff_lines(got, want) == 0
    assert SL.LAST_SUMMARY == open(os.path.join(tmp, "ref.err")).read()


@needs_ref
@pytest.mark.parametrize("seed,sub,extra", [(321, 0.005, ()), (322, 0.02, ()),
                                            (343, 0.01, ("-k", "3", "--pen-noncansplice", "6", "--min-intronlen", "50", "--max-intronlen", "6000")),
                                            (344, 0.01, ("--dta-cufflinks", "--rna-strandness", "R")), (345, 0.005, ("--dta", "--rna-strandness", "F"))])
def test_spliced_lines_identical(seed, sub, extra):
    """spliced alignment (the reference's default mode, --no-temp-splicesite): introns placed by combineWith, CIGAR N, XS:A,
    MD / NM around the intron, MAPQ (best vs second best by the whole hisat2_score: splice bits included) and NH — every line
    byte-identical"""
    import fuzz_spliced as F
    from h2gemu_align import emu_align
    bad, tmp = F.run_case(seed, 3000, sub=sub, verbose=2, extra=extra)
    assert bad == 0
    names, reads = read_fa(os.path.join(tmp, "r.fa"))
    outs, recs = emu_align(os.path.join(tmp, "g"), reads, names, no_spliced=0, options=list(extra))
    res, aln = SL.emu_to_abi(outs, recs)
    got = SL.format_unpaired(SL.load_sam_lib(), os.path.join(tmp, "g"), reads, names, res, aln, options=list(extra))
    want = SL.body_lines(os.path.join(tmp, "ref.sam"))
    assert sum(1 for l in want if "N" in l.split("\t")[5]) > 1000
    assert diff_lines(got, want) == 0
    assert SL.LAST_SUMMARY == open(os.path.join(tmp, "ref.err")).read()


@needs_ref
@pytest.mark.parametrize("seed,sub", [(331, 0.01)])
def test_spliced_pairs_lines_identical(seed, sub):
    """paired spliced alignment: mates found by alignMate across introns, concordance, and TLEN, which leaves out the introns
    inside the mates (AlnRes::setFragmentLength over refcoord_right(), aligner_result.h:1256, :1631)"""
    import fuzz_spliced_pairs as F
    bad, tmp = F.run_case(seed, 1500, sub=sub, show=3)
    assert bad == 0
    want = SL.body_lines(os.path.join(tmp, "ref.sam"))
    assert sum(1 for l in want if "N" in l.split("\t")[5]) > 300
    assert SL.LAST_SUMMARY == open(os.path.join(tmp, "ref.err")).read()


@needs_ref
@pytest.mark.parametrize("seed,extra", [
    (501, ("-k", "3", "--pen-noncansplice", "6", "--min-intronlen", "50", "--max-intronlen", "6000")),
    (502, ("--pen-cansplice", "3", "--pen-canintronlen", "S,-2,0.1", "--pen-noncanintronlen", "L,1,0.001")),
])
def test_spliced_scoring_options(seed, extra):
    """splice scoring options (hisat2.cpp:1631-1688): penalties, the intron-length SimpleFuncs, the intron length window"""
    import fuzz_spliced as F
    bad, _ = F.run_case(seed, 2000, sub=0.01, verbose=2, extra=extra)
    assert bad == 0


@needs_ref
@pytest.mark.parametrize("seed,sub,known", [(611, 0.01, 0.6), (612, 0.005, 1.0)])
def test_known_splice_sites_lines_identical(seed, sub, known):
    """--known-splicesite-infile (static splice-site database): joins through database sites of full alignments
    (spliced_aligner.h:409-676) and of partial ones (:685-811, :1365-1496), known splices cost nothing and rank above novel ones
    (transcript bits of hisat2_score) — every line byte-identical; the database changes about a fifth of the reads"""
    import fuzz_spliced as F
    from h2gemu_align import emu_align
    bad, tmp = F.run_case(seed, 2500, sub=sub, verbose=2, known=known)
    assert bad == 0
    names, reads = read_fa(os.path.join(tmp, "r.fa"))
    sites = api.read_splice_site_file(os.path.join(tmp, "ss.txt"), ["chr1"])
    outs, recs = emu_align(os.path.join(tmp, "g"), reads, names, no_spliced=0, splice_sites=sites)
    res, aln = SL.emu_to_abi(outs, recs)
    opts = ["--known-splicesite-infile", os.path.join(tmp, "ss.txt")]
    got = SL.format_unpaired(SL.load_sam_lib(), os.path.join(tmp, "g"), reads, names, res, aln, options=opts)
    want = SL.body_lines(os.path.join(tmp, "ref.sam"))
    assert diff_lines(got, want) == 0
    assert SL.LAST_SUMMARY == open(os.path.join(tmp, "ref.err")).read()


@needs_ref
@pytest.mark.parametrize("seed,known", [(5101, 0.0), (5102, 0.5)])
def test_spliced_lines_multi_junction_multi_contig(seed, known, monkeypatch):
    """the rich generator: reads drawn from the spliced transcript over 20..90 bp exons (two or three junctions per read), the genes on
    the SECOND of three contigs (site text ids > 0, a leading contig with an N gap = two fragments), N runs inside exons, whole-gene
    copies and processed pseudogenes (the intron-less copy competes with the spliced placement) — every line byte-identical"""
    monkeypatch.setenv("H2G_FUZZ_RICH", "1")
    monkeypatch.setenv("H2G_FUZZ_MULTI", "0.5")
    import fuzz_spliced as F
    from h2gemu_align import emu_align
    bad, tmp = F.run_case(seed, 3000, sub=0.01, verbose=2, known=known, indel=0.003)
    assert bad == 0
    names, reads = read_fa(os.path.join(tmp, "r.fa"))
    sites, opts = None, []
    if known:
        sites = api.read_splice_site_file(os.path.join(tmp, "ss.txt"), ["lead", "chr1", "tail"])
        opts = ["--known-splicesite-infile", os.path.join(tmp, "ss.txt")]
    outs, recs = emu_align(os.path.join(tmp, "g"), reads, names, no_spliced=0, splice_sites=sites)
    res, aln = SL.emu_to_abi(outs, recs)
    got = SL.format_unpaired(SL.load_sam_lib(), os.path.join(tmp, "g"), reads, names, res, aln, options=opts)
    want = SL.body_lines(os.path.join(tmp, "ref.sam"))
    assert sum(1 for l in want if l.split("\t")[5].count("N") >= 2) > 100            # reads across two junctions
    assert len({l.split("\t")[2] for l in want}) == 4                                  # lead, chr1, tail and *
    assert diff_lines(got, want) == 0
    assert SL.LAST_SUMMARY == open(os.path.join(tmp, "ref.err")).read()


@needs_ref
def test_temporary_splice_sites_multi_junction_multi_contig(monkeypatch):
    """the default mode on the rich generator, at -p 3"""
    monkeypatch.setenv("H2G_FUZZ_RICH", "1")
    monkeypatch.setenv("H2G_FUZZ_MULTI", "0.4")
    import temp_splice as T
    bad, _ = T.run_case(5201, 7000, P=3, show=3, known=0.3)
    assert bad == 0


@needs_ref
def test_known_splice_sites_pairs_lines_identical():
    """pairs with a splice-site database: TLEN leaves the longest database intron between the mates out (aligner_result.h:1669)"""
    import fuzz_spliced_pairs as F
    bad, tmp = F.run_case(711, 1500, sub=0.01, show=3, known=0.8)
    assert bad == 0


@needs_ref
def test_temporary_splice_sites_wave_scheme():
    """the reference's default mode through the host instantiation: waves of 1000 x P reads, each wave's junctions merged into the
    database before the next (tests/temp_splice.py) == `hisat2 -p 2 --reorder`, line for line"""
    import temp_splice as T
    bad, _ = T.run_case(811, 5000, P=2, show=3)
    assert bad == 0


@needs_ref
def test_temporary_splice_sites_with_known_file_and_outfile():
    """default mode + --known-splicesite-infile: the file's sites stay visible to every read and are never replaced by a read's;
    --novel-splicesite-outfile (SpliceSiteDB::print splice_site.cpp:565: per-site read counts, the 70 % cut-off, the merging of
    near-identical sites) == the reference's file (temp_splice.run_case compares it in every case)"""
    import temp_splice as T
    bad, tmp = T.run_case(821, 5000, P=2, show=3, known=0.5)
    assert bad == 0
    assert sum(1 for _ in open(os.path.join(tmp, "ref.ss"))) > 50


@needs_ref
def test_novel_splicesite_outfile_pairs_no_database():
    """--no-temp-splicesite --novel-splicesite-outfile without any database: the sites are written but never read (hisat2.cpp:4092);
    both mates' lines count"""
    import fuzz_spliced_pairs as F
    bad, tmp = F.run_case(712, 1500, sub=0.01, show=3, novel_out=True)
    assert bad == 0
    assert sum(1 for _ in open(os.path.join(tmp, "ref.ss"))) > 50


@needs_ref
def test_no_templatelen_adjustment_pairs():
    """--no-templatelen-adjustment: setMateParams without the database (aln_sink.h:2070-2076), TLEN keeps the introns between mates"""
    import fuzz_spliced_pairs as F
    bad, tmp = F.run_case(713, 1500, sub=0.01, show=3, known=0.8, extra=("--no-templatelen-adjustment",))
    assert bad == 0


@needs_ref
def test_golden_spliced_default_mode_p1(tmp_path):
    """tests/golden/ref_se_spliced.sam.gz = `hisat2-align-s -p 1` in its default mode on the golden reads: at -p 1 the window is 0,
    every read sees the junctions of all reads before it, i.e. waves of ONE read (what only a test can afford)"""
    import gzip
    import temp_splice as T
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for k in range(1, 9):
        open(os.path.join(str(tmp_path), f"g1.{k}.ht2"), "wb").write(gzip.open(os.path.join(gold, f"g1.{k}.ht2.gz")).read())
    rfa = os.path.join(str(tmp_path), "r.fa")
    open(rfa, "wb").write(gzip.open(os.path.join(gold, "reads_se.fa.gz")).read())
    names, reads = read_fa(rfa)
    base = os.path.join(str(tmp_path), "g1")
    got, db = T.wave_run(base, reads, names, 1, lambda lo, hi, o, r, a, k, W: T.format_wave(base, reads, names, lo, hi, o, r, a, k, W))
    want = [l for l in gzip.open(os.path.join(gold, "ref_se_spliced.sam.gz"), "rt").read().splitlines() if not l.startswith("@")]
    assert diff_lines(got, want) == 0


@needs_ref
def test_spliced_on_snp_graph_lines_identical(monkeypatch):
    """spliced alignment on a SNP-graph index (the reference's genome_snp indexes in their default mode): reads from the alternate
    haplotype over planted introns, a splice-site file on top — introns + known variants in one alignment (Zs:Z next to XS:A)"""
    import fuzz_spliced as F
    from h2gemu_align import emu_align
    monkeypatch.setenv("H2G_FUZZ_SNPS", "150")
    bad, tmp = F.run_case(1021, 2000, sub=0.01, verbose=2, known=0.5)
    assert bad == 0
    names, reads = read_fa(os.path.join(tmp, "r.fa"))
    sites = api.read_splice_site_file(os.path.join(tmp, "ss.txt"), ["chr1"])
    outs, recs = emu_align(os.path.join(tmp, "g"), reads, names, no_spliced=0, splice_sites=sites)
    res, aln = SL.emu_to_abi(outs, recs)
    got = SL.format_unpaired(SL.load_sam_lib(), os.path.join(tmp, "g"), reads, names, res, aln, options=["--known-splicesite-infile", os.path.join(tmp, "ss.txt")])
    want = SL.body_lines(os.path.join(tmp, "ref.sam"))
    assert sum(1 for l in want if "Zs:Z" in l and "N" in l.split("\t")[5]) > 50
    assert diff_lines(got, want) == 0


@needs_ref
def test_spliced_pairs_on_snp_graph_lines_identical(monkeypatch):
    import fuzz_spliced_pairs as F
    monkeypatch.setenv("H2G_FUZZ_SNPS", "200")
    bad, _ = F.run_case(1022, 1200, sub=0.01, show=3, known=0.6)
    assert bad == 0


@needs_ref
def test_temporary_splice_sites_on_snp_graph(monkeypatch):
    import temp_splice as T
    monkeypatch.setenv("H2G_FUZZ_SNPS", "200")
    bad, _ = T.run_case(1023, 4500, P=2, show=3)
    assert bad == 0


@needs_ref
@pytest.mark.parametrize("seed,extra", [(346, ("--rna-strandness", "FR", "--dta")), (347, ("--rna-strandness", "RF"))])
def test_spliced_pairs_stranded_library(seed, extra):
    """--rna-strandness: XS:A on every aligned line from the mate and its strand (sam.h:940-966); --dta: anchor minima 15 / 20"""
    import fuzz_spliced_pairs as F
    bad, _ = F.run_case(seed, 1200, sub=0.01, show=3, extra=extra)
    assert bad == 0


@needs_ref
@pytest.mark.parametrize("seed,snps", [(1231, 0), (1232, 150)])
def test_splice_site_alt_index_lines_identical(seed, snps):
    """--ss / --exon indexes (the reference's _tran indexes, here with and without --snp): splice sites are graph edges the FM search
    and the ALT-aware extension run through (alignWithALTs_recur hi_aligner.h:3083 / :3425, findSSOffs :2482) and known sites of
    the database (SpliceSiteDB::read(gfm, alts)); local indexes without a variant stay linear inside the graph index"""
    import fuzz_tran as T
    bad, _ = T.run_case(seed, 2000, 0.01, snps, verbose=3, lines=True)
    assert bad == 0


@needs_ref
def test_splice_site_alt_index_pairs_and_waves(monkeypatch):
    import fuzz_tran as T
    import temp_splice as W
    assert T.run_pairs(1233, 1000, 0.01, 200, show=3)[0] == 0
    monkeypatch.setenv("H2G_FUZZ_TRAN", "1")
    assert W.run_case(1234, 4500, P=2, show=3)[0] == 0


@needs_ref
@pytest.mark.parametrize("seed,every,kw", [(2101, 25, {}), (2102, 40, {"ht_file": False}), (2103, 15, {"extra": ("--spliced",)}),
                                           (2104, 30, {"extra": ("-k", "10", "--secondary")})])
def test_haplotype_option(seed, every, kw):
    """--haplotype (GraphPolicy::useHaplotype): an index built with --snp + --haplotype (or without the file: one haplotype per SNP,
    gfm.h:1645); reads from donors that carry the index's haplotypes or arbitrary SNP subsets.  An ALT is only walked when a
    haplotype carries it together with the ALTs already taken (alignWithALTs_recur hi_aligner.h:2898-2996, :3251-3331) — equal to the
    reference read for read, on an input where the option changes its output"""
    import fuzz_haplotype as H
    bad, tmp = H.run_case(seed, 3000, every=every, verbose=2, **kw)
    assert bad == 0
    import sam_util as SU
    _, a = SU.parse_sam(os.path.join(tmp, "ref.sam"))
    _, b = SU.parse_sam(os.path.join(tmp, "ref_nohap.sam"))
    assert sum(1 for q in a if a[q] != b[q]) >= (5 if kw.get("ht_file", True) else 1)


@needs_ref
@pytest.mark.parametrize("mode", ["notemp", "temp", "tran"])
def test_spliced_on_real_sequence(mode, monkeypatch):
    """tests/fuzz_real.py: introns at GT..AG pairs of the reference's own chr22 example contig, reads from that transcript (two or three
    junctions per read).  Real sequence has what random genomes lack — here inverted segmental duplications: a read whose reverse
    complement aligns spliced and perfectly must still get its forward strand aligned, because nextBWT / align give a strand
    bestSplicedUnp more partial searches (hi_aligner.h:4680, :5520; the term was a constant 0 until this case found it)"""
    import fuzz_real as R
    import fuzz_spliced as F
    monkeypatch.setattr(F, "make_case", R.make_case)
    if mode == "tran":
        monkeypatch.setenv("H2G_FUZZ_TRAN", "1")
    if mode == "notemp":
        bad, tmp = F.run_case(9001, 4000, 0.005, known=0.0, verbose=2)
        import sam_util as SU
        _, want = SU.parse_sam(os.path.join(tmp, "ref.sam"))
        assert sum(1 for q in want if len(want[q]) > 1) >= 5              # multi-mappers in the duplications
    else:
        import temp_splice as T
        bad, _ = T.run_case(9001, 4000, P=2, show=3)
    assert bad == 0


@needs_ref
@pytest.mark.parametrize("extra", [("--no-mixed",), ("--no-discordant",), ("--no-mixed", "--no-discordant", "-k", "3")])
def test_no_mixed_no_discordant(extra, monkeypatch):
    """--no-mixed / --no-discordant (ReportingParams::mixed / discord): mates without a paired alignment are reported unaligned; a pair
    with one alignment per mate is not turned into a discordant pair — lines and summary equal the reference's on pairs of which 15 %
    have their second mate drawn from elsewhere"""
    monkeypatch.setenv("H2G_FUZZ_CHIMERA", "0.15")
    import fuzz_spliced_pairs as F
    bad, tmp = F.run_case(7301, 2000, sub=0.03, show=3, extra=extra)
    assert bad == 0
    assert SL.LAST_SUMMARY == open(os.path.join(tmp, "ref.err")).read()


@needs_ref
def test_read_group_no_sq_omit_sec_seq(monkeypatch):
    """--rg-id / --rg (the @RG header line and RG:Z: on aligned and unaligned records), --no-sq, --omit-sec-seq ('*' for SEQ / QUAL of
    secondary lines) — header (up to @PG's command line) and every line equal the reference's; pairs with multi-mappers"""
    monkeypatch.setenv("H2G_FUZZ_RICH", "1")              # gene copies: multi-mapping pairs, i.e. secondary lines
    import fuzz_spliced_pairs as F
    extra = ("--rg-id", "grp1", "--rg", "SM:sample one", "--rg", "PL:ILLUMINA", "--no-sq", "--omit-sec-seq", "-k", "4")
    bad, tmp = F.run_case(7701, 1500, sub=0.02, show=3, extra=extra)
    assert bad == 0
    ref_head = [l.rstrip("\n") for l in open(os.path.join(tmp, "ref.sam")) if l.startswith("@") and not l.startswith("@PG")]
    got_head = [l for l in SL.LAST_HEADER.splitlines() if not l.startswith("@PG")]
    assert got_head == ref_head and any(l.startswith("@RG\tID:grp1\tSM:sample one\tPL:ILLUMINA") for l in got_head)
    body = SL.body_lines(os.path.join(tmp, "ref.sam"))
    assert all("RG:Z:grp1" in l for l in body) and any(l.split("\t")[9] == "*" for l in body)


@needs_ref
@pytest.mark.parametrize("paired", [False, True])
def test_new_summary(paired):
    """--new-summary: the "HISAT2 summary stats:" text (aln_sink.h:1659-1679) equals the reference's stderr"""
    if paired:
        import fuzz_spliced_pairs as F
        bad, tmp = F.run_case(7801, 1200, sub=0.03, show=3, extra=("--new-summary",))
    else:
        import fuzz_spliced as F
        from h2gemu_align import emu_align
        bad, tmp = F.run_case(7802, 2000, sub=0.02, verbose=2, extra=("--new-summary",))
        names, reads = read_fa(os.path.join(tmp, "r.fa"))
        outs, recs = emu_align(os.path.join(tmp, "g"), reads, names, no_spliced=0)
        res, aln = SL.emu_to_abi(outs, recs)
        SL.format_unpaired(SL.load_sam_lib(), os.path.join(tmp, "g"), reads, names, res, aln, options=["--new-summary"])
    assert bad == 0
    want = open(os.path.join(tmp, "ref.err")).read()
    assert want.startswith("HISAT2 summary stats:") and SL.LAST_SUMMARY == want


@needs_ref
def test_remove_chrname():
    """--remove-chrname: chr1 -> 1 in @SQ, RNAME / RNEXT and in --novel-splicesite-outfile (hisat2.cpp:3962)"""
    import fuzz_spliced_pairs as F
    bad, tmp = F.run_case(7901, 1200, sub=0.01, show=3, extra=("--remove-chrname",), novel_out=True)
    assert bad == 0
    assert "@SQ\tSN:1\t" in SL.LAST_HEADER and open(os.path.join(tmp, "ref.ss")).read().startswith("1\t")


def test_compact_layout_is_checked_before_a_line_is_written(g1_index):
    """the compact and the fixed-row formatters must refuse a long record (nedits > 32) whose long-edit area is missing or too short, and a compact buffer
    whose bytes are not whole records — H2G_ERR_ARG, never an out-of-bounds read or a silently missing read"""
    import ctypes as C
    from hisat2_amd import api
    L = SL.load_sam_lib()
    h = C.c_void_p()
    assert L.h2g_sam_open(g1_index.encode(), C.byref(h)) == 0
    codes = np.zeros(50, dtype=np.uint8); offs = np.array([0, 50], dtype=np.uint32)
    nb, noffs = SL.flat_names(["r0"])
    res = np.zeros(1, dtype=api.READ_RESULT_DTYPE)
    res["nres"] = 1; res["nselect"] = 1; res["best"] = -6; res["secbest"] = -(2 ** 31)
    rec = (api.AlnRes * 1)()
    rec[0].fw, rec[0].tidx, rec[0].toff, rec[0].len, rec[0].nedits, rec[0].score = 1, 0, 100, 50, 0, -6
    out = C.create_string_buffer(1 << 16); used = C.c_size_t(0)
    fc = L.h2g_sam_format_unpaired_compact
    fc.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    raw = bytes(rec)[:40]
    buf = np.frombuffer(raw, dtype=np.uint8).copy()
    boffs = np.array([0, 40], dtype=np.uint64)
    args = lambda b, bo: (h, codes.ctypes.data, offs.ctypes.data, None, nb, noffs.ctypes.data, 1, res.ctypes.data, b.ctypes.data, bo.ctypes.data, out, 1 << 16, C.byref(used))
    assert fc(*args(buf, boffs)) == 0 and used.value > 0                              # a well-formed record prints
    assert fc(*args(buf, np.array([0, 32], dtype=np.uint64))) != 0                    # the read's bytes end inside its record
    assert fc(*args(buf[:0].copy() if False else buf, np.array([0, 0], dtype=np.uint64))) != 0   # nselect = 1 but no record in the read's bytes
    rec[0].nedits = 40; rec[0].edits[0].pos = 0                                         # a long record: one marker entry inline, the list in the long-edit area
    buf2 = np.frombuffer(bytes(rec)[:56], dtype=np.uint8).copy()
    assert fc(*args(buf2, np.array([0, 56], dtype=np.uint64))) != 0                   # ... which was never set
    assert L.h2g_sam_format_unpaired(h, codes.ctypes.data, offs.ctypes.data, None, nb, noffs.ctypes.data, 1, res.ctypes.data, C.byref(rec), out, 1 << 16, C.byref(used)) != 0
    L.h2g_sam_close(h)
