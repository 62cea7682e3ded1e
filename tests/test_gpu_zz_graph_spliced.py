"""GPU: spliced alignment on SNP-graph indexes through the command line (the reference's genome_snp indexes in their default
mode) — unpaired with a splice-site file, paired, and temporary splice sites at -p 3; every SAM line and the summary identical to
the reference binary's.  (Sorted last on purpose: the graph + spliced units are the newest kernels.)"""
import os
import subprocess

import pytest

import sam_lines as SL
from hisat2_amd import synth
from test_sam_lines import diff_lines

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "hisat2_amd", "hisat2-align-amd")
REF = os.path.join(ROOT, "oracle", "_ref")
needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "hisat2-align-s")), reason="needs oracle/_ref")


def _graph_index(tmp, contigs, seed, every):
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs)
    synth.write_snps(os.path.join(tmp, "g.snp"), synth.make_snps(contigs, seed + 5, every=every))
    base = os.path.join(tmp, "g")
    subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", "--snp", os.path.join(tmp, "g.snp"), fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return base


def _compare(tmp, base, inputs, ref_opts, amd_opts):
    ref_sam, amd_sam = os.path.join(tmp, "ref.sam"), os.path.join(tmp, "amd.sam")
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-x", base, "-S", ref_sam] + inputs + ref_opts, check=True, stdout=subprocess.DEVNULL,
                   stderr=open(os.path.join(tmp, "ref.err"), "w"))
    subprocess.run([CLI, "-f", "-x", base, "-S", amd_sam] + inputs + amd_opts, check=True, stderr=open(os.path.join(tmp, "amd.err"), "w"))
    want = SL.body_lines(ref_sam)
    assert diff_lines(SL.body_lines(amd_sam), want) == 0
    assert open(os.path.join(tmp, "amd.err")).read() == open(os.path.join(tmp, "ref.err")).read()
    return want


@needs_ref
def test_unpaired_known_sites_on_snp_graph(tmp_path):
    import fuzz_spliced as F
    tmp = str(tmp_path)
    contigs, reads, introns = F.make_case(1031, 12000, sub=0.01)
    base = _graph_index(tmp, contigs, 1031, 150)
    rfa = os.path.join(tmp, "r.fa")
    synth.write_reads_fasta(rfa, reads)
    ss = os.path.join(tmp, "ss.txt")
    with open(ss, "w") as f:
        for t, l, r, d in F.known_sites(introns, 1031, 0.6):
            f.write("chr1\t%d\t%d\t%s\n" % (l, r, d))
    opts = ["--no-temp-splicesite", "--known-splicesite-infile", ss]
    want = _compare(tmp, base, ["-U", rfa], ["-p", "1"] + opts, ["-p", "4"] + opts)
    assert sum(1 for l in want if "N" in l.split("\t")[5]) > 2000


@needs_ref
def test_paired_on_snp_graph(tmp_path):
    import fuzz_spliced_pairs as F
    tmp = str(tmp_path)
    contigs, m1, m2, _ = F.make_case(1032, 6000, sub=0.01)
    base = _graph_index(tmp, contigs, 1032, 200)
    f1, f2 = os.path.join(tmp, "r1.fa"), os.path.join(tmp, "r2.fa")
    synth.write_reads_fasta(f1, m1)
    synth.write_reads_fasta(f2, m2)
    _compare(tmp, base, ["-1", f1, "-2", f2], ["-p", "1", "--no-temp-splicesite"], ["-p", "4", "--no-temp-splicesite"])


@needs_ref
def test_temporary_splice_sites_on_snp_graph(tmp_path):
    import fuzz_spliced as F
    tmp = str(tmp_path)
    contigs, reads, _ = F.make_case(1033, 15000, sub=0.01)
    base = _graph_index(tmp, contigs, 1033, 200)
    rfa = os.path.join(tmp, "r.fa")
    synth.write_reads_fasta(rfa, reads)
    _compare(tmp, base, ["-U", rfa], ["-p", "3", "--reorder"], ["-p", "3"])


@needs_ref
def test_dta_and_stranded_library_command_line(tmp_path):
    """--dta-cufflinks + --rna-strandness on a linear index: anchor minima 15 / 20, unknown-strand junctions dropped, XS:A from the strand"""
    import fuzz_spliced_pairs as F
    tmp = str(tmp_path)
    contigs, m1, m2, _ = F.make_case(1041, 6000, sub=0.01)
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs)
    base = os.path.join(tmp, "g")
    subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f1, f2 = os.path.join(tmp, "r1.fa"), os.path.join(tmp, "r2.fa")
    synth.write_reads_fasta(f1, m1)
    synth.write_reads_fasta(f2, m2)
    opts = ["--no-temp-splicesite", "--dta-cufflinks", "--rna-strandness", "RF"]
    _compare(tmp, base, ["-1", f1, "-2", f2], ["-p", "1"] + opts, ["-p", "4"] + opts)


@needs_ref
def test_splice_site_alt_index_command_line(tmp_path):
    """a --ss / --exon / --snp index (the shape of genome_snp_tran) in the reference's default mode at -p 3"""
    import fuzz_spliced as F
    import fuzz_tran as T
    tmp = str(tmp_path)
    contigs, reads, introns = F.make_case(1051, 12000, sub=0.01)
    base = T.build(tmp, contigs, introns, 1051, snps=200)
    rfa = os.path.join(tmp, "r.fa")
    synth.write_reads_fasta(rfa, reads)
    want = _compare(tmp, base, ["-U", rfa], ["-p", "3", "--reorder"], ["-p", "3"])
    assert sum(1 for l in want if "N" in l.split("\t")[5]) > 3000


@needs_ref
def test_novel_splicesite_outfile_and_templatelen_command_line(tmp_path):
    """pairs on a linear index in the default mode with a splice-site file: --novel-splicesite-outfile (SpliceSiteDB::print: read
    counts per site, the 70 % cut-off, near-identical sites merged) equals the reference's file, and --no-templatelen-adjustment
    (TLEN keeps the database introns between the mates) equals its lines"""
    import fuzz_spliced as FS
    import fuzz_spliced_pairs as F
    tmp = str(tmp_path)
    contigs, m1, m2, introns = F.make_case(1061, 9000, sub=0.01)
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs)
    base = os.path.join(tmp, "g")
    subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f1, f2, ss = os.path.join(tmp, "r1.fa"), os.path.join(tmp, "r2.fa"), os.path.join(tmp, "ss.txt")
    synth.write_reads_fasta(f1, m1)
    synth.write_reads_fasta(f2, m2)
    with open(ss, "w") as f:
        for _, l, r, d in FS.known_sites(introns, 1061, 0.5):
            f.write("chr1\t%d\t%d\t%s\n" % (l, r, d))
    common = ["--known-splicesite-infile", ss, "--no-templatelen-adjustment", "--novel-splicesite-outfile"]
    want = _compare(tmp, base, ["-1", f1, "-2", f2], ["-p", "3", "--reorder"] + common + [os.path.join(tmp, "ref.ss")], ["-p", "3"] + common + [os.path.join(tmp, "amd.ss")])
    assert sum(1 for l in want if "N" in l.split("\t")[5]) > 2000
    got_ss = open(os.path.join(tmp, "amd.ss")).read()
    assert got_ss == open(os.path.join(tmp, "ref.ss")).read() and got_ss.count("\n") > 50


@needs_ref
@pytest.mark.parametrize("mode", [["--no-spliced-alignment"], ["--no-temp-splicesite"]])
def test_haplotype_command_line(tmp_path, mode):
    """--haplotype on an index built with --snp + --haplotype: ALT combinations no haplotype carries are not walked; lines and summary
    equal the reference's (the go() units with H2G_HAPLOTYPE run both the spliced and the unspliced mode)"""
    import numpy as np
    import fuzz_haplotype as H
    tmp = str(tmp_path)
    rng = np.random.default_rng(1071)
    contigs = [rng.integers(0, 4, size=300000, dtype=np.uint8), rng.integers(0, 4, size=90000, dtype=np.uint8)]
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs)
    snps = synth.make_snps(contigs, 1076, every=25)
    synth.write_snps(os.path.join(tmp, "g.snp"), snps)
    lines, clusters, per = H.make_haplotypes(snps, rng)
    with open(os.path.join(tmp, "g.haplotype"), "w") as f:
        f.write("\n".join(lines) + "\n")
    base = os.path.join(tmp, "g")
    subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", "--snp", os.path.join(tmp, "g.snp"), "--haplotype", os.path.join(tmp, "g.haplotype"), fa, base],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    parts = []
    for d in range(4):                                         # two donors carrying haplotypes of the index, two carrying arbitrary subsets
        chosen = []
        for c, hs in zip(clusters, per):
            chosen += (hs[int(rng.integers(0, len(hs)))] if rng.random() < 0.8 else []) if d < 2 else [s for s in c if rng.random() < 0.5]
        parts.append(synth.make_reads(synth.apply_snps(contigs, chosen), 5000, 101, 1080 + d, sub_rate=0.004, indel_rate=0.0, n_rate=0.0)[0])
    rfa = os.path.join(tmp, "r.fa")
    synth.write_reads_fasta(rfa, np.concatenate(parts))
    want = _compare(tmp, base, ["-U", rfa], ["-p", "2", "--reorder", "--haplotype"] + mode, ["-p", "2", "--haplotype"] + mode)
    nohap = os.path.join(tmp, "nohap.sam")
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", "2", "--reorder", "-x", base, "-U", rfa, "-S", nohap] + mode, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert sum(1 for a, b in zip(want, SL.body_lines(nohap)) if a != b) > 100      # the option matters on this input


@needs_ref
def test_spliced_on_real_sequence_command_line(tmp_path):
    """the default mode on the reference's own chr22 example contig with introns at its GT..AG pairs (tests/fuzz_real.py): reads whose
    two strands both align, one of them spliced, in the inverted duplications of real sequence (bestSplicedUnp, hi_aligner.h:4680)"""
    import fuzz_real as R
    tmp = str(tmp_path)
    contigs, reads, _ = R.make_case(9001, 12000)
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs)
    base = os.path.join(tmp, "g")
    subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rfa = os.path.join(tmp, "r.fa")
    synth.write_reads_fasta(rfa, reads)
    want = _compare(tmp, base, ["-U", rfa], ["-p", "2", "--reorder"], ["-p", "2"])
    assert sum(1 for l in want if "NH:i:1" not in l and "N" in l.split("\t")[5]) > 20      # spliced multi-mappers


@needs_ref
@pytest.mark.parametrize("opts,library", [(["--rf"], "rf"), (["--ff", "--no-mixed"], "ff"), (["-I", "250", "-X", "330", "--no-discordant"], "fr"),
                                           (["--nofw"], "fr"), (["--norc", "--rf", "--no-spliced-alignment"], "rf")])
def test_pair_policy_options_command_line(tmp_path, opts, library):
    """-I / -X, --fr / --rf / --ff, --nofw / --norc, --no-mixed / --no-discordant through the command line, unspliced and spliced
    (the options other than -X are compiled into the units that carry the splice-site database, which run either mode)"""
    import numpy as np
    tmp = str(tmp_path)
    contigs = synth.make_genome([300000, 120000], 1091, n_gaps=2, gap_len=300, repeats=6, repeat_len=500)
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs)
    base = os.path.join(tmp, "g")
    subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    m1, m2 = synth.make_pairs(contigs, 8000, 101, 1092, frag_mean=300, frag_sd=40, sub_rate=0.012)
    rc = lambda m: np.where(m[:, ::-1] > 3, 4, 3 - m[:, ::-1]).astype(np.uint8)
    if library == "ff":
        m2 = rc(m2)
    elif library == "rf":
        m1, m2 = rc(m1), rc(m2)
    f1, f2 = os.path.join(tmp, "r1.fa"), os.path.join(tmp, "r2.fa")
    synth.write_reads_fasta(f1, m1)
    synth.write_reads_fasta(f2, m2)
    mode = [] if "--no-spliced-alignment" in opts else ["--no-temp-splicesite"]
    want = _compare(tmp, base, ["-1", f1, "-2", f2], ["-p", "2", "--reorder"] + mode + opts, ["-p", "2"] + mode + opts)
    assert sum(1 for l in want if l.split("\t")[1] != "77" and l.split("\t")[1] != "141") > 4000


@needs_ref
def test_command_line_and_raw_reads(tmp_path):
    """-c (sequences on the command line, unpaired and paired) and -r (one sequence per line): names 0, 1, … and the lines of the reference"""
    import numpy as np
    tmp = str(tmp_path)
    contigs = synth.make_genome([120000], 1101, n_gaps=0, gap_len=0, repeats=2, repeat_len=300)
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs)
    base = os.path.join(tmp, "g")
    subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    m1, m2 = synth.make_pairs(contigs, 40, 75, 1102, frag_mean=250, frag_sd=30, sub_rate=0.01)
    txt = lambda m: ["".join("ACGTN"[int(x)] for x in r) for r in m]
    s1, s2 = txt(m1), txt(m2)
    for inputs in (["-c", "-U", ",".join(s1)], ["-c", "-1", ",".join(s1), "-2", ",".join(s2)]):
        os.makedirs(os.path.join(tmp, "a"), exist_ok=True)
        _compare(tmp, base, inputs, ["-p", "2", "--reorder", "--no-spliced-alignment"], ["-p", "2", "--no-spliced-alignment"])
    raw = os.path.join(tmp, "r.txt")
    open(raw, "w").write("\n".join(s1) + "\n")
    want = _compare(tmp, base, ["-r", "-U", raw], ["-p", "2", "--reorder", "--no-spliced-alignment"], ["-p", "2", "--no-spliced-alignment"])
    assert [l.split("\t")[0] for l in want[:3]] == ["0", "1", "2"]
