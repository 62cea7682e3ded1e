"""CPU go()-level parity: the device state machine (hisat2_amd/csrc/h2g_align.h), instantiated on the host by
tests/emul, against SAM written by the REAL reference binary (hisat2-align-s -p 1 --no-spliced-alignment): FLAG
(incl. the primary/secondary choice, which replays the per-read PRNG), RNAME, POS, CIGAR and AS:i of every line."""
import os

import pytest

import h2o_py as H
import sam_util as SU
from h2gemu_align import emu_align

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_golden_sam(g1_index, golden_dir):
    names, seqs = H.read_fasta_reads(os.path.join(golden_dir, "reads_se.fa.gz"))
    outs, recs = emu_align(g1_index, seqs, names)
    refnames, want = SU.parse_sam(os.path.join(golden_dir, "ref_se_nospliced.sam.gz"))
    got = SU.render(outs, recs, refnames, seqs, names)
    assert sum(1 for q in names if want[q][0][0] != 4) > 300
    for i, q in enumerate(names):
        assert outs[i].overflow == 0
        assert got[q] == want[q], q


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")
@pytest.mark.parametrize("case", [
    dict(seed=101, nreads=3000, rdlen=101, sub=0.02, indel=0.002, nrate=0.002),
    dict(seed=102, nreads=2000, rdlen=60, sub=0.01, indel=0.001, nrate=0.0),
    dict(seed=103, nreads=2000, rdlen=101, sub=0.003, indel=0.0, nrate=0.0, lens=(120000,), repeats=300, gaps=0),
    # the opt-in SwAligner pass inside hybridSearch (spliced_aligner.h:209): unconditional and conditional
    dict(seed=104, nreads=1500, rdlen=101, sub=0.02, indel=0.006, nrate=0.001, extra=("--bowtie2-dp", "2"), bowtie2_dp=2),
    dict(seed=105, nreads=1500, rdlen=75, sub=0.02, indel=0.01, nrate=0.0, extra=("--bowtie2-dp", "1"), bowtie2_dp=1),
    # SNP-graph index (hisat2-build --snp), reads drawn from the alternate haplotype
    dict(seed=106, nreads=1500, rdlen=101, sub=0.01, indel=0.001, nrate=0.001, snps=250),
    dict(seed=107, nreads=1000, rdlen=101, sub=0.02, indel=0.002, nrate=0.0, snps=100),
    # dense variants: local graph indexes with several '$' rows, SNP ids kept across combineWith, the SAM printer's
    # own gap left-alignment next to ALT gaps (seed 903 is the case that exposed all three)
    dict(seed=903, nreads=8000, rdlen=101, sub=0.02, indel=0.003, nrate=0.0, snps=80),
    dict(seed=923, nreads=5000, rdlen=76, sub=0.03, indel=0.005, nrate=0.0, snps=30),
    dict(seed=924, nreads=3000, rdlen=250, sub=0.01, indel=0.002, nrate=0.001, snps=100),
    # SwAligner pass on a graph index: replace_edits_with_alts (spliced_aligner.h:282)
    dict(seed=925, nreads=4000, rdlen=101, sub=0.02, indel=0.006, nrate=0.001, snps=50, extra=("--bowtie2-dp", "2"), bowtie2_dp=2),
    # the option surface of h2g_align_params: -k / --secondary / --mp / --np / --rdg / --rfg / --sp / --no-softclip / --score-min,
    # with FASTQ qualities so that the quality-aware penalties differ per base
    dict(seed=961, nreads=5000, rdlen=101, sub=0.025, indel=0.004, nrate=0.002, fastq=True, extra=("-k", "3", "--mp", "4,2", "--np", "3")),
    dict(seed=962, nreads=5000, rdlen=101, sub=0.025, indel=0.004, nrate=0.002, fastq=True, extra=("--rdg", "4,2", "--rfg", "7,2", "--score-min", "L,0,-0.4")),
    dict(seed=963, nreads=5000, rdlen=101, sub=0.025, indel=0.004, nrate=0.002, fastq=True, extra=("--secondary", "--sp", "3,1", "--score-min", "C,-18")),
    dict(seed=964, nreads=5000, rdlen=101, sub=0.025, indel=0.004, nrate=0.002, snps=60, extra=("--no-softclip", "-k", "4", "--mp", "5,1")),
    dict(seed=965, nreads=5000, rdlen=101, sub=0.03, indel=0.005, nrate=0.002, fastq=True, extra=("--sensitive",)),
    # reads longer than 256 bp: the combineWith score scan holds up to 512 positions (longer reads set the overflow bit)
    dict(seed=995, nreads=2500, rdlen=300, sub=0.01, indel=0.003, nrate=0.001),
    dict(seed=996, nreads=2500, rdlen=300, sub=0.01, indel=0.003, nrate=0.001, snps=100),
    # SwAligner pass on reads up to 256 bp (4 row chunks; mask-table keys hold 10-bit columns)
    dict(seed=771, nreads=3000, rdlen=250, sub=0.02, indel=0.006, nrate=0.002, extra=("--bowtie2-dp", "1"), bowtie2_dp=1),
    dict(seed=773, nreads=3000, rdlen=230, sub=0.03, indel=0.008, nrate=0.002, extra=("--sensitive",)),
    # --score-min below -254: SwAligner runs its 16-bit DP (aligner_sw.cpp:496); linear and graph index
    dict(seed=1604, nreads=1500, rdlen=101, sub=0.02, indel=0.008, nrate=0.001, extra=("--bowtie2-dp", "2", "--score-min", "L,0,-3"), bowtie2_dp=2),
    dict(seed=1606, nreads=1500, rdlen=101, sub=0.02, indel=0.008, nrate=0.001, snps=60, extra=("--bowtie2-dp", "2", "--score-min", "L,0,-2.6"), bowtie2_dp=2),
])
def test_live_reference(case):
    """Fresh genome + reads, index by the reference's builder, SAM by the reference's aligner."""
    import fuzz_align as F
    bad, _ = F.run_case(verbose=3, **case)
    assert bad == 0


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")
def test_live_reference_pairs_with_filtered_mates():
    """one mate N-filtered: the other is aligned alone; a lone mate 2 (`rightendonly`) reports into the sink's mate-2 list
    while the aligner keeps reading the empty mate-1 list back (no redundancy check, no best-score pruning)"""
    import fuzz_pairs as F
    for kw in (dict(seed=421, npairs=6000, rdlen=101, sub=0.03, repeats=60), dict(seed=411, npairs=15000, rdlen=101, sub=0.02)):
        bad, _ = F.run_case(verbose=3, mutate="nmask" if kw["seed"] == 421 else None, **kw)
        assert bad == 0


def test_golden_pairs_sam(g1_index, golden_dir):
    """Paired go() (pairReads, alignMate, lone mates, N filter) + the host-side finishRead mirror vs the
    reference's -1/-2 SAM: every line, in order."""
    import numpy as np
    import fuzz_pairs as F
    import pe_sink as PS
    _, s1 = H.read_fasta_reads(os.path.join(golden_dir, "reads_pe_1.fa.gz"))
    _, s2 = H.read_fasta_reads(os.path.join(golden_dir, "reads_pe_2.fa.gz"))
    import gzip, tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".sam", delete=False) as t:
        t.write(gzip.open(os.path.join(golden_dir, "ref_pe_nospliced.sam.gz"), "rt").read())
    refnames, want = F.parse_pe_sam(t.name)
    os.unlink(t.name)
    q = [str(i) for i in range(len(s1))]
    outs, r1, r2 = F.emu_pairs(g1_index, np.stack(s1), np.stack(s2), q, q)
    kinds = set()
    for i in range(len(s1)):
        got = PS.finish_pair(outs[i], r1, r2, i * SU.AL_MAX_RESULTS, refnames, (s1[i], s2[i]))
        assert outs[i].overflow == 0
        assert got == want[q[i]], i
        kinds.add(got[0][0] & 0xF)
    assert len(kinds) >= 2   # concordant and at least one other outcome class are exercised


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")
def test_live_reference_pairs_graph_index(monkeypatch):
    import fuzz_pairs as F
    monkeypatch.setattr(F, "SNPS", 200)
    bad, _ = F.run_case(verbose=3, seed=108, npairs=800, rdlen=101, sub=0.01)
    assert bad == 0


def _rc(m):
    import numpy as np
    return np.where(m[:, ::-1] > 3, 4, 3 - m[:, ::-1]).astype(np.uint8)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")
@pytest.mark.parametrize("opts,library", [(("--ff",), "ff"), (("--rf",), "rf"), (("--rf",), "fr"), (("-I", "250", "-X", "330"), "fr"),
                                           (("--nofw",), "fr"), (("--norc", "--rf"), "rf"), (("-X", "2000"), "fr")])
def test_live_reference_pair_policy_options(opts, library, monkeypatch):
    """-I / -X (PairedEndPolicy min / max fragment), --fr / --rf / --ff (gMate1fw, gMate2fw: the other mate's strand in alignMate,
    the orientation test of pairReads, the policy of peClassifyPair), --nofw / --norc (the strand pickNextReadToSearch skips, per
    mate of a pair) on libraries of the matching and of the wrong orientation"""
    import fuzz_pairs as F
    monkeypatch.setattr(F, "OPTS", opts)
    mut = {"fr": None, "ff": lambda a, b: (a, _rc(b)), "rf": lambda a, b: (_rc(a), _rc(b))}[library]
    bad, _ = F.run_case(verbose=3, seed=7500 + len(opts) + len(library), npairs=2500, rdlen=101, sub=0.012, frag_mean=300, frag_sd=40, mutate=mut)
    assert bad == 0


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")
@pytest.mark.parametrize("opt", ["--nofw", "--norc"])
def test_live_reference_unpaired_strand_options(opt):
    import fuzz_align as F
    bad, _ = F.run_case(7601, 3000, 101, 0.01, 0.001, 0.0, extra=(opt,), verbose=2)
    assert bad == 0


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")
@pytest.mark.parametrize("extra", [("--ignore-quals",), ("--ignore-quals", "--mp", "5,3")])
def test_live_reference_ignore_quals(extra):
    """--ignore-quals on FASTQ input: every mismatch costs --mp's maximum (COST_MODEL_CONSTANT); an explicit --mp brings the quality
    model back (aligner_seed_policy.cpp:279, :418)"""
    import fuzz_align as F
    bad, _ = F.run_case(8001, 3000, 101, 0.02, 0.001, 0.0, extra=extra, verbose=2, fastq=True)
    assert bad == 0
