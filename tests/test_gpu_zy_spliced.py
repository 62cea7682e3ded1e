"""GPU: spliced alignment through the command line (`--no-temp-splicesite`: every read independent of the others, the mode the
device path implements) — unpaired and paired, every SAM body line and the alignment summary byte-identical to the reference
binary's on genomes with planted GT..AG / GC..AG / AT..AC introns of 60 bp to 9 kbp."""
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


def _index(tmp, contigs):
    fa = os.path.join(str(tmp), "g.fa")
    synth.write_fasta(fa, contigs)
    base = os.path.join(str(tmp), "g")
    subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return base


def _both(tmp, base, inputs, extra):
    ref_sam, amd_sam = os.path.join(str(tmp), "ref.sam"), os.path.join(str(tmp), "amd.sam")
    ref_err, amd_err = os.path.join(str(tmp), "ref.err"), os.path.join(str(tmp), "amd.err")
    common = ["-f", "--no-temp-splicesite", "-x", base] + inputs + list(extra)
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-p", "1", "-S", ref_sam] + common, check=True, stdout=subprocess.DEVNULL, stderr=open(ref_err, "w"))
    subprocess.run([CLI, "-p", "4", "--batch", "3000", "-S", amd_sam] + common, check=True, stderr=open(amd_err, "w"))
    want = SL.body_lines(ref_sam)
    assert diff_lines(SL.body_lines(amd_sam), want) == 0
    assert open(amd_err).read() == open(ref_err).read()
    return want


@needs_ref
@pytest.mark.parametrize("seed,n,sub,extra", [
    (341, 20000, 0.005, ()),
    (342, 12000, 0.02, ()),
    (343, 8000, 0.01, ("-k", "3", "--pen-noncansplice", "6", "--min-intronlen", "50", "--max-intronlen", "6000")),
])
def test_unpaired_spliced_command_line(tmp_path, seed, n, sub, extra):
    import fuzz_spliced as F
    contigs, reads, _ = F.make_case(seed, n, sub=sub)
    base = _index(tmp_path, contigs)
    rfa = os.path.join(str(tmp_path), "r.fa")
    synth.write_reads_fasta(rfa, reads)
    want = _both(tmp_path, base, ["-U", rfa], extra)
    assert sum(1 for l in want if "N" in l.split("\t")[5]) > n // 5


@needs_ref
@pytest.mark.parametrize("seed,n,sub", [(351, 10000, 0.005), (352, 6000, 0.02)])
def test_paired_spliced_command_line(tmp_path, seed, n, sub):
    import fuzz_spliced_pairs as F
    contigs, m1, m2, _ = F.make_case(seed, n, sub=sub)
    base = _index(tmp_path, contigs)
    f1, f2 = os.path.join(str(tmp_path), "r1.fa"), os.path.join(str(tmp_path), "r2.fa")
    synth.write_reads_fasta(f1, m1)
    synth.write_reads_fasta(f2, m2)
    want = _both(tmp_path, base, ["-1", f1, "-2", f2], ())
    assert sum(1 for l in want if "N" in l.split("\t")[5]) > n // 5


@needs_ref
@pytest.mark.parametrize("seed,n,sub,known", [(361, 15000, 0.01, 0.7), (362, 8000, 0.03, 1.0)])
def test_known_splice_sites_command_line(tmp_path, seed, n, sub, known):
    """--known-splicesite-infile: the device database (h2g_index_set_splice_sites) through go() and the SAM formatter"""
    import fuzz_spliced as F
    contigs, reads, introns = F.make_case(seed, n, sub=sub)
    base = _index(tmp_path, contigs)
    rfa = os.path.join(str(tmp_path), "r.fa")
    synth.write_reads_fasta(rfa, reads)
    ss = os.path.join(str(tmp_path), "ss.txt")
    with open(ss, "w") as f:
        for t, l, r, d in F.known_sites(introns, seed, known):
            f.write("chr1\t%d\t%d\t%s\n" % (l, r, d))
    want = _both(tmp_path, base, ["-U", rfa], ("--known-splicesite-infile", ss))
    assert sum(1 for l in want if "N" in l.split("\t")[5]) > n // 5


@needs_ref
def test_known_splice_sites_pairs_command_line(tmp_path):
    import fuzz_spliced as FS
    import fuzz_spliced_pairs as F
    contigs, m1, m2, introns = F.make_case(371, 8000, sub=0.01)
    base = _index(tmp_path, contigs)
    f1, f2 = os.path.join(str(tmp_path), "r1.fa"), os.path.join(str(tmp_path), "r2.fa")
    synth.write_reads_fasta(f1, m1)
    synth.write_reads_fasta(f2, m2)
    ss = os.path.join(str(tmp_path), "ss.txt")
    with open(ss, "w") as f:
        for t, l, r, d in FS.known_sites(introns, 371, 0.8):
            f.write("chr1\t%d\t%d\t%s\n" % (l, r, d))
    _both(tmp_path, base, ["-1", f1, "-2", f2], ("--known-splicesite-infile", ss))


@needs_ref
@pytest.mark.parametrize("seed,n,P,sub", [(381, 30000, 4, 0.01), (382, 20000, 2, 0.005)])
def test_temporary_splice_sites_command_line(tmp_path, seed, n, P, sub):
    """the reference's DEFAULT mode: junctions found by earlier reads help later ones (SpliceSiteDB, window 1000 x -p).  The
    command line runs waves of <= 1000 x -p reads and merges each wave's junctions before the next: byte-identical to
    `hisat2 -p P --reorder`, whose output the shared database changes on about a quarter of the lines"""
    import fuzz_spliced as F
    contigs, reads, _ = F.make_case(seed, n, sub=sub)
    base = _index(tmp_path, contigs)
    rfa = os.path.join(str(tmp_path), "r.fa")
    synth.write_reads_fasta(rfa, reads)
    t = str(tmp_path)
    ref_sam, amd_sam, nt_sam = os.path.join(t, "ref.sam"), os.path.join(t, "amd.sam"), os.path.join(t, "ref_notemp.sam")
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", str(P), "--reorder", "-x", base, "-U", rfa, "-S", ref_sam],
                   check=True, stdout=subprocess.DEVNULL, stderr=open(os.path.join(t, "ref.err"), "w"))
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", str(P), "--reorder", "--no-temp-splicesite", "-x", base, "-U", rfa, "-S", nt_sam],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run([CLI, "-f", "-p", str(P), "-x", base, "-U", rfa, "-S", amd_sam], check=True, stderr=open(os.path.join(t, "amd.err"), "w"))
    want = SL.body_lines(ref_sam)
    assert diff_lines(SL.body_lines(amd_sam), want) == 0
    assert open(os.path.join(t, "amd.err")).read() == open(os.path.join(t, "ref.err")).read()
    # the window is the reference's thread count, not this program's: --ss-window 1000 x P with any -p gives the same lines
    subprocess.run([CLI, "-f", "-p", "1" if seed % 2 else "16", "--ss-window", str(1000 * P), "-x", base, "-U", rfa, "-S", amd_sam + "2"], check=True, stderr=open(os.path.join(t, "amd2.err"), "w"))
    assert diff_lines(SL.body_lines(amd_sam + "2"), want) == 0
    assert open(os.path.join(t, "amd2.err")).read() == open(os.path.join(t, "ref.err")).read()
    # --gpus 2: every wave is cut into two shards that run side by side (here on the one device), their junctions merged before the next wave
    subprocess.run([CLI, "-f", "-p", str(P), "--gpus", "2", "-x", base, "-U", rfa, "-S", amd_sam + "3"], check=True, stderr=open(os.path.join(t, "amd3.err"), "w"),
                   env=dict(os.environ, H2G_GPUS_SHARE_DEVICE="1"))
    assert diff_lines(SL.body_lines(amd_sam + "3"), want) == 0
    assert open(os.path.join(t, "amd3.err")).read() == open(os.path.join(t, "ref.err")).read()
    assert sum(1 for a, b in zip(want, SL.body_lines(nt_sam)) if a != b) > n // 20      # the database matters on this input


@needs_ref
def test_temporary_splice_sites_pairs_command_line(tmp_path):
    import fuzz_spliced_pairs as F
    contigs, m1, m2, _ = F.make_case(391, 12000, sub=0.01)
    base = _index(tmp_path, contigs)
    t = str(tmp_path)
    f1, f2 = os.path.join(t, "r1.fa"), os.path.join(t, "r2.fa")
    synth.write_reads_fasta(f1, m1)
    synth.write_reads_fasta(f2, m2)
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", "3", "--reorder", "-x", base, "-1", f1, "-2", f2, "-S", os.path.join(t, "ref.sam")],
                   check=True, stdout=subprocess.DEVNULL, stderr=open(os.path.join(t, "ref.err"), "w"))
    subprocess.run([CLI, "-f", "-p", "3", "-x", base, "-1", f1, "-2", f2, "-S", os.path.join(t, "amd.sam")], check=True, stderr=open(os.path.join(t, "amd.err"), "w"))
    assert diff_lines(SL.body_lines(os.path.join(t, "amd.sam")), SL.body_lines(os.path.join(t, "ref.sam"))) == 0
    assert open(os.path.join(t, "amd.err")).read() == open(os.path.join(t, "ref.err")).read()


def test_bare_default_invocation_equals_the_reference(tmp_path):
    """`hisat2 -f -x idx -U reads` with no other option: the reference's defaults are -p 1 and temporary splice sites, i.e. window 0 — every
    read sees the junctions of all reads before it (hisat2.cpp:3687).  The command line runs that as waves of one read; the SAM equals the
    committed output of `hisat2-align-s -f -p 1` (tests/golden/ref_se_spliced.sam.gz) byte for byte."""
    import gzip
    gold = os.path.join(ROOT, "tests", "golden")
    t = str(tmp_path)
    for k in range(1, 9):
        open(os.path.join(t, f"g1.{k}.ht2"), "wb").write(gzip.open(os.path.join(gold, f"g1.{k}.ht2.gz")).read())
    rfa = os.path.join(t, "r.fa")
    open(rfa, "wb").write(gzip.open(os.path.join(gold, "reads_se.fa.gz")).read())
    subprocess.run([CLI, "-f", "-x", os.path.join(t, "g1"), "-U", rfa, "-S", os.path.join(t, "amd.sam")], check=True, stderr=open(os.path.join(t, "amd.err"), "w"), timeout=600)
    got = [l for l in open(os.path.join(t, "amd.sam")).read().splitlines() if not l.startswith("@")]
    want = [l for l in gzip.open(os.path.join(gold, "ref_se_spliced.sam.gz"), "rt").read().splitlines() if not l.startswith("@")]
    assert len(want) > 100 and got == want
