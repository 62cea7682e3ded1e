"""GPU: the whole drop-in path — reads in, SAM text out — against the reference binary: h2g_align_run / h2g_align_pairs_run on
the device, include/h2g_sam.h on the host, and the `hisat2-align-amd` command line (FASTA and FASTQ, unpaired and paired,
linear and SNP-graph index).  Every non-header SAM line must be byte-identical to `hisat2-align-s --no-spliced-alignment`."""
import os
import subprocess

import numpy as np
import pytest

import sam_lines as SL
from test_sam_lines import diff_lines, read_fa

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "hisat2_amd", "hisat2-align-amd")
needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")


@needs_ref
@pytest.mark.parametrize("case", [
    dict(seed=401, nreads=20000, rdlen=101, sub=0.02, indel=0.003, nrate=0.004),
    dict(seed=402, nreads=8000, rdlen=101, sub=0.004, indel=0.0, nrate=0.0, lens=(120000,), repeats=300, gaps=0),
    dict(seed=403, nreads=20000, rdlen=101, sub=0.02, indel=0.003, nrate=0.0, snps=80),
    dict(seed=404, nreads=10000, rdlen=101, sub=0.02, indel=0.003, nrate=0.003, fastq=True),
    dict(seed=405, nreads=10000, rdlen=101, sub=0.03, indel=0.004, nrate=0.001, fastq=True, extra=("--score-min", "L,0,-0.4", "--mp", "4,2", "-k", "3", "--rdg", "4,2")),
    dict(seed=406, nreads=8000, rdlen=101, sub=0.02, indel=0.003, nrate=0.001, snps=70, extra=("--secondary", "--no-softclip", "--np", "2")),
])
def test_unpaired_sam_text(case):
    """device alignments -> h2g_sam_format_unpaired == the reference's lines; then the same through the command line"""
    import fuzz_align as F
    from test_gpu_align import _backend, gpu_align
    bad, tmp = F.run_case(verbose=2, backend=_backend, **case)
    assert bad == 0
    names, reads = read_fa(os.path.join(tmp, "r.fa"))
    quals = None
    if case.get("fastq"):
        lines = open(os.path.join(tmp, "r.fq"), "rb").read().split(b"\n")
        quals = np.frombuffer(b"".join(lines[3::4]), dtype=np.uint8)
    opts = list(case.get("extra", ()))
    res, aln, _ = gpu_align(os.path.join(tmp, "g"), reads, names, quals=quals, options=opts)
    want = SL.body_lines(os.path.join(tmp, "ref.sam"))
    got = SL.format_unpaired(SL.load_sam_lib(), os.path.join(tmp, "g"), reads, names, res, aln, quals=quals, options=opts)
    assert diff_lines(got, want) == 0
    out = os.path.join(tmp, "amd.sam")
    rd = os.path.join(tmp, "r.fq" if case.get("fastq") else "r.fa")
    subprocess.run([CLI, "-x", os.path.join(tmp, "g"), "-q" if case.get("fastq") else "-f", "-U", rd, "--no-spliced-alignment", "-S", out,
                    "--batch", "7000", "-p", "5"] + opts, check=True, stderr=open(os.path.join(tmp, "amd.err"), "w"))
    assert diff_lines(SL.body_lines(out), want) == 0
    assert open(os.path.join(tmp, "amd.err")).read() == open(os.path.join(tmp, "ref.err")).read()          # alignment summary
    hdr = [l for l in open(out) if l.startswith("@")]
    ref_hdr = [l for l in open(os.path.join(tmp, "ref.sam")) if l.startswith("@")]
    assert hdr[:-1] == ref_hdr[:-1] and hdr[-1].startswith("@PG\tID:hisat2\tPN:hisat2\tVN:")      # @HD, @SQ identical; @PG differs by CL


@needs_ref
@pytest.mark.parametrize("snps,case", [
    (0, dict(seed=411, npairs=15000, rdlen=101, sub=0.02)),
    (0, dict(seed=412, npairs=6000, rdlen=101, sub=0.01, mutate="flip")),
    (60, dict(seed=413, npairs=10000, rdlen=101, sub=0.02)),
    (0, dict(seed=414, npairs=12000, rdlen=101, sub=0.03, repeats=60, mutate="nmask")),
])
def test_paired_command_line(monkeypatch, snps, case):
    import fuzz_pairs as F
    from test_gpu_pairs import _backend
    from hisat2_amd import api
    monkeypatch.setattr(F, "SNPS", snps)
    bad, tmp = F.run_case(verbose=2, backend=_backend, stride=api.PAIR_RES_CAP, **case)
    assert bad == 0
    out = os.path.join(tmp, "amd.sam")
    subprocess.run([CLI, "-x", os.path.join(tmp, "g"), "-f", "-1", os.path.join(tmp, "r1.fa"), "-2", os.path.join(tmp, "r2.fa"),
                    "--no-spliced-alignment", "-S", out, "--batch", "4000", "-p", "3"], check=True, stderr=open(os.path.join(tmp, "amd.err"), "w"))
    assert diff_lines(SL.body_lines(out), SL.body_lines(os.path.join(tmp, "ref.sam"))) == 0
    assert open(os.path.join(tmp, "amd.err")).read() == open(os.path.join(tmp, "ref.err")).read()


@needs_ref
def test_command_line_skip_upto_trim_no_unal():
    """-s / -u / -5 / -3 / --no-unal and gzipped input behave as in the reference"""
    import gzip
    import shutil
    import fuzz_align as F
    from test_gpu_align import _backend
    bad, tmp = F.run_case(verbose=2, backend=_backend, seed=421, nreads=6000, rdlen=101, sub=0.02, indel=0.003, nrate=0.004, fastq=True)
    assert bad == 0
    with open(os.path.join(tmp, "r.fq"), "rb") as fi, gzip.open(os.path.join(tmp, "r.fq.gz"), "wb") as fo:
        shutil.copyfileobj(fi, fo)
    opts = ["-s", "500", "-u", "3000", "-5", "4", "-3", "7", "--no-unal"]
    ref = os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")
    subprocess.run([ref, "-q", "-p", "1", "--no-spliced-alignment", "-x", os.path.join(tmp, "g"), "-U", os.path.join(tmp, "r.fq"), "-S",
                    os.path.join(tmp, "ref2.sam")] + opts, check=True, stderr=open(os.path.join(tmp, "ref2.err"), "w"))
    subprocess.run([CLI, "-q", "-p", "4", "--no-spliced-alignment", "-x", os.path.join(tmp, "g"), "-U", os.path.join(tmp, "r.fq.gz"), "-S",
                    os.path.join(tmp, "amd2.sam"), "--batch", "1000"] + opts, check=True, stderr=open(os.path.join(tmp, "amd2.err"), "w"))
    want = SL.body_lines(os.path.join(tmp, "ref2.sam"))
    assert 1000 < len(want) < 3000
    assert diff_lines(SL.body_lines(os.path.join(tmp, "amd2.sam")), want) == 0
    assert open(os.path.join(tmp, "amd2.err")).read() == open(os.path.join(tmp, "ref2.err")).read()


def test_command_line_refuses_what_is_not_built(tmp_path):
    r = subprocess.run([CLI, "-x", "nonexistent", "-U", "x.fq", "--no-spliced-alignment", "--un", "y.fq"], capture_output=True, text=True)
    assert r.returncode != 0 and "is not built" in r.stderr


def test_dense_fetch_equals_slot_fetch(g1_index, golden_dir):
    """h2g_align_fetch_dense / h2g_align_pairs_fetch_dense return the same records as the slot layout, back to back"""
    import ctypes as C
    import h2o_py as H
    from hisat2_amd import api
    names, seqs = H.read_fasta_reads(os.path.join(golden_dir, "reads_se.fa.gz"))
    codes = np.concatenate(seqs)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in seqs])]).astype(np.uint32)
    ix = api.Index(g1_index)
    st = api.Stream(ix, max_reads=len(seqs), max_bases=codes.size)
    st.set_reads(codes, offs)
    st.set_read_names(names)
    st.align_run()
    res, aln = st.align_fetch()
    dres, daln, doffs = st.align_fetch_dense()
    assert res.tobytes() == dres.tobytes()
    tot = 0
    for i in range(len(seqs)):
        assert int(doffs[i]) == tot
        for k in range(int(res[i]["nselect"])):
            a, b = aln[i * api.ALN_CAP + k], daln[tot + k]
            assert (a.fw, a.tidx, a.toff, a.len, a.trim5, a.trim3, a.nedits, a.score) == (b.fw, b.tidx, b.toff, b.len, b.trim5, b.trim3, b.nedits, b.score)
            assert bytes(a.edits)[:12 * a.nedits] == bytes(b.edits)[:12 * a.nedits]
        tot += int(res[i]["nselect"])
    assert int(doffs[len(seqs)]) == tot and tot > 300
    st.close()
    _, s1 = H.read_fasta_reads(os.path.join(golden_dir, "reads_pe_1.fa.gz"))
    _, s2 = H.read_fasta_reads(os.path.join(golden_dir, "reads_pe_2.fa.gz"))
    c1, o1 = np.concatenate(s1), np.concatenate([[0], np.cumsum([len(r) for r in s1])]).astype(np.uint32)
    c2, o2 = np.concatenate(s2), np.concatenate([[0], np.cumsum([len(r) for r in s2])]).astype(np.uint32)
    q = [str(i) for i in range(len(s1))]
    st = api.Stream(ix, max_reads=len(s1), max_bases=max(c1.size, c2.size))
    st.set_reads(c1, o1)
    st.set_read_names(q)
    st.set_mates(c2, o2, q)
    st.align_pairs_run()
    pres, a1, a2 = st.align_pairs_fetch()
    dpres, d1, do1, d2, do2 = st.align_pairs_fetch_dense()
    assert bytes(pres) == bytes(dpres)
    for m, (slot, dense, doff) in enumerate(((a1, d1, do1), (a2, d2, do2))):
        for i in range(len(s1)):
            for k in range(min(pres[i].nres[m], api.PAIR_RES_CAP)):
                a, b = slot[i * api.PAIR_RES_CAP + k], dense[int(doff[i]) + k]
                assert (a.fw, a.tidx, a.toff, a.len, a.nedits, a.score) == (b.fw, b.tidx, b.toff, b.len, b.nedits, b.score)
                assert bytes(a.edits)[:12 * a.nedits] == bytes(b.edits)[:12 * a.nedits]
    st.close()
    ix.close()


@needs_ref
def test_command_line_ragged_and_odd_reads():
    """ragged lengths (1 .. 260 bp), lower case, IUPAC codes, names with blanks / over 255 characters, all-N and very short reads"""
    import fuzz_align as F
    from test_gpu_align import _backend
    from hisat2_amd import synth
    bad, tmp = F.run_case(verbose=2, backend=_backend, seed=431, nreads=500, rdlen=101, sub=0.01, indel=0.001, nrate=0.001)
    assert bad == 0
    rng = np.random.default_rng(17)
    names, seqs = read_fa(os.path.join(tmp, "r.fa"))
    contig = "".join(l.strip() for l in open(os.path.join(tmp, "g.fa")) if l[0] != ">")
    fq = os.path.join(tmp, "odd.fq")
    with open(fq, "w") as f:
        for i in range(3000):
            L = int(rng.choice([1, 2, 5, 12, 19, 20, 21, 33, 50, 75, 101, 150, 200, 260]))
            p = int(rng.integers(0, len(contig) - 300))
            s = list(contig[p:p + L].replace("N", "A"))
            for k in range(len(s)):
                if rng.random() < 0.02:
                    s[k] = "ACGT"[int(rng.integers(0, 4))]
                if rng.random() < 0.01:
                    s[k] = "RYKMN"[int(rng.integers(0, 5))]
            s = "".join(s)
            if i % 7 == 0:
                s = s.lower()
            if i % 97 == 0:
                s = "N" * len(s)
            if rng.random() < 0.5:
                s = s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))
            name = ("read %d with blanks\tand a tab" % i) if i % 5 == 0 else ("r%d" % i) if i % 11 else "x" * 300 + str(i)
            q = "".join(chr(int(c)) for c in rng.integers(35, 74, size=len(s)))
            f.write("@%s\n%s\n+\n%s\n" % (name, s, q))
    ref = os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")
    subprocess.run([ref, "-q", "-p", "1", "--no-spliced-alignment", "-x", os.path.join(tmp, "g"), "-U", fq, "-S", os.path.join(tmp, "ref3.sam")], check=True,
                   stderr=open(os.path.join(tmp, "ref3.err"), "w"))
    subprocess.run([CLI, "-q", "-p", "3", "--no-spliced-alignment", "-x", os.path.join(tmp, "g"), "-U", fq, "-S", os.path.join(tmp, "amd3.sam"), "--batch", "700"],
                   check=True, stderr=open(os.path.join(tmp, "amd3.err"), "w"))
    want = SL.body_lines(os.path.join(tmp, "ref3.sam"))
    assert len(want) >= 3000
    assert diff_lines(SL.body_lines(os.path.join(tmp, "amd3.sam")), want) == 0
    ref_err = [l for l in open(os.path.join(tmp, "ref3.err")) if not l.startswith("Warning")]
    assert open(os.path.join(tmp, "amd3.err")).read() == "".join(ref_err)


@needs_ref
@pytest.mark.parametrize("snps", [0, 70])
def test_command_line_paired_fastq_unequal_mates(monkeypatch, snps):
    """pairs from FASTQ files (quality-dependent penalties on both mates), mate 2 shorter than mate 1"""
    import fuzz_pairs as F
    from test_gpu_pairs import _backend
    from hisat2_amd import api
    monkeypatch.setattr(F, "SNPS", snps)
    bad, tmp = F.run_case(verbose=2, backend=_backend, stride=api.PAIR_RES_CAP, seed=441 + snps, npairs=6000, rdlen=101, sub=0.025, mutate="nmask")
    assert bad == 0
    rng = np.random.default_rng(23)
    for m, cut in (("1", 101), ("2", 76)):
        names, seqs = read_fa(os.path.join(tmp, "r%s.fa" % m))
        with open(os.path.join(tmp, "q%s.fq" % m), "w") as f:
            for nm, s in zip(names, seqs):
                s = s[:cut]
                q = rng.choice(np.array([2, 8, 15, 20, 25, 30, 37, 40]) + 33, size=len(s))
                f.write("@%s/%s\n%s\n+\n%s\n" % (nm, m, "".join("ACGTN"[c] for c in s), "".join(chr(int(c)) for c in q)))
    ref = os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")
    args = ["-q", "--no-spliced-alignment", "-x", os.path.join(tmp, "g"), "-1", os.path.join(tmp, "q1.fq"), "-2", os.path.join(tmp, "q2.fq")]
    subprocess.run([ref, "-p", "1"] + args + ["-S", os.path.join(tmp, "ref4.sam")], check=True, stderr=open(os.path.join(tmp, "ref4.err"), "w"))
    subprocess.run([CLI, "-p", "4", "--batch", "2500"] + args + ["-S", os.path.join(tmp, "amd4.sam")], check=True, stderr=open(os.path.join(tmp, "amd4.err"), "w"))
    want = SL.body_lines(os.path.join(tmp, "ref4.sam"))
    assert diff_lines(SL.body_lines(os.path.join(tmp, "amd4.sam")), want) == 0
    assert open(os.path.join(tmp, "amd4.err")).read() == "".join(l for l in open(os.path.join(tmp, "ref4.err")) if not l.startswith("Warning"))
    if snps == 0:
        # -2 shorter than -1: the reference's message and exit status 1 (pat.cpp:420), and the process ENDS (the threaded pipeline used to
        # hang here, a detached writer waiting on a condition variable of the frame being left)
        with open(os.path.join(tmp, "q2short.fq"), "w") as f:
            f.writelines(open(os.path.join(tmp, "q2.fq")).readlines()[:4 * 3100])
        r = subprocess.run([CLI, "-p", "4", "--batch", "2500", "-q", "--no-spliced-alignment", "-x", os.path.join(tmp, "g"), "-1", os.path.join(tmp, "q1.fq"),
                            "-2", os.path.join(tmp, "q2short.fq"), "-S", os.path.join(tmp, "short.sam")], capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "fewer reads in file specified with -2" in r.stderr


@needs_ref
@pytest.mark.parametrize("paired", [False, True])
def test_command_line_gpus_front_end(paired):
    """--gpus N: batches round-robin over N streams (here sharing the one device: H2G_GPUS_SHARE_DEVICE), completed in order —
    the SAM file and the summary are those of the single-stream run, i.e. the reference's"""
    if paired:
        import fuzz_pairs as F
        from test_gpu_pairs import _backend
        from hisat2_amd import api
        bad, tmp = F.run_case(verbose=2, backend=_backend, stride=api.PAIR_RES_CAP, seed=431, npairs=9000, rdlen=101, sub=0.02)
        rd = ["-1", os.path.join(tmp, "r1.fa"), "-2", os.path.join(tmp, "r2.fa")]
    else:
        import fuzz_align as F
        from test_gpu_align import _backend
        bad, tmp = F.run_case(verbose=2, backend=_backend, seed=432, nreads=15000, rdlen=101, sub=0.02, indel=0.003, nrate=0.002)
        rd = ["-U", os.path.join(tmp, "r.fa")]
    assert bad == 0
    want = SL.body_lines(os.path.join(tmp, "ref.sam"))
    for gpus in (2, 3):
        out = os.path.join(tmp, f"amd{gpus}.sam")
        subprocess.run([CLI, "-x", os.path.join(tmp, "g"), "-f"] + rd + ["--no-spliced-alignment", "-S", out, "--batch", "1700", "-p", "3", "--gpus", str(gpus)],
                       check=True, stderr=open(os.path.join(tmp, "amd.err"), "w"), env=dict(os.environ, H2G_GPUS_SHARE_DEVICE="1"))
        assert diff_lines(SL.body_lines(out), want) == 0
        assert open(os.path.join(tmp, "amd.err")).read() == open(os.path.join(tmp, "ref.err")).read()
