"""GPU go()-level parity (run with -m gpu): h2g_align_run / h2g_align_fetch through the C ABI against SAM written by
the REAL reference binary — committed golden SAM, and live oracle/_ref runs on fresh genomes when available."""
import os

import numpy as np
import pytest

import h2o_py as H
import sam_util as SU
from hisat2_amd import api

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpu_align(base, reads, qnames, bowtie2_dp=0, quals=None, options=()):
    """reads: (n, L) uint8 array or list of arrays"""
    lst = [np.asarray(r, dtype=np.uint8) for r in reads]
    codes = np.concatenate(lst)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in lst])]).astype(np.uint32)
    ix = api.Index(base, device=0)
    st = api.Stream(ix, max_reads=len(lst), max_bases=codes.size)
    st.set_reads(codes, offs, quals)
    st.set_read_names(qnames)
    p = st.align_params()
    p.bowtie2_dp = bowtie2_dp
    rest = p.apply_options(list(options))
    assert not rest, rest
    st.align_run(p)
    res, aln = st.align_fetch()
    c = st.counters()
    st.close()
    ix.close()
    return res, aln, c


def test_golden_sam(g1_index, golden_dir):
    names, seqs = H.read_fasta_reads(os.path.join(golden_dir, "reads_se.fa.gz"))
    res, aln, c = gpu_align(g1_index, seqs, names)
    refnames, want = SU.parse_sam(os.path.join(golden_dir, "ref_se_nospliced.sam.gz"))
    got = SU.render_selected(res, aln, refnames, seqs, names)
    assert (res["overflow"] == 0).all()
    for q in names:
        assert got[q] == want[q], q
    assert c.n_aligned == sum(1 for q in names if want[q][0][0] != 4)
    assert c.n_rank == int(res["nrank"].sum())


class _Out:
    def __init__(self, r):
        self.overflow, self.depth = int(r["overflow"]), int(r["depth"])


def _backend(base, reads, qnames, refnames, bowtie2_dp=0, quals=None, options=()):
    res, aln, _ = gpu_align(base, [reads[i] for i in range(len(reads))], qnames, bowtie2_dp=bowtie2_dp, quals=quals, options=options)
    got = SU.render_selected(res, aln, refnames, [reads[i] for i in range(len(reads))], qnames)
    return [_Out(r) for r in res], got


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")
@pytest.mark.parametrize("case", [
    dict(seed=201, nreads=20000, rdlen=101, sub=0.02, indel=0.002, nrate=0.002),
    dict(seed=202, nreads=10000, rdlen=150, sub=0.01, indel=0.001, nrate=0.001),
    dict(seed=203, nreads=10000, rdlen=101, sub=0.003, indel=0.0, nrate=0.0, lens=(120000,), repeats=300, gaps=0),
    dict(seed=204, nreads=5000, rdlen=36, sub=0.01, indel=0.0, nrate=0.0),
    dict(seed=205, nreads=6000, rdlen=300, sub=0.01, indel=0.003, nrate=0.001),
])
def test_live_reference(case):
    import fuzz_align as F
    bad, _ = F.run_case(verbose=3, backend=_backend, **case)
    assert bad == 0


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")
@pytest.mark.parametrize("dp", [1, 2])
def test_live_reference_bowtie2_dp(dp):
    """--bowtie2-dp 1 / 2: the SwAligner pass of hybridSearch runs inside the go() kernel (lane-sequential DP over HBM scratch)"""
    import functools
    import fuzz_align as F
    bad, _ = F.run_case(verbose=3, backend=functools.partial(_backend, bowtie2_dp=dp), seed=210 + dp, nreads=6000, rdlen=101, sub=0.02,
                        indel=0.006, nrate=0.001, extra=("--bowtie2-dp", str(dp)))
    assert bad == 0


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")
@pytest.mark.parametrize("case", [
    dict(seed=221, nreads=8000, rdlen=101, sub=0.01, indel=0.001, nrate=0.001, snps=250),
    dict(seed=222, nreads=5000, rdlen=101, sub=0.02, indel=0.002, nrate=0.0, snps=100),
    dict(seed=903, nreads=30000, rdlen=101, sub=0.02, indel=0.003, nrate=0.0, snps=80),
    dict(seed=904, nreads=20000, rdlen=150, sub=0.01, indel=0.002, nrate=0.001, snps=150, lens=(400000, 150000), repeats=40),
    dict(seed=923, nreads=20000, rdlen=76, sub=0.03, indel=0.005, nrate=0.0, snps=30),
])
def test_live_reference_graph_index(case):
    """go() on a SNP-graph index (hisat2-build --snp): graph LF, node walk, ALT-aware extension; reads from the alt haplotype"""
    import fuzz_align as F
    bad, _ = F.run_case(verbose=3, backend=_backend, **case)
    assert bad == 0


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")
def test_live_reference_graph_index_bowtie2_dp():
    """SwAligner pass on a graph index: SW edits that coincide with known variants get their ids (replace_edits_with_alts)"""
    import functools
    import fuzz_align as F
    bad, _ = F.run_case(verbose=3, backend=functools.partial(_backend, bowtie2_dp=2), seed=925, nreads=10000, rdlen=101, sub=0.02,
                        indel=0.006, nrate=0.001, snps=50, extra=("--bowtie2-dp", "2"))
    assert bad == 0


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")
@pytest.mark.parametrize("case", [
    dict(extra=("--sensitive",)),                                                       # wide linear kernel k_align<3, false> + SwAligner + L,0,-0.5
    dict(extra=("-k", "10")),
    dict(extra=("-k", "3", "--mp", "4,2", "--np", "3", "--rdg", "4,2", "--rfg", "7,2")),
    dict(extra=("--sensitive",), snps=60),
    dict(extra=("--sensitive",), rdlen=230, nreads=6000),
    dict(extra=("--bowtie2-dp", "2"), rdlen=250, nreads=5000, fastq=False),
])
def test_live_reference_options(case):
    import fuzz_align as F
    kw = dict(seed=231, nreads=10000, rdlen=101, sub=0.03, indel=0.005, nrate=0.002, fastq=True)
    kw.update(case)
    bad, _ = F.run_case(verbose=3, backend=_backend, **kw)
    assert bad == 0


def test_align_requires_names_and_valid_splice_scoring(g1_index, golden_dir):
    names, seqs = H.read_fasta_reads(os.path.join(golden_dir, "reads_se.fa.gz"))
    codes = np.concatenate(seqs)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in seqs])]).astype(np.uint32)
    ix = api.Index(g1_index)
    st = api.Stream(ix, max_reads=len(seqs), max_bases=codes.size)
    st.set_reads(codes, offs)
    with pytest.raises(api.H2GError):
        st.align_run()                      # names missing
    st.set_read_names(names)
    p = st.align_params()
    p.no_spliced_alignment = 0
    p.pen_canintronlen_type = 9
    with pytest.raises(api.H2GError):
        st.align_run(p)                     # splice scoring outside its range: refused, not approximated
    st.close()
    ix.close()


@pytest.mark.parametrize("spliced", [False, True])
def test_combine_with_golden_on_the_device(g1_index, golden_dir, spliced):
    """GenomeHit::combineWith (hi_aligner.h:1420-2025) through the C ABI ON THE DEVICE (h2g_combine_with -> k_combine -> hit_combine): the 1 500 anchor pairs of
    the reference's own class (tests/gen_golden.py combine; oracle/ref_probe.cpp `combine`) — return value, extent, score and every edit.  Round 5 held only the host
    instantiation to these vectors (tests/test_emul_golden.py::test_combine_with_golden); hipcc and g++ have differed on these sources before."""
    import ctypes as C
    names, seqs = H.read_fasta_reads(os.path.join(golden_dir, "reads_combine.fa.gz"))
    n = len(seqs)
    codes = np.concatenate(seqs).astype(np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint32)
    ix = api.Index(g1_index, device=0)
    st = api.Stream(ix, max_reads=n, max_bases=codes.size)
    st.set_reads(codes, offs)
    a = (api.GHit * n)()
    b = (api.GHit * n)()
    minsc = np.zeros(n, dtype=np.int64)
    lines = list(H.glines(golden_dir, "probe_combine_spliced.txt.gz" if spliced else "probe_combine.txt.gz"))
    assert len(lines) == n
    for i, nm in enumerate(names):
        rid, fw, tidx, roA, lenA, toA, roB, lenB, toB = (int(x) for x in nm.split("|"))
        assert rid == i
        for h, (ro, ln, to) in ((a[i], (roA, lenA, toA)), (b[i], (roB, lenB, toB))):
            h.read, h.fw, h.rdoff, h.len, h.trim5, h.trim3, h.tidx, h.toff, h.joinedOff, h.score, h.nedits, h.overflow = i, fw, ro, ln, 0, 0, tidx, to, 0, 0, 0, 0
        minsc[i] = int(lines[i].split()[2])
    p = st.align_params()
    p.no_spliced_alignment = 0 if spliced else 1
    ok = np.zeros(n, dtype=np.uint32)
    L = api.lib()
    L.h2g_combine_with.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    assert L.h2g_combine_with(st.h, C.byref(p), a, b, minsc.ctypes.data, n, ok.ctypes.data) == 0, L.h2g_last_error()
    kinds, ncomb = set(), 0
    for i, l in enumerate(lines):
        t = l.split()
        want_ok = int(t[4])
        assert int(ok[i]) == want_ok, (i, l)
        if not want_ok:
            continue
        ncomb += 1
        h = a[i]
        assert not h.overflow
        assert (h.rdoff, h.len, h.toff, h.score, h.nedits) == (int(t[5]), int(t[6]), int(t[7]), int(t[8]), int(t[9])), (i, l, (h.rdoff, h.len, h.toff, h.score, h.nedits))
        for k, tok in enumerate(t[10:]):
            ed = h.edits[k]
            f = tok.split(":")
            if f[1] == "S":                                   # intron: splLen, splDir, knownSpl in chr | qchr << 8 | (pad & 15) << 16, (pad >> 4) & 7, pad >> 7
                assert ed.type == 5 and ed.pos == int(f[0]), (i, l)
                assert (ed.chr | (ed.qchr << 8) | ((ed.pad & 15) << 16), (ed.pad >> 4) & 7, ed.pad >> 7) == (int(f[2]), int(f[3]), int(f[4])), (i, l)
            else:
                chr_, qchr = f[1].split(">")
                assert (ed.pos, chr(ed.chr), chr(ed.qchr), ed.type) == (int(f[0]), chr_, qchr, int(f[2])), (i, l, k)
            kinds.add(ed.type)
    assert ncomb > 700 and kinds >= ({1, 2, 3, 5} if spliced else {1, 2, 3})
    st.close(); ix.close()
