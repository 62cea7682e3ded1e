"""GPU paired-end go() parity (-m gpu): h2g_align_pairs_run through the C ABI; the report events are fed to the
host-side finishRead mirror (tests/pe_sink.py) and every SAM line (FLAG incl. pairing bits, RNAME, POS, CIGAR, AS:i,
line order) is compared with oracle/_ref/hisat2-align-s -1/-2 output."""
import os

import pytest

import fuzz_pairs as F
from hisat2_amd import api, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _backend(base, m1, m2, q1, q2):
    c1, o1 = synth.flatten_reads(m1)
    c2, o2 = synth.flatten_reads(m2)
    ix = api.Index(base, device=0)
    st = api.Stream(ix, max_reads=len(m1), max_bases=c1.size)
    st.set_reads(c1, o1)
    st.set_read_names(q1)
    st.set_mates(c2, o2, q2)
    p = st.align_params()
    rest = p.apply_options(list(F.OPTS))          # scoring / reporting options of the case (fuzz_pairs.OPTS)
    assert not rest, rest
    st.align_pairs_run(p)
    res, a1, a2 = st.align_pairs_fetch()
    st.close()
    ix.close()
    return res, a1, a2


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")
@pytest.mark.parametrize("case", [
    dict(seed=301, npairs=20000, rdlen=101, sub=0.01),
    dict(seed=302, npairs=8000, rdlen=101, sub=0.04),
    dict(seed=303, npairs=8000, rdlen=75, sub=0.02, frag_mean=400, frag_sd=250),
    dict(seed=304, npairs=6000, rdlen=150, sub=0.02, lens=(200000, 80000), repeats=150, gaps=4),
])
def test_live_reference_pairs(case):
    bad, _ = F.run_case(verbose=3, backend=_backend, stride=api.PAIR_RES_CAP, **case)
    assert bad == 0


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")
def test_live_reference_pairs_graph_index(monkeypatch):
    """paired go() on a SNP-graph index, pairs drawn from the alternate haplotype"""
    monkeypatch.setattr(F, "SNPS", 200)
    bad, _ = F.run_case(verbose=3, backend=_backend, stride=api.PAIR_RES_CAP, seed=311, npairs=6000, rdlen=101, sub=0.01)
    assert bad == 0


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")
def test_live_reference_pairs_dense_graph_index(monkeypatch):
    """variants every ~100 bp, 125 bp mates, wide fragments (the case that exposed the multi-'$' local index bug)"""
    monkeypatch.setattr(F, "SNPS", 100)
    bad, _ = F.run_case(verbose=3, backend=_backend, stride=api.PAIR_RES_CAP, seed=913, npairs=20000, rdlen=125, sub=0.02, frag_mean=350, frag_sd=120)
    assert bad == 0


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")), reason="needs oracle/_ref")
@pytest.mark.parametrize("opts", [("--sensitive",), ("-k", "9"), ("-k", "2", "--mp", "5,3", "--score-min", "L,0,-0.35")])
def test_live_reference_pairs_options(monkeypatch, opts):
    """option surface on pairs; -k > 5 / --sensitive run the wide linear kernel (k_align_pairs<false, 1>)"""
    monkeypatch.setattr(F, "OPTS", opts)
    bad, _ = F.run_case(verbose=3, backend=_backend, stride=api.PAIR_RES_CAP, seed=331, npairs=8000, rdlen=101, sub=0.03)
    assert bad == 0


def test_golden_pairs_sam(g1_index, golden_dir):
    import gzip
    import tempfile

    import numpy as np

    import h2o_py as H
    import pe_sink as PS
    _, s1 = H.read_fasta_reads(os.path.join(golden_dir, "reads_pe_1.fa.gz"))
    _, s2 = H.read_fasta_reads(os.path.join(golden_dir, "reads_pe_2.fa.gz"))
    with tempfile.NamedTemporaryFile("w", suffix=".sam", delete=False) as t:
        t.write(gzip.open(os.path.join(golden_dir, "ref_pe_nospliced.sam.gz"), "rt").read())
    refnames, want = F.parse_pe_sam(t.name)
    os.unlink(t.name)
    q = [str(i) for i in range(len(s1))]
    res, a1, a2 = _backend(g1_index, np.stack(s1), np.stack(s2), q, q)
    for i in range(len(s1)):
        assert res[i].overflow == 0
        assert PS.finish_pair(res[i], a1, a2, i * api.PAIR_RES_CAP, refnames, (s1[i], s2[i])) == want[q[i]], i


def test_pair_records_beyond_the_rows_are_all_returned(monkeypatch):
    """a mate with more reports than its fixed device rows: the pair's records live in the stream's overflow area (MachOut::ovf,
    PairOut::pad) and the dense fetch returns every one of them.  H2G_PAIR_SLOTS=1 pushes every multi-report pair through it."""
    import subprocess
    import tempfile

    import numpy as np
    build = os.path.join(ROOT, "oracle", "_ref", "hisat2-build-s")
    if not os.path.exists(build):
        pytest.skip("needs oracle/_ref")
    tmp = tempfile.mkdtemp(prefix="h2ovf")
    contigs = synth.make_genome([300000, 100000], 515, n_gaps=2, gap_len=300, repeats=60, repeat_len=500)
    synth.write_fasta(os.path.join(tmp, "g.fa"), contigs)
    base = os.path.join(tmp, "g")
    subprocess.run([build, "-q", os.path.join(tmp, "g.fa"), base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    m1, m2 = synth.make_pairs(contigs, 20000, 101, 516, frag_mean=300, frag_sd=30, sub_rate=0.01)
    q = [str(i) for i in range(len(m1))]

    def run():
        c1, o1 = synth.flatten_reads(m1)
        c2, o2 = synth.flatten_reads(m2)
        ix = api.Index(base, device=0)
        st = api.Stream(ix, max_reads=len(m1), max_bases=c1.size)
        st.set_reads(c1, o1)
        st.set_read_names(q)
        st.set_mates(c2, o2, q)
        p = st.align_params()
        p.no_spliced_alignment = 1
        st.align_pairs_run(p)
        res, a1, f1, a2, f2 = st.align_pairs_fetch_dense()
        n1, n2 = int(f1[len(m1)]), int(f2[len(m1)])
        # (the device-side block offset — PairOut::pad — is never shown to a caller: a pair lives in the area iff a mate has more records than its rows)
        slots = int(os.environ.get("H2G_PAIR_SLOTS", "0"))
        out = (bytes(res), bytes(a1)[:n1 * C.sizeof(api.AlnRes)], f1.copy(), bytes(a2)[:n2 * C.sizeof(api.AlnRes)], f2.copy(),
               [int(slots and (r.nres[0] > slots or r.nres[1] > slots)) for r in res], [r.pad for r in res])
        st.close()
        ix.close()
        return out

    import ctypes as C
    want = run()
    assert not any(want[5]) and not any(want[6])
    monkeypatch.setenv("H2G_PAIR_SLOTS", "1")
    got = run()
    assert not any(got[6])                                     # the returned headers carry no device-side offset
    npad = sum(1 for x in got[5] if x)
    print("pairs in the overflow area:", npad)
    assert npad > 100
    # everything but `pad` itself is identical: counts, flags, PRNG state, every record of both mates
    rw = np.frombuffer(want[0], dtype=np.uint32).reshape(len(m1), -1).copy()
    rg = np.frombuffer(got[0], dtype=np.uint32).reshape(len(m1), -1).copy()
    k = api.PairResult.pad.offset // 4
    rw[:, k] = 0
    rg[:, k] = 0
    assert (rw == rg).all()
    assert (want[2] == got[2]).all() and (want[4] == got[4]).all()
    assert want[1] == got[1] and want[3] == got[3]
