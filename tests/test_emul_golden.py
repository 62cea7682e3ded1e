"""CPU tests of the *device* item functions (hisat2_amd/csrc/h2g_core.h) instantiated on the host by
tests/emul — same golden vectors as the oracle, plus emul == oracle on fresh seeded inputs."""
import os

import numpy as np
import pytest

import h2o_py as H
import parity_cases as PC
from h2gemu_py import Emu
from hisat2_amd import synth


@pytest.fixture(scope="module")
def emu(g1_index, golden_dir):
    e = Emu(g1_index)
    reads, offs = PC.load_reads(golden_dir)
    e.set_reads(reads.reshape(-1), offs)
    return e


def test_rank(emu, golden_dir):
    PC.check_rank(emu.rank, golden_dir)


def test_fm_search(emu, golden_dir):
    assert PC.check_fm_search(emu, golden_dir, "probe_psearch.txt.gz", 0) == 800
    assert PC.check_fm_search(emu, golden_dir, "probe_psearch_spliced.txt.gz", 1) == 800


def test_coords(emu, golden_dir):
    assert PC.check_coords(emu, golden_dir) > 300


def test_extend(emu, golden_dir):
    assert PC.check_extend(emu, golden_dir) > 1000


def test_seed_stage_matches_oracle(oracle_lib, g1_index, golden_dir):
    # fresh reads (incl. Ns and indels) drawn from the golden genome
    contigs = PC.load_contigs(golden_dir)
    reads, _ = synth.make_reads(contigs, 600, 101, 77, sub_rate=0.02, indel_rate=0.001, n_rate=0.002)
    codes, offs = synth.flatten_reads(reads)
    e = Emu(g1_index)
    e.set_reads(codes, offs)
    oix = H.load_index(oracle_lib, g1_index)
    for pseudo in (0, 1):
        got = e.seed_extend(pseudogeneStop=pseudo)
        want = PC.oracle_seed_extend(oracle_lib, oix, reads, pseudo)
        PC.assert_seed_equal(got, want)
    assert (got["ncoords"] > 0).sum() > 300


# ---------------------------------------------------------------- graph index primitives (h2g_graph.h)
@pytest.fixture(scope="module")
def gemu(g1s_index, golden_dir):
    e = Emu(g1s_index)
    reads, offs = PC.load_snp_reads(golden_dir)
    e.set_reads(reads.reshape(-1), offs)
    return e


def test_graph_rank(gemu, golden_dir):
    assert PC.check_graph_rank(gemu.rank, golden_dir) == 3000


def test_graph_lf(gemu, golden_dir):
    assert PC.check_graph_lf(gemu, golden_dir) == 6000


def test_graph_fm_search(gemu, golden_dir):
    assert PC.check_graph_fm_search(gemu, golden_dir) == 600
    assert PC.check_graph_fm_search(gemu, golden_dir, "probe_g1s_psearch_spliced.txt.gz") == 600


def test_graph_genome_coords(gemu, golden_dir):
    n, multi = PC.check_graph_coords(gemu, golden_dir, "probe_g1s_coords.txt.gz")
    assert n > 250 and multi >= 4
    n, multi = PC.check_graph_coords(gemu, golden_dir, "probe_g1s_coords_short.txt.gz")
    assert n > 3000 and multi >= 100


def test_graph_extend_with_alts(gemu, golden_dir):
    assert PC.check_graph_extend(gemu, golden_dir) > 1000


def test_graph_adjust_with_alt(golden_dir, g1s_index):
    for fn, reads in (("probe_g1s_adjust.txt.gz", PC.load_snp_reads), ("probe_g1s_adjust_short.txt.gz", None)):
        e = Emu(g1s_index)
        if reads is None:
            _, seqs = H.read_fasta_reads(__import__("os").path.join(golden_dir, "reads_snp_short.fa.gz"))
            import numpy as np
            arr = np.stack(seqs)
            offs = (np.arange(len(seqs) + 1, dtype=np.uint64) * arr.shape[1]).astype(np.uint32)
        else:
            arr, offs = reads(golden_dir)
        e.set_reads(arr.reshape(-1), offs)
        assert PC.check_graph_adjust(e, golden_dir, fn) > 250


def test_local_graph_lf(gemu, golden_dir):
    import numpy as np
    rows = [l.split() for l in H.glines(golden_dir, "probe_g1s_lglf.txt.gz")]
    q = np.array([[int(v) for v in f[:6]] for f in rows], dtype=np.uint32)
    res, ie = gemu.local_graph_lf(q, k=10)
    for f, r, e in zip(rows, res, ie):
        exp = tuple(map(int, f[6:10]))
        if exp[:2] == (0, 0):
            assert not r.ok
        else:
            assert r.ok and (r.top, r.bot, r.node_top, r.node_bot) == exp and e.pairs() == [tuple(map(int, x.split(":"))) for x in f[11:]], f


def test_graph_lf_matches_oracle_on_random_ranges(gemu, oracle_lib, g1s_index):
    """fresh seeded ranges, incl. ranges that straddle sides and tiny ranges around multi-in-edge nodes"""
    import ctypes as C
    import numpy as np
    from hisat2_amd import api
    oix = H.load_index(oracle_lib, g1s_index)
    g = C.byref(oix.contents.g)
    glen = oix.contents.g.p.gbwtLen
    rng = np.random.default_rng(99)
    qs = []
    for _ in range(20000):
        top = int(rng.integers(0, glen - 2))
        spread = int(rng.integers(2, 7)) if rng.random() < 0.5 else int(rng.integers(2, 600))
        qs.append(api.GlfQuery(top, min(glen, top + spread), int(rng.integers(0, 4)), 0))
    qs = [q for q in qs if q.bot > q.top + 1]
    res, ie = gemu.graph_lf(qs, k=20)
    u32 = C.c_uint32
    for q, r, e in zip(qs, res, ie):
        a, b, na, nb, n = u32(), u32(), u32(), u32(), u32()
        buf = (u32 * 128)()
        ok = oracle_lib.h2o_map_glf(g, q.top, q.bot, q.c, 20, a, b, na, nb, buf, 64, n)
        assert bool(ok) == bool(r.ok)
        if ok:
            assert (r.top, r.bot, r.node_top, r.node_bot) == (a.value, b.value, na.value, nb.value)
            assert e.n == n.value and e.pairs() == [(buf[2 * i], buf[2 * i + 1]) for i in range(min(n.value, 24))]


# ---------------------------------------------------------------- Smith-Waterman (h2g_sw.h)
def test_sw_align(g1_index, golden_dir):
    e = Emu(g1_index)
    reads, offs = PC.load_sw_reads(golden_dir)
    e.set_reads(reads.reshape(-1), offs)
    assert PC.check_sw(e, golden_dir)[0] > 250


def test_sw_align_16bit_cells(g1_index, golden_dir):
    """--score-min below -254: SwAligner's 16-bit path (aligner_sw.cpp:496); both matrix layouts (even / odd problems)"""
    e = Emu(g1_index)
    reads, offs = PC.load_sw_reads(golden_dir, "reads_sw16.fa.gz")
    e.set_reads(reads.reshape(-1), offs)
    n, nbig = PC.check_sw(e, golden_dir, rdlen=150, fn="probe_sw16.txt.gz")
    assert n > 700 and nbig > 100


def test_ext_search_golden(emu, golden_dir):
    """globalGFMSearch / localGFMSearch (hi_aligner.h:6606 / :6751): the host instantiation against the real classes' vectors"""
    import ctypes as C
    from hisat2_amd import api
    emu.L.h2gemu_local_index_of.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    emu.L.h2gemu_local_index_of.restype = C.c_uint32
    emu.L.h2gemu_ext_search.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]

    def search(qs):
        n = len(qs)
        arr = (api.ExtSearchQuery * n)(*qs)
        out = (api.ExtSearchHit * n)()
        emu.L.h2gemu_ext_search(emu.h, arr, n, out)
        return out
    n, nel = PC.check_ext_search(search, lambda t, o: emu.L.h2gemu_local_index_of(emu.h, t, o), golden_dir, "probe_extsearch.txt.gz")
    assert n > 2000 and nel > 1500


def test_sw_reference_known_answer_cases(g1_index, golden_dir):
    """the reference's own SwAligner known-answer cases (aligner_sw.cpp:1470-2727, tests/golden/sw_kat.json) through the product's DP code (h2g_sw.h:
    fill, gather, backtrace), both cell layouts; the last case runs on 16-bit cells (minsc -260)"""
    import ctypes as C
    import numpy as np
    import sw_kat as K
    from hisat2_amd import api
    e = Emu(g1_index)
    cases = K.load(golden_dir)
    reads = [K.codes(c["read"]) for c in cases]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint32)
    e.set_reads(np.concatenate(reads), offs, quals="".join(c["qual"] for c in cases).encode())
    f = e.L.h2gemu_sw_align_window
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_uint32,
                  C.c_uint32, C.c_void_p]
    nfound = 0
    for i, case in enumerate(cases):
        s = case["scoring"]
        mx, mn = K.mm_range(s)
        sc = (C.c_int * 7)(mx, mn, s["npen"], s["rdGapConst"], s["rdGapLinear"], s["rfGapConst"], s["rfGapLinear"])
        ref = K.codes(case["ref"])
        refl, refr, corel, corer = K.window(case)
        for layout in (0, 1):
            o = api.SwResult()
            f(e.h, i, ref.ctypes.data, len(ref), refl, refr, corel, corer, sc, s["gapbar"], case["minsc"], 10 ** 6 if case["nceil"] is None else case["nceil"], layout, 1,
              C.byref(o))
            assert not o.overflow
            eds = [(o.edits[k].type, o.edits[k].pos) for k in range(o.nedits)]
            ns = sum(1 for k in range(o.nedits) if o.edits[k].type == 3 and (chr(o.edits[k].chr) == "N" or chr(o.edits[k].qchr) == "N"))
            K.check(case, dict(found=o.found, score=o.score, off=o.off, gaps=o.gaps, ns=ns, edits=eds))
        nfound += o.found
    assert len(cases) == 105 and nfound > 60


def test_sw_align_vs_oracle_random_both_widths(oracle_lib, g1_index, golden_dir):
    """fresh indel-rich reads (60-250 bases, Ns, hits at the contig ends, unrelated placements), minsc from the default down to -3 per base: the
    product's DP code against the C oracle, 8-bit and 16-bit cells, both matrix layouts (even / odd problems)"""
    import ctypes as C
    import numpy as np
    import h2o_py as H
    from hisat2_amd import api, synth
    contigs = PC.load_contigs(golden_dir)
    rng = np.random.default_rng(11)
    reads, truth = [], []
    for L in (60, 101, 150, 200, 250):
        for sub in (0.02, 0.12):
            r, t = synth.make_reads(contigs, 60, L, 3000 + L + int(sub * 100), sub_rate=sub, indel_rate=0.01, n_rate=0.003)
            reads += [x for x in r]
            truth += [tuple(int(v) for v in x) for x in t]
    codes = np.concatenate(reads).astype(np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint32)
    e = Emu(g1_index)
    e.set_reads(codes, offs)
    oix = H.load_index(oracle_lib, g1_index)
    sc = H.Scoring()
    oracle_lib.h2o_scoring_default(C.byref(sc))
    qs = []
    for i, (ci, pos, fw) in enumerate(truth):
        L = len(reads[i])
        minsc = int(-float(rng.choice([0.2, 1.0, 2.0, 3.0])) * L)
        off = pos + int(rng.integers(-3, 4))
        if i % 7 == 0:
            off = int(rng.integers(0, 30))
        if i % 11 == 0:
            off = len(contigs[ci]) - L - int(rng.integers(0, 25))
        if i % 13 == 0:
            off = int(rng.integers(1000, len(contigs[ci]) - 1000))      # an unrelated placement
        qs.append(api.SwQuery(i, fw, ci, max(off, 0), minsc, i * 977 + 1))
    out, _ = e.sw_align(qs)
    nfound = nwide = 0
    for q, o in zip(qs, out):
        seq = np.ascontiguousarray(reads[q.read] if q.fw else H.revcomp(reads[q.read]))
        rnd = C.c_uint32(q.rnd)
        w = H.SwResult()
        oracle_lib.h2o_sw_align(oix, C.byref(sc), seq.ctypes.data, None, len(seq), q.tidx, q.refoff, q.minsc, int(0.15 * len(seq)), 4, C.byref(rnd), C.byref(w))
        assert (o.refl, o.refr, o.found_align, o.best, o.found, o.rnd) == (w.refl, w.refr, w.found_align, w.best, w.found, rnd.value), (q.read, q.refoff, q.minsc)
        if w.found and (w.nedits > api.MAX_EDITS or w.overflow):
            assert o.overflow and (o.score, o.off) == (w.score, w.off)
        elif w.found:
            assert not o.overflow and (o.score, o.off, o.nedits) == (w.score, w.off, w.nedits)
            for k in range(w.nedits):
                assert (o.edits[k].pos, o.edits[k].chr, o.edits[k].qchr, o.edits[k].type) == (w.edits[k].pos, w.edits[k].chr, w.edits[k].qchr, w.edits[k].type)
        nfound += w.found
        nwide += q.minsc < -254
    assert nfound > 300 and nwide > 200


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_probe")), reason="needs oracle/_ref")
@pytest.mark.parametrize("case", [
    dict(seed=5, nreads=400, rdlen=101, sub=0.03, minsc=-303),                 # --score-min L,0,-3: every problem on 16-bit cells
    dict(seed=6, nreads=300, rdlen=150, sub=0.1, minsc=-450, shift=333),       # + unrelated placements
    dict(seed=7, nreads=300, rdlen=250, sub=0.05, minsc=-700),                 # four row chunks
    dict(seed=9, nreads=300, rdlen=101, sub=0.2, minsc=-254, shift=555),       # the last 8-bit minsc ...
    dict(seed=10, nreads=300, rdlen=101, sub=0.2, minsc=-255, shift=555),      # ... and the first 16-bit one
])
def test_sw_align_live_reference(golden_dir, case):
    """fresh reads through the reference's own SwAligner (oracle/_ref/ref_probe sw) and through h2g_sw.h, problem by problem"""
    import fuzz_sw as F
    n, bad = F.run_case(F.emu_backend, golden_dir, **case)
    assert n > 50 and bad == 0


@pytest.mark.parametrize("spliced", [False, True])
def test_combine_with_golden(g1_index, golden_dir, spliced):
    """GenomeHit::combineWith (hi_aligner.h:1420-2025) at unit level: 1 500 pairs of exact anchors with a plain join, mismatches, an insertion,
    a deletion or an intron between them (tests/gen_golden.py combine: oracle/ref_probe.cpp `combine` calls the reference's own class) -> hit_combine of the
    device sources on the host: the return value, the combined extent, the score and every edit (type, position, bases; intron length, direction, known flag)."""
    import ctypes as C
    from h2gemu_py import Emu
    from hisat2_amd import api
    names, seqs = H.read_fasta_reads(os.path.join(golden_dir, "reads_combine.fa.gz"))
    n = len(seqs)
    e = Emu(g1_index)
    codes = np.concatenate(seqs).astype(np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint32)
    e.set_reads(codes, offs)
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
    ok = np.zeros(n, dtype=np.uint32)
    e.L.h2gemu_combine.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    e.L.h2gemu_combine(e.h, 0 if spliced else 1, a, b, minsc.ctypes.data, n, ok.ctypes.data)
    kinds = set()
    ncomb = 0
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
