"""The fast path of go() (hisat2_amd/csrc/h2g_fast.h) against the general machine, both instantiated on the host (tests/emul): every read
/ pair runs through both; whatever the fast path completes must equal the machine's result bit for bit (PairOut / ReadOut incl. the PRNG
state and the work counters, every record).  The machine itself is pinned to the reference binary by test_go_parity_cpu.py."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from hisat2_amd import synth
import fast_check as FC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "oracle", "_ref", "hisat2-build-s")
pytestmark = pytest.mark.skipif(not os.path.exists(BUILD), reason="needs oracle/_ref/hisat2-build-s")


@pytest.fixture(scope="module")
def genome():
    tmp = tempfile.mkdtemp(prefix="h2fast")
    contigs = synth.make_genome([400000, 150000, 60000], 9101, n_gaps=3, gap_len=300, repeats=40, repeat_len=500)
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs)
    base = os.path.join(tmp, "g")
    subprocess.run([BUILD, "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return base, contigs


@pytest.mark.parametrize("case", [
    dict(n=6000, rdlen=101, sub=0.005),
    dict(n=4000, rdlen=101, sub=0.03, least=0.25),      # 3 mismatches per read on average: every other read needs a 4th edit
    dict(n=3000, rdlen=75, sub=0.01, frag_mean=400, frag_sd=200),
    dict(n=3000, rdlen=125, sub=0.02),
    dict(n=2000, rdlen=40, sub=0.01, frag_mean=200, frag_sd=40),
])
def test_pairs_equal_the_machine(genome, case):
    base, contigs = genome
    m1, m2 = synth.make_pairs(contigs, case["n"], case["rdlen"], 77 + case["n"], frag_mean=case.get("frag_mean", 300), frag_sd=case.get("frag_sd", 30), sub_rate=case["sub"])
    r = FC.fast_check(base, list(m1), list(m2))
    assert r["mismatching"] == 0, r
    assert r["completed"] > case.get("least", 0.5) * r["n"], r            # the fast path is the path: it must take most pairs even on a repeat-rich genome


@pytest.mark.parametrize("case", [
    dict(n=8000, rdlen=101, sub=0.005),
    dict(n=4000, rdlen=101, sub=0.02, indel=0.002),
    dict(n=3000, rdlen=60, sub=0.02, nrate=0.01),
    dict(n=3000, rdlen=128, sub=0.01),
    dict(n=1000, rdlen=150, sub=0.01),                  # longer than the packed form: everything is handed on
])
def test_reads_equal_the_machine(genome, case):
    base, contigs = genome
    reads, _ = synth.make_reads(contigs, case["n"], case["rdlen"], 5 + case["n"], sub_rate=case["sub"], indel_rate=case.get("indel", 0.0), n_rate=case.get("nrate", 0.0))
    r = FC.fast_check(base, list(reads))
    assert r["mismatching"] == 0, r
    if case["rdlen"] > 128:
        assert r["completed"] == 0 and r["bails"].get("input") == case["n"], r
    else:
        assert r["completed"] > 0.4 * r["n"], r


def test_qualities_and_scoring_options(genome):
    base, contigs = genome
    n = 4000
    m1, m2 = synth.make_pairs(contigs, n, 101, 4242, sub_rate=0.02)
    rng = np.random.default_rng(7)
    quals = (33 + rng.integers(2, 41, size=n * 101)).astype(np.uint8)       # FASTQ: quality-aware mismatch penalties and PRNG seeds
    r = FC.fast_check(base, list(m1), None, quals=quals)
    assert r["mismatching"] == 0, r
    for opts in (("--mp", "4,2", "--score-min", "L,0,-0.4"), ("-k", "2"), ("--no-softclip",), ("--sp", "3,1", "--np", "2")):
        r = FC.fast_check(base, list(m1), list(m2), options=opts)
        assert r["mismatching"] == 0, (opts, r)
        assert r["completed"] > 0.4 * n, (opts, r)


def test_pair_records_beyond_the_rows_go_to_the_overflow_area(genome):
    """MachOut::ovf: a pair with more reports than a mate's fixed rows keeps all of them in a block of the area (PairOut::pad); only a
    full area flags the pair (overflow bit 4) -- the sink's lists grow on demand (aln_sink.h:2565)"""
    base, contigs = genome
    m1, m2 = synth.make_pairs(contigs, 3000, 101, 4242, frag_mean=300, frag_sd=30, sub_rate=0.01)
    r = FC.pairs_overflow_check(base, list(m1), list(m2), 1, 1 << 16)
    assert r["mismatching"] == 0 and r["flagged"] == 0 and r["in_area"] > 20, r
    small = FC.pairs_overflow_check(base, list(m1), list(m2), 1, 40)
    assert small["mismatching"] == 0 and small["flagged"] > 0 and small["in_area"] + small["flagged"] == r["in_area"], (small, r)


def test_chunk_accessors_equal_the_base_accessors(genome):
    """SeqView::chunk32 / RefCursor::chunk32 (32 bases per step in extend / combineWith) against SeqView::at / RefCursor::get: every view
    position of reads of several lengths on both strands, and seeded reference windows incl. those next to N stretches and contig ends"""
    import ctypes as C
    from h2gemu_py import Emu
    base, contigs = genome
    e = Emu(base)
    e.L.h2gemu_chunk_check.restype = C.c_uint64
    e.L.h2gemu_chunk_check.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
    for rdlen in (33, 64, 101, 127, 128):
        reads, _ = synth.make_reads(contigs, 300, rdlen, 900 + rdlen, sub_rate=0.02)
        codes = np.concatenate([np.concatenate(list(reads)).astype(np.uint8), np.zeros(8, np.uint8)])
        offs = (np.arange(len(reads) + 1, dtype=np.uint64) * rdlen).astype(np.uint32)
        e.set_reads(codes, offs, None)
        assert e.L.h2gemu_chunk_check(e.h, 20000, rdlen) == 0


def test_mate_rescue_in_the_fast_path(genome):
    """alignMate (hi_aligner.h:5579) restated in the fast path (FG_ALIGN_MATE; compiled into libh2gemu_am.so, not into the shipped kernel):
    pairs whose second mate carries too many mismatches for an end-to-end alignment are rescued through the local index next to the first
    mate — the result, incl. the work counters and the wrap of the live minimum score for a mate without an alignment, equals the machine's"""
    base, contigs = genome
    n = 4000
    m1, m2 = synth.make_pairs(contigs, n, 101, 5151, frag_mean=300, frag_sd=30, sub_rate=0.004)
    rng = np.random.default_rng(11)
    m2 = np.array(m2, dtype=np.uint8).copy()
    for i in range(0, n, 2):                                      # every other pair: 5..8 extra mismatches in mate 2
        pos = rng.choice(101, size=int(rng.integers(5, 9)), replace=False)
        m2[i, pos] = (m2[i, pos] + rng.integers(1, 4, size=len(pos))) & 3
    r = FC.fast_check(base, list(m1), list(m2), variant="am")
    assert r["mismatching"] == 0, r
    assert "mate" not in r["bails"], r
    assert r["completed"] > 0.8 * n, r
    # the shipped configuration hands exactly these pairs on (FB_MATE) and completes the rest equal to the machine
    s = FC.fast_check(base, list(m1), list(m2))
    assert s["mismatching"] == 0 and s["bails"].get("mate", 0) > 0.3 * n, s


@pytest.fixture(scope="module")
def graph_genome():
    """a SNP-graph index (single-base variants, deletions, insertions about every 200 bp) and the alternate haplotype that carries all of them"""
    tmp = tempfile.mkdtemp(prefix="h2fastg")
    contigs = synth.make_genome([400000, 150000, 60000], 9102, n_gaps=3, gap_len=300, repeats=20, repeat_len=500)
    var = synth.make_snps(contigs, 78, every=200)
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs)
    synth.write_snps(os.path.join(tmp, "g.snp"), var)
    base = os.path.join(tmp, "g")
    subprocess.run([BUILD, "-q", "--snp", os.path.join(tmp, "g.snp"), fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return base, contigs, synth.apply_snps(contigs, var)


@pytest.mark.parametrize("case", [
    dict(n=6000, rdlen=101, sub=0.005, alt=True),
    dict(n=3000, rdlen=101, sub=0.005, alt=False),             # reads from the reference haplotype: the graph paths without ALT edits
    dict(n=3000, rdlen=101, sub=0.03, alt=True, least=0.2),
    dict(n=2000, rdlen=60, sub=0.01, alt=True, frag_mean=200, frag_sd=40),
    dict(n=2000, rdlen=125, sub=0.01, alt=True),
])
def test_graph_pairs_equal_the_machine(graph_genome, case):
    """the fast path over a graph index (FG_GRAPH = 1, tests/emul/libh2gemu_g.so = the configuration of h2g_k_go_fast_graph.hip): node ranges and
    in-edge lists of the partial hits, adjustWithALT of every coordinate, ALT-aware extension and joins, ALT ids in the reported edits"""
    base, ref, alt = graph_genome
    m1, m2 = synth.make_pairs(alt if case["alt"] else ref, case["n"], case["rdlen"], 177 + case["n"], frag_mean=case.get("frag_mean", 300), frag_sd=case.get("frag_sd", 30), sub_rate=case["sub"])
    r = FC.fast_check(base, list(m1), list(m2), variant="g")
    assert r["mismatching"] == 0, r
    assert r["completed"] > case.get("least", 0.5) * r["n"], r


@pytest.mark.parametrize("case", [
    dict(n=6000, rdlen=101, sub=0.005),
    dict(n=3000, rdlen=101, sub=0.02, indel=0.002),
    dict(n=2000, rdlen=50, sub=0.01),
])
def test_graph_reads_equal_the_machine(graph_genome, case):
    base, ref, alt = graph_genome
    reads, _ = synth.make_reads(alt, case["n"], case["rdlen"], 15 + case["n"], sub_rate=case["sub"], indel_rate=case.get("indel", 0.0))
    r = FC.fast_check(base, list(reads), variant="g")
    assert r["mismatching"] == 0, r
    assert r["completed"] > 0.4 * r["n"], r


def test_the_graph_library_refuses_a_linear_index(genome):
    base, contigs = genome
    reads, _ = synth.make_reads(contigs, 10, 101, 5)
    r = FC.fast_check(base, list(reads), variant="g")
    assert r["mismatching"] > 10 ** 9          # (the library says so instead of running the wrong fast path)


def test_fused_graph_lf_and_single_row_walk(graph_genome):
    """glf1_top_fused (one LF step of one row with the M / F bits and side headers taken from sides held in registers) == map_glf1_nochar,
    and gw_walk_single (a one-node, one-row walk in chunks) == gw_resolve's offset and step count, on seeded rows of the global graph
    index and of its graph local indexes"""
    import ctypes as C
    from h2gemu_py import Emu
    base, ref, alt = graph_genome
    e = Emu(base, "g")
    e.L.h2gemu_glf_fused_check.restype = C.c_uint64
    e.L.h2gemu_glf_fused_check.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
    assert e.L.h2gemu_glf_fused_check(e.h, 100000, 11) == 0


def test_local_indexes_packed_straight_from_the_files(genome, graph_genome):
    """load_local_pack (h2g_local_pack.h: one walk over the local headers, then threaded copies of the sides and the 16-bit words — what
    h2g_index_load uses) == pack_local over the per-index host objects, byte for byte, on a linear and on a graph index"""
    import ctypes as C
    from h2gemu_py import Emu
    for base in (genome[0], graph_genome[0]):
        e = Emu(base)
        e.L.h2gemu_local_pack_check.argtypes = [C.c_void_p, C.c_char_p]
        assert e.L.h2gemu_local_pack_check(e.h, base.encode()) == 0


def test_index_arrays_straight_from_the_mapped_files(genome, graph_genome):
    """the device loader's host side (round 6): any byte range of the packed local-index arrays gathered out of the files (plan_local_pack + local_fill_*) == load_local_pack's
    arrays, and the global index's sides / SA sample / reference bases as views into the mapped files (BigViews) == the copies a plain load makes — linear and graph index"""
    import ctypes as C
    from h2gemu_py import Emu
    for base in (genome[0], graph_genome[0]):
        e = Emu(base)
        e.L.h2gemu_local_fill_check.argtypes = [C.c_void_p, C.c_char_p]
        assert e.L.h2gemu_local_fill_check(e.h, base.encode()) == 0


def test_staged_graph_lf_step_equals_the_fused_one(g1s_index):
    """h2g_graph_staged.h (the LF step of one row cut at its dependent loads, four rows in flight stage by stage: measurement kernel k_glf_chain and the
    groundwork of several rows per lane) against glf1_top_fused: 50 000 x 4 walks of 12 steps on the global index, 50 000 rows of local indexes; the searches' step with a required character in stages
    against map_glf1_fused (every character, '$' rows included); four coordinate walks advanced together (gw_walk_multi) against gw_walk_single"""
    import ctypes as C
    from h2gemu_py import Emu
    e = Emu(g1s_index)
    e.L.h2gemu_glf_staged_check.restype = C.c_uint64
    e.L.h2gemu_glf_staged_check.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    assert e.L.h2gemu_glf_staged_check(e.h, 50000, 12, 5) == 0


def test_bail_reason_names_cover_the_fb_enum():
    """h2gemu_fast_check writes 2 + FB_COUNT words and bench.py reads FB_COUNT counters: both name lists must be the FB_* enum of h2g_fast.h,
    entry for entry (a reason added to the enum alone would overflow the stats buffer / drop out of the report)"""
    import re
    import fast_check as FC
    import bench
    src = open(os.path.join(ROOT, "hisat2_amd", "csrc", "h2g_fast.h")).read()
    body = src[src.index("FB_NONE = 0"):]
    body = body[:body.index("FB_COUNT")]
    names = [m.lower() for m in re.findall(r"FB_([A-Z]+)", body)]
    assert names == FC.BAIL_REASONS
    assert names == bench.BAIL_REASONS
