"""-m gpu: the batched Smith-Waterman kernels k_sw_fill / k_sw_backtrace (h2g_sw_align: SwAligner::initRef + the end-to-end fill, 8-bit
or — minsc below -254 — 16-bit cells, + gatherCells + the first nextAlignment with its backtrace and PRNG reseeding, aligner_sw.cpp:137-906,
aligner_swsse_ee_u8.cpp:791-1900, aligner_swsse_ee_i16.cpp) against vectors of the real reference classes and against the C oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import h2o_py as H
import parity_cases as PC
from hisat2_amd import api, synth

pytestmark = pytest.mark.gpu


def test_sw_align_golden(g1_index, golden_dir):
    ix = api.Index(g1_index, device=0)
    reads, offs = PC.load_sw_reads(golden_dir)
    st = api.Stream(ix, max_reads=len(offs) - 1, max_bases=reads.size)
    st.set_reads(reads.reshape(-1), offs)
    assert PC.check_sw(st, golden_dir)[0] > 250
    st.close()
    ix.close()


def test_sw_align_golden_16bit_cells(g1_index, golden_dir):
    """probe_sw16: the reference's 16-bit path (scores down to -407, placements whose every candidate fails: the walk outgrows the mask
    table and is run again over the mask matrix); mixed with 8-bit problems in one launch"""
    ix = api.Index(g1_index, device=0)
    reads, offs = PC.load_sw_reads(golden_dir, "reads_sw16.fa.gz")
    st = api.Stream(ix, max_reads=len(offs) - 1, max_bases=reads.size)
    st.set_reads(reads.reshape(-1), offs)
    n, nbig = PC.check_sw(st, golden_dir, rdlen=150, fn="probe_sw16.txt.gz")
    assert n > 700 and nbig > 100
    # the same problems with an 8-bit minsc in between (cell width is per problem)
    cases = PC.parse_sw_probe(golden_dir, "probe_sw16.txt.gz")[:200]
    qs = []
    for k, d in enumerate(cases):
        qs.append(api.SwQuery(d["rid"], d["fw"], d["tidx"], d["refoff"], d["minsc"] if k % 2 == 0 else -200, (d["rid"] * 7 + d["k"] + 1) & 0xFFFFFFFF))
    out, _ = st.sw_align(qs)
    for k, (d, o) in enumerate(zip(cases, out)):
        if k % 2 == 0:
            assert (o.found_align, o.best, o.found) == (d["found_align"], d["best"], d["found"]), d
        elif d["best"] >= -200:
            assert o.best == d["best"], d                     # the best cell does not depend on the cell width while it fits
    st.close()
    ix.close()


def test_sw_align_vs_oracle_random(oracle_lib, g1_index, golden_dir):
    """fresh indel-rich reads incl. Ns, ragged lengths (60..250: one, two, three and four 64-row chunks of the wave kernel)
    and hits at the contig ends (trimmed rectangles)"""
    contigs = PC.load_contigs(golden_dir)
    rng = np.random.default_rng(4)
    reads, truth = [], []
    for L in (60, 101, 150, 200, 250):
        r, t = synth.make_reads(contigs, 300, L, 1000 + L, sub_rate=0.02, indel_rate=0.01, n_rate=0.002)
        reads += [x for x in r]
        truth += [tuple(int(v) for v in x) for x in t]
    codes = np.concatenate(reads).astype(np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint32)
    ix = api.Index(g1_index, device=0)
    st = api.Stream(ix, max_reads=len(reads), max_bases=codes.size)
    st.set_reads(codes, offs)
    oix = H.load_index(oracle_lib, g1_index)
    sc = H.Scoring()
    oracle_lib.h2o_scoring_default(C.byref(sc))
    qs = []
    for i, (ci, pos, fw) in enumerate(truth):
        L = len(reads[i])
        minsc = int(-0.2 * L) if i % 3 else int(-2.8 * L)     # every third problem below -254: 16-bit cells
        off = pos + int(rng.integers(-3, 4))
        if i % 7 == 0:
            off = int(rng.integers(0, 30))                    # left end of the contig: triml > 0
        if i % 11 == 0:
            off = len(contigs[ci]) - L - int(rng.integers(0, 25))   # right end: trimr > 0
        qs.append(api.SwQuery(i, fw, ci, max(off, 0), minsc, i * 977 + 1))
    out, ms = st.sw_align(qs)
    nfound = 0
    for q, o in zip(qs, out):
        seq = np.ascontiguousarray(reads[q.read] if q.fw else H.revcomp(reads[q.read]))
        rnd = C.c_uint32(q.rnd)
        w = H.SwResult()
        oracle_lib.h2o_sw_align(oix, C.byref(sc), seq.ctypes.data, None, len(seq), q.tidx, q.refoff, q.minsc, int(0.15 * len(seq)), 4,
                                C.byref(rnd), C.byref(w))
        assert (o.refl, o.refr, o.found_align, o.best, o.found, o.rnd) == (w.refl, w.refr, w.found_align, w.best, w.found, rnd.value), (q.read, q.refoff)
        if w.found and (w.nedits > api.MAX_EDITS or w.overflow):
            assert o.overflow and (o.score, o.off) == (w.score, w.off)       # more edits than a record holds: flagged
        elif w.found:
            assert (o.score, o.off, o.nedits) == (w.score, w.off, w.nedits)
            for k in range(w.nedits):
                assert (o.edits[k].pos, o.edits[k].chr, o.edits[k].qchr, o.edits[k].type) == (w.edits[k].pos, w.edits[k].chr, w.edits[k].qchr, w.edits[k].type)
            nfound += 1
    assert nfound > 400 and ms > 0
    st.close()
    ix.close()


def test_go_runs_the_16bit_dp_below_minus_254(g1_index, golden_dir):
    """--bowtie2-dp with a --score-min below -254 (SwAligner::align's 16-bit path, aligner_sw.cpp:496) inside go(): every read against the
    live reference (the emulator runs the same cases in tests/test_go_parity_cpu.py)"""
    import functools
    import fuzz_align as F
    from test_gpu_align import _backend
    if not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "hisat2-align-s")):
        pytest.skip("needs oracle/_ref")
    bad, _ = F.run_case(verbose=3, backend=functools.partial(_backend, bowtie2_dp=2), seed=1604, nreads=4000, rdlen=101, sub=0.02, indel=0.008, nrate=0.001,
                        extra=("--bowtie2-dp", "2", "--score-min", "L,0,-3"))
    assert bad == 0
