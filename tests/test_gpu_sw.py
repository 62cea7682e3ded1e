"""-m gpu: the batched Smith-Waterman kernels k_sw_fill / k_sw_backtrace (h2g_sw_align: SwAligner::initRef + the 8-bit end-to-end
fill + gatherCells + the first nextAlignment with its backtrace and PRNG reseeding, aligner_sw.cpp:137-851,
aligner_swsse_ee_u8.cpp:791-1900) against vectors of the real reference classes and against the C oracle."""
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
    assert PC.check_sw(st, golden_dir) > 250
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
        minsc = int(-0.2 * L)
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
        if w.found:
            assert (o.score, o.off, o.nedits) == (w.score, w.off, w.nedits)
            for k in range(w.nedits):
                assert (o.edits[k].pos, o.edits[k].chr, o.edits[k].qchr, o.edits[k].type) == (w.edits[k].pos, w.edits[k].chr, w.edits[k].qchr, w.edits[k].type)
            nfound += 1
    assert nfound > 400 and ms > 0
    st.close()
    ix.close()


def test_sw_refuses_scores_beyond_the_8bit_fill(g1_index, golden_dir):
    """SwAligner::align runs its 16-bit DP when minsc < -254 (aligner_sw.cpp:494-504); only the 8-bit fill is built, so go() with
    --bowtie2-dp must refuse such a --score-min instead of silently saturating"""
    contigs = PC.load_contigs(golden_dir)
    reads, _ = synth.make_reads(contigs, 50, 200, 77, sub_rate=0.01)
    codes, offs = synth.flatten_reads(reads)
    ix = api.Index(g1_index, device=0)
    st = api.Stream(ix, max_reads=len(reads), max_bases=codes.size)
    st.set_reads(codes, offs)
    st.set_read_names([str(i) for i in range(len(reads))])
    p = st.align_params()
    p.apply_options(["--bowtie2-dp", "2", "--score-min", "L,0,-2"])     # -400 for 200 bp
    with pytest.raises(api.H2GError):
        st.align_run(p)
    p2 = st.align_params()
    p2.apply_options(["--bowtie2-dp", "2", "--score-min", "L,0,-1.2"])  # -240: fine
    st.align_run(p2)
    st.sync()
    st.close()
    ix.close()
