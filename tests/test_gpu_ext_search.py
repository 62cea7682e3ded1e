"""-m gpu: h2g_ext_search — globalGFMSearch / localGFMSearch (hi_aligner.h:6606 / :6751) as hybridSearch_recur calls them, with the
local queries bucketed by local index and the fat buckets served from LDS (the index's sides + ftab staged once per workgroup).
Staged, unstaged and the host instantiation of the same item function must agree on every field; the LDS hit-rate is reported."""
import ctypes as C
import os

import numpy as np
import pytest

import parity_cases as PC
from h2gemu_py import Emu
from hisat2_amd import api, synth

pytestmark = pytest.mark.gpu
FIELDS = ("nelt", "hitlen", "top", "bot", "uniqueStop", "nrank", "nside")


def _queries(ix, contigs, reads, truth, rng):
    L = api.lib()
    qs = []
    for i, (ci, pos, fw) in enumerate(truth):
        n = len(reads[i])
        for _ in range(3):
            rdoff = int(rng.integers(8, n))
            kind = int(rng.integers(0, 10))
            if kind == 0:
                lidx = 0xffffffff                                   # global search
            else:
                # the local index under the read, sometimes its neighbour (as the recursion walks to prev / next)
                lidx = L.h2g_local_index_of(ix.h, ci, max(0, pos + int(rng.integers(-60000, 60000)) * (kind == 1)))
                if lidx == 0xffffffff:
                    lidx = L.h2g_local_index_of(ix.h, ci, pos)
            q = api.ExtSearchQuery()
            q.read, q.rdoff, q.lidx = i, rdoff, lidx
            q.maxHitLen = 0xffff if kind % 2 else int(rng.integers(8, 40))
            q.fw, q.uniqueStop = int(fw), int(rng.integers(0, 2))
            qs.append(q)
    return qs


@pytest.mark.parametrize("which", ["g1", "g1s"])
def test_ext_search_staged_equals_unstaged_equals_host(which, g1_index, g1s_index, golden_dir):
    base = g1_index if which == "g1" else g1s_index
    contigs = PC.load_contigs(golden_dir)
    rng = np.random.default_rng(11)
    reads, truth = synth.make_reads(contigs, 4000, 101, 555, sub_rate=0.01)
    truth = [tuple(int(v) for v in t) for t in truth]
    codes, offs = synth.flatten_reads(reads)
    ix = api.Index(base, device=0)
    st = api.Stream(ix, max_reads=len(reads), max_bases=codes.size)
    st.set_reads(codes, offs)
    qs = _queries(ix, contigs, reads, truth, rng)
    n = len(qs)
    arr = (api.ExtSearchQuery * n)(*qs)
    staged, s1 = st.ext_search(arr, stage_min=2)
    plain, s0 = st.ext_search(arr, stage_min=0)
    assert s0.n_staged == 0 and s0.n_local == s1.n_local > 0
    for a, b in zip(staged, plain):
        assert all(getattr(a, f) == getattr(b, f) for f in FIELDS)
    linear = bool(ix.info.linear)
    if linear:
        assert s1.n_staged > 0.9 * s1.n_local and all(h.staged == (q.lidx != 0xffffffff) for h, q in zip(staged, qs) if True) or s1.n_buckets_staged < s1.n_buckets
        e = Emu(base)
        e.set_reads(codes, offs)
        want = (api.ExtSearchHit * n)()
        e.L.h2gemu_ext_search.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        e.L.h2gemu_ext_search(e.h, arr, n, want)
        for a, b in zip(staged, want):
            assert all(getattr(a, f) == getattr(b, f) for f in FIELDS)
    else:
        assert s1.n_staged == 0          # graph local indexes (128 B sides with F/M bits) are searched from HBM
    found = sum(1 for h in staged if h.nelt > 0)
    assert found > n // 3
    print(f"{which}: {n} queries, {s1.n_local} local in {s1.n_buckets} buckets, staged {s1.n_staged} ({100.0 * s1.n_staged / max(1, s1.n_local):.1f} % LDS hit-rate) "
          f"in {s1.n_buckets_staged} workgroup buckets ({s1.lds_bytes_staged} B staged), staged kernel {s1.ms_staged:.3f} ms, HBM kernel {s1.ms_hbm:.3f} ms; "
          f"unstaged all-HBM {s0.ms_hbm:.3f} ms")
    st.close()
    ix.close()


@pytest.mark.parametrize("which", ["g1", "g1s"])
def test_ext_search_equals_the_reference_classes(which, g1_index, g1s_index, golden_dir):
    """staged and unstaged h2g_ext_search against vectors of the REAL HI_Aligner::globalGFMSearch / localGFMSearch (oracle/ref_probe.cpp
    extsearch, tests/gen_golden.py extsearch): linear index and SNP-graph index (graph local indexes incl. the linear ones inside it)"""
    import h2o_py as H
    base, reads_fn, vec = (g1_index, "reads_se.fa.gz", "probe_extsearch.txt.gz") if which == "g1" else (g1s_index, "reads_snp.fa.gz", "probe_g1s_extsearch.txt.gz")
    _, seqs = H.read_fasta_reads(os.path.join(golden_dir, reads_fn))
    codes = np.concatenate(seqs).astype(np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(x) for x in seqs])]).astype(np.uint32)
    ix = api.Index(base, device=0)
    st = api.Stream(ix, max_reads=len(seqs), max_bases=codes.size)
    st.set_reads(codes, offs)
    L = api.lib()
    for stage_min in (2, 0):
        n, nel = PC.check_ext_search(lambda qs: st.ext_search((api.ExtSearchQuery * len(qs))(*qs), stage_min=stage_min)[0],
                                     lambda t, o: L.h2g_local_index_of(ix.h, t, o), golden_dir, vec)
        assert n > 1500 and nel > 1200
