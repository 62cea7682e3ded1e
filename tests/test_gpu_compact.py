"""The compact result fetch (h2g_align_fetch_compact / h2g_align_pairs_fetch_compact: sized, scanned and gathered on the device; a record travels as its 40 bytes of
fields + 12 per edit held) against the dense fetch of the same resident batch: byte for byte the host-compacted dense records, the same offsets, the same headers —
and the compact SAM formatter's text equals the dense formatter's.  Pairs on the golden genome + its SNP graph, single reads, page-locked buffers."""
import ctypes as C
import os

import numpy as np
import pytest

import h2o_py as H
import sam_lines as SL
from hisat2_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("graph,slots", [(False, 0), (True, 0), (False, 1)])       # slots 1: one-record rows push every multi-record mate through the overflow area
def test_pairs_compact_equals_dense(g1_index, g1s_index, golden_dir, graph, slots):
    base = g1s_index if graph else g1_index
    _, s1 = H.read_fasta_reads(os.path.join(golden_dir, "reads_pe_1.fa.gz"))
    _, s2 = H.read_fasta_reads(os.path.join(golden_dir, "reads_pe_2.fa.gz"))
    n = len(s1)
    m1, m2 = np.stack(s1), np.stack(s2)
    c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
    q = [str(i) for i in range(n)]
    ix = api.Index(base, device=0)
    st = api.Stream(ix, max_reads=n, max_bases=c1.size)
    st.set_reads(c1, o1); st.set_read_names(q); st.set_mates(c2, o2, q)
    p = st.align_params()
    if slots:
        st.tune("pair_slots", slots)
    st.align_pairs_run(p)
    res, a1, f1, a2, f2 = st.align_pairs_fetch_dense()
    pool = api.PinnedPool()
    cres, r1, b1, r2, b2 = st.align_pairs_fetch_compact(pinned=pool)
    assert bytes(res) == cres.tobytes()                                    # the same headers (the device-side block offset is not a caller's business: 0)
    for a, f, r, b in ((a1, f1, r1, b1), (a2, f2, r2, b2)):
        want, woffs = SL.to_compact(a, f[:-1], np.diff(f.astype(np.int64)))
        assert np.array_equal(woffs, b)
        assert r[:int(b[n])].tobytes() == want[:int(woffs[n])]
    assert int(f1[n]) > n // 2                                             # most mates aligned: there is something to compare
    L = SL.load_sam_lib()
    dense_text = SL.format_paired(L, base, list(m1), list(m2), q, q, res, a1, a2, int(p.khits), dense=(f1, f2))   # (also runs the compact formatter on a host-compacted copy)
    h = C.c_void_p()
    assert L.h2g_sam_open(base.encode(), C.byref(h)) == 0
    nb, no = SL.flat_names(q)
    L.h2g_sam_format_paired_compact.argtypes = [C.c_void_p] + [C.c_void_p] * 10 + [C.c_size_t] + [C.c_void_p] * 5 + [C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    cap = 1 << 24
    buf = C.create_string_buffer(cap)
    used = C.c_size_t(0)
    rc = L.h2g_sam_format_paired_compact(h, c1.ctypes.data, o1.ctypes.data, None, nb, no.ctypes.data, c2.ctypes.data, o2.ctypes.data, None, nb, no.ctypes.data, n,
                                         cres.ctypes.data, r1.ctypes.data, b1.ctypes.data, r2.ctypes.data, b2.ctypes.data, int(p.khits), buf, cap, C.byref(used))
    L.h2g_sam_close(h)
    assert rc == 0 and buf.raw[:used.value].decode().splitlines() == dense_text
    pool.close(); st.close(); ix.close()


def test_unpaired_compact_equals_dense(g1_index, golden_dir):
    names, seqs = H.read_fasta_reads(os.path.join(golden_dir, "reads_se.fa.gz"))
    n = len(seqs)
    reads = np.stack(seqs)
    codes, offs = synth.flatten_reads(reads)
    ix = api.Index(g1_index, device=0)
    st = api.Stream(ix, max_reads=n, max_bases=codes.size)
    st.set_reads(codes, offs); st.set_read_names(names)
    st.align_run()
    res, aln, f = st.align_fetch_dense()
    cres, rec, b = st.align_fetch_compact()
    assert res.tobytes() == cres.tobytes()
    want, woffs = SL.to_compact(aln, f[:-1], np.diff(f.astype(np.int64)))
    assert np.array_equal(woffs, b) and rec[:int(b[n])].tobytes() == want[:int(woffs[n])]
    assert int(f[n]) > n // 2
    st.close(); ix.close()


def test_compact_offsets_over_many_scan_tiles(g1_index, golden_dir):
    """30 011 pairs: the prefix sum behind the compact fetch runs over 15 tiles of 2048 entries (tile scan, scan of the tile sums, add back) — offsets and bytes
    against the host compaction of the dense fetch"""
    import parity_cases as PC
    contigs = PC.load_contigs(golden_dir)
    n = 30011
    m1, m2 = synth.make_pairs(contigs, n, 101, 4711, frag_mean=300, frag_sd=30, sub_rate=0.02)
    c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
    q = [str(i) for i in range(n)]
    ix = api.Index(g1_index, device=0)
    st = api.Stream(ix, max_reads=n, max_bases=c1.size)
    st.set_reads(c1, o1); st.set_read_names(q); st.set_mates(c2, o2, q)
    st.align_pairs_run()
    res, a1, f1, a2, f2 = st.align_pairs_fetch_dense()
    cres, r1, b1, r2, b2 = st.align_pairs_fetch_compact()
    assert bytes(res) == cres.tobytes()
    for a, f, r, b in ((a1, f1, r1, b1), (a2, f2, r2, b2)):
        want, woffs = SL.to_compact(a, f[:-1], np.diff(f.astype(np.int64)))
        assert np.array_equal(woffs, b)
        assert r[:int(b[n])].tobytes() == want[:int(woffs[n])]
    assert int(f1[n]) > n // 2
    st.close(); ix.close()
