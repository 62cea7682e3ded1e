"""Resident batches (h2g_stream_select_batch): several read sets with result rows of their own on ONE stream, runs over them queued back to back (machine passes of one batch
next to the fast passes of the others).  Every batch's fetched results must equal what a stream that only ever held that batch returns."""
import os

import numpy as np
import pytest

import h2o_py as H
from hisat2_amd import api, synth

pytestmark = pytest.mark.gpu


def _one(ix, m1, m2, names, fetch):
    c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
    st = api.Stream(ix, max_reads=len(m1), max_bases=c1.size)
    st.set_reads(c1, o1); st.set_read_names(names); st.set_mates(c2, o2, names)
    st.align_pairs_run()
    out = fetch(st)
    st.close()
    return out


ALN_DT = np.dtype([("fw", "<u4"), ("tidx", "<u4"), ("toff", "<u4"), ("len", "<u4"), ("trim5", "<u4"), ("trim3", "<u4"), ("nedits", "<u4"), ("spl", "<u4"),
                   ("score", "<i8"), ("edits", [("pos", "<u4"), ("chr", "u1"), ("qchr", "u1"), ("type", "u1"), ("pad", "u1"), ("snp", "<u4")], 32)])


def _recs(arr, n):
    """n dense records with the edit entries a record does not use zeroed (they are whatever the row held before)"""
    a = np.frombuffer(arr, dtype=ALN_DT, count=n).copy()
    keep = np.arange(32)[None, :] < a["nedits"][:, None]
    for f in ("pos", "chr", "qchr", "type", "pad", "snp"):
        a["edits"][f][~keep] = 0
    return a.tobytes()


def _dense(st):
    res, a1, f1, a2, f2 = st.align_pairs_fetch_dense()
    return bytes(res), _recs(a1, int(f1[-1])), f1.tobytes(), _recs(a2, int(f2[-1])), f2.tobytes()


@pytest.mark.parametrize("graph", [False, True])
def test_batches_in_flight_keep_their_rows(g1_index, g1s_index, golden_dir, graph):
    base = g1s_index if graph else g1_index
    _, s1 = H.read_fasta_reads(os.path.join(golden_dir, "reads_pe_1.fa.gz"))
    _, s2 = H.read_fasta_reads(os.path.join(golden_dir, "reads_pe_2.fa.gz"))
    n = len(s1)
    ix = api.Index(base, device=0)
    # three different batches out of the golden pairs: a rotation, a reversed order, every other pair twice
    orders = [np.arange(n), np.arange(n)[::-1].copy(), np.repeat(np.arange(0, n, 2), 2)[:n]]
    sets = []
    for k, o in enumerate(orders):
        m1 = np.stack([s1[i] for i in o]); m2 = np.stack([s2[i] for i in o])
        names = ["b%d_%d" % (k, i) for i in range(len(o))]          # (names feed genRandSeed: each batch has its own)
        sets.append((m1, m2, names))
    want = [_one(ix, m1, m2, names, _dense) for m1, m2, names in sets]
    assert want[0] != want[1]
    cmax = max(m1.size for m1, _, _ in sets)
    st = api.Stream(ix, max_reads=n, max_bases=cmax + 64)
    for k, (m1, m2, names) in enumerate(sets):
        st.select_batch(k)
        c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
        st.set_reads(c1, o1); st.set_read_names(names); st.set_mates(c2, o2, names)
    for rep in range(4):                                             # 12 runs queued without a sync: batches alternate, up to 8 machine passes in flight
        for k in range(len(sets)):
            st.select_batch(k)
            st.align_pairs_run()
    st.sync()
    for k in (2, 0, 1):
        st.select_batch(k)
        assert _dense(st) == want[k], "batch %d" % k
    st.close(); ix.close()


def test_a_small_batch_behind_a_large_one(g1_index, golden_dir):
    """Two resident batches of different sizes queued alternately without a sync: the large one ends in a drain launch (the policy of >= 200 000 pairs), the small one does not —
    its fast launch uses slot pool 0, which the large batch's drain launch may still be reading (go_run: every fast launch waits for the last reader of its pool)."""
    _, s1 = H.read_fasta_reads(os.path.join(golden_dir, "reads_pe_1.fa.gz"))
    _, s2 = H.read_fasta_reads(os.path.join(golden_dir, "reads_pe_2.fa.gz"))
    n = len(s1)
    ix = api.Index(g1_index, device=0)
    big_n, small_n = 220_000, 3 * n
    sets = []
    for k, cnt in enumerate((big_n, small_n)):
        o = (np.arange(cnt) * (7 + 4 * k)) % n
        m1 = np.stack([s1[i] for i in o]); m2 = np.stack([s2[i] for i in o])
        sets.append((m1, m2, ["m%d_%d" % (k, i) for i in range(cnt)]))
    want = [_one(ix, m1, m2, names, _dense) for m1, m2, names in sets]
    st = api.Stream(ix, max_reads=big_n, max_bases=sets[0][0].size + 64)
    for k, (m1, m2, names) in enumerate(sets):
        st.select_batch(k)
        c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
        st.set_reads(c1, o1); st.set_read_names(names); st.set_mates(c2, o2, names)
    for rep in range(6):
        for k in (0, 1):
            st.select_batch(k)
            st.align_pairs_run()
    st.sync()
    st.select_batch(0)
    assert int(st.counters().n_adopted) >= 0
    for k in (1, 0):
        st.select_batch(k)
        assert _dense(st) == want[k], "batch %d" % k
    st.close(); ix.close()
