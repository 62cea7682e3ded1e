"""Queued-run stress of go() on the device: the fast pass (+ the machine passes it leaves in flight on the machine streams) against the
general machine alone, per read.  One process: the pass is switched with the development hook h2g_stream_tune("fast", 0 / 1).

The reference result of every read / pair comes from ONE machine-only run of the whole batch (a read's result does not depend on its
neighbours: the PRNG is seeded per read).  Then, with the fast pass on, `runs` runs are queued back to back without a fetch between them
(machine passes of runs k - 1 and k - 2 in flight next to the fast pass of run k), over prefixes of the batch of varying size, and what
the fetch returns is compared read by read: result struct (PairOut::pad, the block offset inside the overflow area, apart), every record.

usage: fast_stress.py index_base reads.npz [runs=6] [sizes=all,0.63,0.2,999]   -> one JSON line; exit 1 when anything differs"""
import ctypes as C
import json
import sys

import numpy as np

from hisat2_amd import api, synth

ALN_DT = np.dtype([("fw", "<u4"), ("tidx", "<u4"), ("toff", "<u4"), ("len", "<u4"), ("trim5", "<u4"), ("trim3", "<u4"), ("nedits", "<u4"), ("spl", "<u4"),
                   ("score", "<i8"), ("edits", [("pos", "<u4"), ("chr", "u1"), ("qchr", "u1"), ("type", "u1"), ("pad", "u1"), ("snp", "<u4")], 32)])


def tune(st, key, v):
    f = api.lib().h2g_stream_tune
    f.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
    rc = f(st.h, key.encode(), v)
    assert rc == 0, (key, v, rc)


def records(arr, n):
    """the first n records with the edit slots past nedits zeroed (they are not part of a record)"""
    a = np.frombuffer(arr, dtype=ALN_DT, count=n).copy()
    keep = np.arange(32)[None, :] < a["nedits"][:, None]
    for f in ("pos", "chr", "qchr", "type", "pad", "snp"):
        a["edits"][f][~keep] = 0
    return a


def per_read(recs, offs, n):
    b = recs.view(np.uint8).reshape(len(recs), -1) if len(recs) else np.zeros((0, ALN_DT.itemsize), np.uint8)
    return [b[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n)]


def fetch_pairs(st, n):
    res, a1, f1, a2, f2 = st.align_pairs_fetch_dense(0, n)
    r = np.frombuffer(bytes(res), dtype=np.uint8).reshape(n, -1).copy()
    off_pad = api.PairResult.pad.offset
    r[:, off_pad:off_pad + 4] = 0
    return r, per_read(records(a1, int(f1[n])), f1, n), per_read(records(a2, int(f2[n])), f2, n)


def fetch_reads(st, n):
    res, aln, offs = st.align_fetch_dense(0, n)
    r = res.view(np.uint8).reshape(n, -1).copy()
    return r, per_read(records(aln, int(offs[n])), offs, n)


def differing(ref, got, n):
    bad = []
    for i in range(n):
        if not np.array_equal(ref[0][i], got[0][i]) or any(ref[k][i] != got[k][i] for k in range(1, len(ref))):
            bad.append(i)
    return bad


def describe(ref, got, i):
    """what differs for read i: the struct, the records, or only the records' order"""
    d = {"id": i, "struct": not np.array_equal(ref[0][i], got[0][i])}
    if d["struct"]:
        w = np.flatnonzero(ref[0][i] != got[0][i])
        d["struct_bytes"] = [int(x) for x in w[:12]]
        d["struct_ref"] = [int(x) for x in ref[0][i].view(np.uint32)[:12]]; d["struct_got"] = [int(x) for x in got[0][i].view(np.uint32)[:12]]
    for k in range(1, len(ref)):
        a, b = ref[k][i], got[k][i]
        if a != b and len(a) == len(b):
            ra, rb = np.frombuffer(a, dtype=ALN_DT), np.frombuffer(b, dtype=ALN_DT)
            for q in range(len(ra)):
                if ra[q].tobytes() != rb[q].tobytes():
                    d["rec%d_%d" % (k, q)] = {f: (ra[q][f].tolist(), rb[q][f].tolist()) for f in ("fw", "tidx", "toff", "len", "trim5", "trim3", "nedits", "score") if ra[q][f] != rb[q][f]}
                    ne = int(max(ra[q]["nedits"], rb[q]["nedits"]))
                    d["rec%d_%d_edits" % (k, q)] = ([tuple(int(x) if not isinstance(x, bytes) else x for x in e.tolist()) for e in ra[q]["edits"][:ne]], [tuple(e.tolist()) for e in rb[q]["edits"][:ne]])
                    break
    for k in range(1, len(ref)):
        a, b = ref[k][i], got[k][i]
        if a != b:
            sz = ALN_DT.itemsize
            ra = sorted(a[j:j + sz] for j in range(0, len(a), sz))
            rb = sorted(b[j:j + sz] for j in range(0, len(b), sz))
            d["mate%d" % k] = "same records, another order" if ra == rb else "records differ (%d vs %d)" % (len(a) // sz, len(b) // sz)
    return d


def run(base, npz, runs=6, sizes=("all", 0.63, 0.2, 999), log=None):
    d = np.load(npz)
    ix = api.Index(base, device=0)
    m1, m2, rd = d["m1"], d["m2"], d["reads"]
    npairs, nreads = len(m1), len(rd)
    c1, o1 = synth.flatten_reads(m1)
    c2, o2 = synth.flatten_reads(m2)
    rc, ro = synth.flatten_reads(rd)
    st = api.Stream(ix, max_reads=max(npairs, nreads), max_bases=max(c1.size, c2.size, int(rc.size)) + 64)
    p = st.align_params(); p.no_spliced_alignment = 1
    out = {"runs": runs, "cases": [], "differing": 0}

    def sizes_of(n):
        return [n if s == "all" else (int(n * s) if isinstance(s, float) else min(int(s), n)) for s in sizes]

    def load(kind, n):
        if kind == "pairs":
            st.set_reads(c1[:int(o1[n])], o1[:n + 1]); st.set_read_names([str(i) for i in range(n)]); st.set_mates(c2[:int(o2[n])], o2[:n + 1], [str(i) for i in range(n)])
        else:
            st.set_reads(rc[:int(ro[n])], ro[:n + 1]); st.set_read_names([str(i) for i in range(n)])

    for kind, ntot in (("pairs", npairs), ("reads", nreads)):
        load(kind, ntot)
        tune(st, "fast", 0)
        (st.align_pairs_run if kind == "pairs" else st.align_run)(p)
        ref = (fetch_pairs if kind == "pairs" else fetch_reads)(st, ntot)
        tune(st, "fast", 1)
        for n in sizes_of(ntot):
            load(kind, n)
            for fast in (1, 0):
                tune(st, "fast", fast)
                for _ in range(runs):
                    (st.align_pairs_run if kind == "pairs" else st.align_run)(p)
                got = (fetch_pairs if kind == "pairs" else fetch_reads)(st, n)
                c = st.counters()
                bad = differing(ref, got, n)
                case = {"kind": kind, "n": n, "fast": fast, "handed_on": int(c.n_fast_bail), "differing": len(bad), "first": [describe(ref, got, i) for i in bad[:4]]}
                out["cases"].append(case)
                out["differing"] += len(bad)
                if log:
                    log(case)
    st.close(); ix.close()
    return out


if __name__ == "__main__":
    runs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    sizes = tuple(("all" if x == "all" else (float(x) if "." in x else int(x))) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else ("all", 0.63, 0.2, 999)
    r = run(sys.argv[1], sys.argv[2], runs, sizes)
    print(json.dumps(r))
    sys.exit(1 if r["differing"] else 0)
