"""Digest of every result of go() over a batch, printed as one JSON line (tests/test_gpu_fast_pass.py runs it twice: the fast pass on / off,
H2G_GO_FAST is read once per process).  usage: fast_digest.py index_base reads.npz"""
import hashlib
import json
import sys

import numpy as np

from hisat2_amd import api, synth

ALN_DT = np.dtype([("fw", "<u4"), ("tidx", "<u4"), ("toff", "<u4"), ("len", "<u4"), ("trim5", "<u4"), ("trim3", "<u4"), ("nedits", "<u4"), ("spl", "<u4"),
                   ("score", "<i8"), ("edits", [("pos", "<u4"), ("chr", "u1"), ("qchr", "u1"), ("type", "u1"), ("pad", "u1"), ("snp", "<u4")], 32)])


def aln_bytes(arr, n):
    a = np.frombuffer(arr, dtype=ALN_DT, count=n).copy()
    keep = np.arange(32)[None, :] < a["nedits"][:, None]
    for f in ("pos", "chr", "qchr", "type", "pad", "snp"):      # edit slots past nedits are not part of a record
        a["edits"][f][~keep] = 0
    return a.tobytes()


def main():
    base, npz = sys.argv[1], sys.argv[2]
    d = np.load(npz)
    out = {}
    ix = api.Index(base, device=0)
    m1, m2, rd = d["m1"], d["m2"], d["reads"]
    n = len(m1)
    c1, o1 = synth.flatten_reads(m1)
    c2, o2 = synth.flatten_reads(m2)
    names = [str(i) for i in range(n)]
    st = api.Stream(ix, max_reads=max(n, len(rd)), max_bases=max(c1.size, int(rd.size)) + 64)
    st.set_reads(c1, o1); st.set_read_names(names); st.set_mates(c2, o2, names)
    p = st.align_params(); p.no_spliced_alignment = 1
    for rep in range(3):                                         # back-to-back runs: the machine passes of earlier runs overlap the later fast passes
        st.align_pairs_run(p)
    res, a1, f1, a2, f2 = st.align_pairs_fetch_dense()
    h = hashlib.sha256()
    h.update(bytes(res)); h.update(f1.tobytes()); h.update(f2.tobytes()); h.update(aln_bytes(a1, int(f1[n]))); h.update(aln_bytes(a2, int(f2[n])))
    c = st.counters()
    out["pairs"] = {"sha": h.hexdigest(), "fast": int(c.n_fast), "handed_on": int(c.n_fast_bail), "aligned": int(c.n_aligned), "overflow": int(c.n_overflow), "adopted": int(c.n_adopted)}
    rc, ro = synth.flatten_reads(rd)
    st.set_reads(rc, ro); st.set_read_names([str(i) for i in range(len(rd))])
    for rep in range(3):
        st.align_run(p)
    res, aln, offs = st.align_fetch_dense()
    h = hashlib.sha256()
    h.update(res.tobytes()); h.update(offs.tobytes()); h.update(aln_bytes(aln, int(offs[len(rd)])))
    c = st.counters()
    out["reads"] = {"sha": h.hexdigest(), "fast": int(c.n_fast), "handed_on": int(c.n_fast_bail), "aligned": int(c.n_aligned), "overflow": int(c.n_overflow), "adopted": int(c.n_adopted)}
    st.close(); ix.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
