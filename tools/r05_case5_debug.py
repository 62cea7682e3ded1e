#!/usr/bin/env python3
"""Which pairs differ between the fast pass (k_go_fast_am + tail hand-off) and the general machine on the repeat-structured genome of test_gpu_fast_pass.py[case5],
and in which fields?  Runs the two settings in separate processes (the switches are read once), dumps every PairResult + dense records, diffs them here."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

PR_DT = np.dtype([("nres", "<u4", 2), ("npairs", "<u4"), ("overflow", "<u4"), ("nrank", "<u4"), ("nsteps", "<u4"), ("depth", "<u4"), ("nside", "<u4"), ("rnd", "<u4"), ("pad", "<u4"),
                  ("pi", "u1", 32), ("pj", "u1", 32)])
ALN_DT = np.dtype([("fw", "<u4"), ("tidx", "<u4"), ("toff", "<u4"), ("len", "<u4"), ("trim5", "<u4"), ("trim3", "<u4"), ("nedits", "<u4"), ("spl", "<u4"),
                   ("score", "<i8"), ("edits", [("pos", "<u4"), ("chr", "u1"), ("qchr", "u1"), ("type", "u1"), ("pad", "u1"), ("snp", "<u4")], 32)])

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from hisat2_amd import api, synth
    base, npz, out, runs = sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
    d = np.load(npz)
    m1, m2 = d["m1"], d["m2"]
    n = len(m1)
    c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
    names = [str(i) for i in range(n)]
    ix = api.Index(base, device=0)
    st = api.Stream(ix, max_reads=n, max_bases=c1.size + 64)
    st.set_reads(c1, o1); st.set_read_names(names); st.set_mates(c2, o2, names)
    p = st.align_params(); p.no_spliced_alignment = 1
    for _ in range(runs):
        st.align_pairs_run(p)
    res, a1, f1, a2, f2 = st.align_pairs_fetch_dense()
    np.savez(out, res=np.frombuffer(bytes(res), dtype=np.uint8), a1=np.frombuffer(bytes(a1), dtype=np.uint8)[:int(f1[n]) * ALN_DT.itemsize].copy(), f1=f1,
             a2=np.frombuffer(bytes(a2), dtype=np.uint8)[:int(f2[n]) * ALN_DT.itemsize].copy(), f2=f2)
    c = st.counters()
    print(json.dumps({"fast": int(c.n_fast), "handed_on": int(c.n_fast_bail), "second": int(c.n_second_pass), "aligned": int(c.n_aligned)}))
    sys.exit(0)

from hisat2_amd import synth
tmp = tempfile.mkdtemp(prefix="h2c5")
seed, npairs = 76, 100000
contigs = synth.make_repeat_genome([5000000, 2000000, 1000000], seed)
fa, base = os.path.join(tmp, "g.fa"), os.path.join(tmp, "g")
synth.write_fasta(fa, contigs)
subprocess.run([os.path.join(ROOT, "oracle", "_ref", "hisat2-build-s"), "-q", "-p", "16", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
m1, m2 = synth.make_pairs(contigs, npairs, 101, seed + 1, frag_mean=300, frag_sd=40, sub_rate=0.01)
npz = os.path.join(tmp, "r.npz")
np.savez(npz, m1=np.stack(m1), m2=np.stack(m2))
cfgs = {"machine": dict(H2G_GO_FAST="0"), "fast_am_tail_3runs": dict(H2G_GO_FAST="1", H2G_FAST_AM="1", H2G_FAST_TAIL="16"), "fast_am_tail_1run": dict(H2G_GO_FAST="1", H2G_FAST_AM="1", H2G_FAST_TAIL="16"),
        "fast_am_notail_3runs": dict(H2G_GO_FAST="1", H2G_FAST_AM="1", H2G_FAST_TAIL="0"), "fast_noam_tail_3runs": dict(H2G_GO_FAST="1", H2G_FAST_AM="0", H2G_FAST_TAIL="16"),
        "fast_am_tail_3runs_M2": dict(H2G_GO_FAST="1", H2G_FAST_AM="1", H2G_FAST_TAIL="16", H2G_MSTREAMS="2")}
outs = {}
for k, env in cfgs.items():
    o = os.path.join(tmp, k + ".npz")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", base, npz, o, "1" if k.endswith("1run") else "3"], env=dict(os.environ, **env), capture_output=True, text=True)
    print(k, r.stdout.strip()[-200:], r.stderr.strip()[-300:])
    outs[k] = np.load(o)
def masked(a8):
    a = np.frombuffer(a8.tobytes(), dtype=ALN_DT).copy()
    keep = np.arange(32)[None, :] < a["nedits"][:, None]
    for f in ("pos", "chr", "qchr", "type", "pad", "snp"):
        a["edits"][f][~keep] = 0
    return np.frombuffer(a.tobytes(), dtype=np.uint8).reshape(-1, ALN_DT.itemsize)


ref = outs["machine"]
R0 = np.frombuffer(ref["res"].tobytes(), dtype=PR_DT)
B0 = np.frombuffer(ref["res"].tobytes(), dtype=np.uint8).reshape(npairs, -1)
for k in cfgs:
    if k == "machine":
        continue
    x = outs[k]
    R1 = np.frombuffer(x["res"].tobytes(), dtype=PR_DT)
    B1 = np.frombuffer(x["res"].tobytes(), dtype=np.uint8).reshape(npairs, -1)
    badmask = np.any(B0 != B1, axis=1)
    for fa_, aa, fb_, ab in ((ref["f1"], ref["a1"], x["f1"], x["a1"]), (ref["f2"], ref["a2"], x["f2"], x["a2"])):
        if np.array_equal(fa_, fb_):
            d = np.any(masked(aa) != masked(ab), axis=1)
            recpair = np.searchsorted(fa_, np.nonzero(d)[0], side="right") - 1
            badmask[recpair] = True
        else:
            badmask |= np.diff(fa_.astype(np.int64)) != np.diff(fb_.astype(np.int64))
    bad = np.nonzero(badmask)[0].tolist()
    print(k, "differing pairs:", len(bad), bad[:12])
    for i in bad[:6]:
        a, b = R0[i], R1[i]
        print("  pair", i, "machine", {f: a[f].tolist() for f in ("nres", "npairs", "overflow", "nrank", "nsteps", "depth", "nside", "rnd", "pad")}, "pairs", a["pi"][:a["npairs"]].tolist(), a["pj"][:a["npairs"]].tolist())
        print("       ", k, {f: b[f].tolist() for f in ("nres", "npairs", "overflow", "nrank", "nsteps", "depth", "nside", "rnd", "pad")}, "pairs", b["pi"][:b["npairs"]].tolist(), b["pj"][:b["npairs"]].tolist())
        for m, (fa_, aa, fb_, ab) in enumerate(((ref["f1"], ref["a1"], x["f1"], x["a1"]), (ref["f2"], ref["a2"], x["f2"], x["a2"]))):
            ra = np.frombuffer(aa.tobytes()[int(fa_[i]) * ALN_DT.itemsize:int(fa_[i + 1]) * ALN_DT.itemsize], dtype=ALN_DT)
            rb = np.frombuffer(ab.tobytes()[int(fb_[i]) * ALN_DT.itemsize:int(fb_[i + 1]) * ALN_DT.itemsize], dtype=ALN_DT)
            print("      mate", m, "machine", [(int(u["fw"]), int(u["tidx"]), int(u["toff"]), int(u["len"]), int(u["trim5"]), int(u["trim3"]), int(u["nedits"]), int(u["score"])) for u in ra])
            print("      mate", m, "fast   ", [(int(u["fw"]), int(u["tidx"]), int(u["toff"]), int(u["len"]), int(u["trim5"]), int(u["trim3"]), int(u["nedits"]), int(u["score"])) for u in rb])
