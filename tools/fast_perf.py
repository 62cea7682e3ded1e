#!/usr/bin/env python3
"""go() timing with the fast pass on / off (env H2G_GO_FAST, read once per process) + a checksum of every result, so that the two
settings can be compared for identical output.  usage: fast_perf.py se|pe|gpe|rpe [n] [genome bases]   (gpe: pairs from the alternate haplotype
on the SNP-graph index of bench.py's graph leg, a variant every ~250 bp)"""
import os, sys, zlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import bench
from hisat2_amd import api, synth

ALN_DT = np.dtype([("fw", "<u4"), ("tidx", "<u4"), ("toff", "<u4"), ("len", "<u4"), ("trim5", "<u4"), ("trim3", "<u4"), ("nedits", "<u4"), ("spl", "<u4"),
                   ("score", "<i8"), ("edits", [("pos", "<u4"), ("chr", "u1"), ("qchr", "u1"), ("type", "u1"), ("pad", "u1"), ("snp", "<u4")], 32)])


def aln_crc(arr, n):
    a = np.frombuffer(arr, dtype=ALN_DT, count=n).copy()
    keep = np.arange(32)[None, :] < a["nedits"][:, None]
    for f in ("pos", "chr", "qchr", "type", "pad", "snp"):
        a["edits"][f][~keep] = 0
    return zlib.crc32(a.tobytes())


mode = sys.argv[1] if len(sys.argv) > 1 else "pe"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
glen = int(float(sys.argv[3])) if len(sys.argv) > 3 else 4_900_000
t0 = time.time()
if mode == "rpe":
    base, contigs = None, None
elif glen < 10_000_000:
    base, contigs = bench.build_index(os.path.join(ROOT, ".bench_cache"), glen)
else:
    import build_bench_index as BB
    base, total, how = bench.headline_index(os.path.join(ROOT, ".bench_cache"), glen)
    contigs = (lambda: BB.genome(total))        # generated only when the reads are not cached
if mode == "rpe":     # pairs on a repeat-structured genome (synth.make_repeat_genome), linear index built here
    import subprocess, tempfile
    import build_bench_index as BB
    glen = max(glen, 20_000_000)
    contigs = synth.make_repeat_genome(BB.contig_lens(glen), bench.SEED + 77)
    rdir = os.path.join("/tmp", "fast_perf_rep%d" % glen)
    base = os.path.join(rdir, "g")
    if not os.path.exists(base + ".8.ht2"):
        os.makedirs(rdir, exist_ok=True)
        synth.write_fasta(base + ".fa", contigs)
        subprocess.run([os.path.join(bench.REF, "hisat2-build-s"), "-q", "-p", "16", base + ".fa", base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
if mode == "gpe":
    import subprocess
    gtmp = os.path.join(ROOT, ".bench_cache", f"rnd{glen}_s{bench.SEED}_snp")
    gbase = os.path.join(gtmp, "g")
    var = synth.make_snps(contigs, bench.SEED + 5, every=250, names=["ecoli_substitute"])
    if not os.path.exists(gbase + ".8.ht2"):
        os.makedirs(gtmp, exist_ok=True)
        synth.write_fasta(gbase + ".fa", contigs, names=["ecoli_substitute"]); synth.write_snps(gbase + ".snp", var)
        subprocess.run([os.path.join(bench.REF, "hisat2-build-s"), "-q", "-p", "16", "--snp", gbase + ".snp", gbase + ".fa", gbase], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    contigs = synth.apply_snps(contigs, var, names=["ecoli_substitute"])
    base = gbase
print("index ready in %.1f s" % (time.time() - t0), flush=True)
ix = api.Index(base)
if mode == "se":
    if callable(contigs): contigs = contigs()
    reads, _ = synth.make_reads(contigs, n, 101, bench.SEED + 1000, sub_rate=0.005)
    codes, offs = synth.flatten_reads(reads)
    st = api.Stream(ix, max_reads=n, max_bases=codes.size)
    st.set_reads(codes, offs); st.set_read_names([str(i) for i in range(n)])
    run = st.align_run
else:
    pc = os.path.join("/tmp", "fast_perf_pairs_%s_%d_%d.npz" % (mode, n, glen))       # (several variant libraries are timed on the same reads: one process each)
    if os.path.exists(pc):
        d_ = np.load(pc); m1, m2 = d_["m1"], d_["m2"]
    else:
        if callable(contigs): contigs = contigs()
        m1, m2 = synth.make_pairs(contigs, n, 101, bench.SEED + (4343 if mode == "gpe" else 7), frag_mean=300, frag_sd=30, sub_rate=0.005) if mode == "gpe" else synth.make_pairs(contigs, n, 101, bench.SEED + 7, sub_rate=0.005)
        np.savez(pc, m1=np.stack(m1), m2=np.stack(m2))
    c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
    names = [str(i) for i in range(n)]
    st = api.Stream(ix, max_reads=n, max_bases=c1.size)
    st.set_reads(c1, o1); st.set_read_names(names); st.set_mates(c2, o2, names)
    run = st.align_pairs_run
ms = []
for _ in range(4):
    run(); st.sync()
    c = st.counters()
    ms.append((c.ms_align, c.ms_fast_kernel, c.ms_align_kernel))
# steady state: runs queued back to back (the machine pass of run k next to the fast pass of run k + 1), one sync at the end
K = int(os.environ.get("H2G_STEADY", "10"))
run(); st.sync()
t1 = time.perf_counter()
for _ in range(K):
    run()
st.sync()
steady = (time.perf_counter() - t1) * 1e3 / K
c = st.counters()
if mode == "se":
    res, aln, offs_ = st.align_fetch_dense()
    ck = zlib.crc32(res.tobytes()) ^ aln_crc(aln, int(offs_[n]))
else:
    res, a1, o1_, a2, o2_ = st.align_pairs_fetch_dense()
    ck = zlib.crc32(bytes(res)) ^ aln_crc(a1, int(o1_[n])) ^ aln_crc(a2, int(o2_[n]))
if os.environ.get("H2G_DUMP"):
    np.save(os.environ["H2G_DUMP"], np.frombuffer(bytes(res) if mode != "se" else res.tobytes(), dtype=np.uint8))
L = api.lib()
if hasattr(L, "h2g_go_fast_prof"):
    import ctypes as C
    v = (C.c_ulonglong * 136)()
    L.h2g_go_fast_prof.argtypes = [C.c_void_p, C.c_void_p]
    if L.h2g_go_fast_prof(st.h, v) == 0:
        reasons = "none input longpool subsample coords nghits edits depth localhits gsearch nres searched redundant mate npairs partial straddle other indel tail iedges gwalk".split()
        print("  bails:", {reasons[k]: int(v[48 + k]) for k in range(len(reasons)) if v[48 + k]})
        if v[47]:
            ops = "NONE PSEARCH GCOORDS EXTEND LSEARCH LCOORDS COMBINE GSEARCH ADJUST ADJMEMBER".split()
            tot = sum(v[k] for k in range(0, 18))
            print("  trips %d, slots per trip %.1f, wave-ticks %d; slots LOADED from their slot per slot-trip %.2f (the others stayed in their lane), - %.2f" % (v[47], v[46] / max(1, v[47]), tot, v[44] / max(1, v[46]), v[45] / max(1, v[47])))
            for k, nm in ((0, "pop+load"), (1, "control"), (2, "store"), (16, "hand-on list"), (17, "release fence + push"), (15, "new reads")):
                print("  %-20s %5.1f %%" % (nm, 100.0 * v[k] / tot))
            sites = ["FETCH", "P", "G", "E:HS", "E", "l", "c", "C", "g", "A", "a", "E:slow", "walk:slow"]
            print("  control by site (us per trip, trips, %% of wave time):", "  ".join("%s %.1f/%d/%.1f%%" % (sites[k] if k < len(sites) else k, v[72 + k] / max(1, v[104 + k]) / 2400.0, v[104 + k], 100.0 * v[72 + k] / tot) for k in range(32) if v[104 + k]))
            for op in range(1, 10):
                if v[3 + op]:
                    print("  %-20s %5.1f %%   executions %d: avg %.1f of 64 lanes" % (ops[op], 100.0 * v[3 + op] / tot, v[32 + op], v[20 + op] / max(1, v[32 + op])))
if hasattr(L, "h2g_go_fast_prof_bins"):
    import ctypes as C
    vb = (C.c_ulonglong * 256)()
    L.h2g_go_fast_prof_bins.argtypes = [C.c_void_p, C.c_void_p]
    if L.h2g_go_fast_prof_bins(st.h, vb) == 0:
        shift = 18 if mode == "gpe" else 15
        print("  per %.2f ms since the wave started: trips / lanes per trip / slots in flight per workgroup" % ((1 << shift) / 1e5))
        print("   ", "  ".join("%d/%.0f/%.0f" % (vb[4 * b], vb[4 * b + 1] / max(1, vb[4 * b]), vb[4 * b + 2] / max(1, vb[4 * b])) for b in range(64) if vb[4 * b]))
print("  drain launch: %.2f ms, adopted %d, sides %d steps %d" % (c.ms_drain_kernel, c.n_adopted, c.n_drain_side, c.n_drain_sa_steps))
print("%s n %d genome %d FAST=%s: steady %.2f ms/run | align total/fast/machine ms %s | fast done %d bailed %d (%.1f %%) second %d overflow %d aligned %d | sides/unit %.1f steps/unit %.1f | crc %08x" % (
    mode, n, glen, os.environ.get("H2G_GO_FAST", "1"), steady, " ".join("%.2f/%.2f/%.2f" % m for m in ms), c.n_fast, c.n_fast_bail, 100.0 * c.n_fast_bail / n,
    c.n_second_pass, c.n_overflow, c.n_aligned, c.n_side / n, c.n_sa_steps / n, ck))
if os.environ.get("H2G_JSONL"):
    import json
    with open(os.environ["H2G_JSONL"], "a") as f_:
        f_.write(json.dumps({"tag": os.environ.get("H2G_TAG", ""), "mode": mode, "n": n, "genome": glen, "steady_ms": steady, "runs_ms": ms, "fast_done": int(c.n_fast), "handed_on": int(c.n_fast_bail),
                             "drain_ms": float(c.ms_drain_kernel), "adopted": int(c.n_adopted), "aligned": int(c.n_aligned), "crc": "%08x" % ck}) + "\n")
