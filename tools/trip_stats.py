#!/usr/bin/env python3
"""Development: how long are the latency chains of the reads the fast pass hands on?  Host instantiation (tests/emul): the batch through
the fast path, then the general machine over every handed-on pair with its primitive requests (trips) counted per pair.
usage: trip_stats.py [pairs] [genome bases] [rep|rnd]"""
import ctypes as C, os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import bench
from hisat2_amd import synth
import build_bench_index as BB
import fast_check as FC
from h2gemu_py import Emu

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
glen = int(float(sys.argv[2])) if len(sys.argv) > 2 else 40_000_000
kind = sys.argv[3] if len(sys.argv) > 3 else "rep"
OPS = "none PSEARCH GCOORDS EXTEND LSEARCH LCOORDS GSEARCH COMBINE ADJUST ADJMEMBER SW FINISH".split()
if kind == "rep":
    contigs = synth.make_repeat_genome(BB.contig_lens(glen), bench.SEED + 77)
    rdir = os.path.join("/tmp", "fast_perf_rep%d" % glen)
    base = os.path.join(rdir, "g")
    if not os.path.exists(base + ".8.ht2"):
        os.makedirs(rdir, exist_ok=True)
        synth.write_fasta(base + ".fa", contigs)
        subprocess.run([os.path.join(bench.REF, "hisat2-build-s"), "-q", "-p", "8", base + ".fa", base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    m1, m2 = synth.make_pairs(contigs, n, 101, bench.SEED + 78, sub_rate=0.005)
else:
    base, contigs = bench.build_index(os.path.join(ROOT, ".bench_cache"), glen)
    m1, m2 = synth.make_pairs(contigs, n, 101, bench.SEED + 7, sub_rate=0.005)
t0 = time.time()
r = FC.fast_check(base, m1, m2)
print("fast_check %.1f s: completed %d mismatching %d bails %s" % (time.time() - t0, r["completed"], r["mismatching"], r["bails"]))
ids = np.nonzero(r["done"] == 0)[0].astype(np.uint32)
e = Emu(base)
codes = np.concatenate(m1).astype(np.uint8)
offs = np.concatenate([[0], np.cumsum([len(x) for x in m1])]).astype(np.uint32)
e.set_reads(codes, offs, None)
names = [str(i) for i in range(n)]
nb = "".join(names).encode()
noffs = np.concatenate([[0], np.cumsum([len(q) for q in names])]).astype(np.uint32)
c2 = np.concatenate([np.concatenate(m2).astype(np.uint8), np.zeros(8, np.uint8)])
o2 = np.concatenate([[0], np.cumsum([len(x) for x in m2])]).astype(np.uint32)
out = np.zeros((len(ids), 16), dtype=np.uint32)
vp = C.c_void_p
e.L.h2gemu_pair_trips.argtypes = [vp, vp, vp, C.c_char_p, vp, C.c_char_p, vp, vp, C.c_size_t, vp]
e.L.h2gemu_pair_trips(e.h, c2.ctypes.data, o2.ctypes.data, nb, noffs.ctypes.data, nb, noffs.ctypes.data, ids.ctypes.data, len(ids), out.ctypes.data)
trips = out[:, 1:12].sum(axis=1)
print("handed on: %d of %d pairs; trips per handed-on pair: mean %.1f median %d p90 %d p99 %d max %d" % (len(ids), n, trips.mean(), np.median(trips), np.percentile(trips, 90), np.percentile(trips, 99), trips.max()))
print("by primitive (mean per handed-on pair): " + ", ".join("%s %.1f" % (OPS[k], out[:, k].mean()) for k in range(1, 12) if out[:, k].any()))
order = np.argsort(-trips)[:10]
for j in order:
    print("  pair %d: %d trips: " % (ids[j], trips[j]) + ", ".join("%s %d" % (OPS[k], out[j, k]) for k in range(1, 12) if out[j, k]))
hist = np.bincount(np.minimum(trips // 50, 40))
print("histogram (bins of 50 trips):", hist.tolist())
# all pairs (incl. completed ones) for scale
allids = np.arange(min(n, 20000), dtype=np.uint32)
out2 = np.zeros((len(allids), 16), dtype=np.uint32)
e.L.h2gemu_pair_trips(e.h, c2.ctypes.data, o2.ctypes.data, nb, noffs.ctypes.data, nb, noffs.ctypes.data, allids.ctypes.data, len(allids), out2.ctypes.data)
t2 = out2[:, 1:12].sum(axis=1)
print("every pair (first %d): machine trips mean %.1f median %d p99 %d max %d" % (len(allids), t2.mean(), np.median(t2), np.percentile(t2, 99), t2.max()))
