#!/usr/bin/env python3
"""Wave-level time split of the go() kernel (library built with -DH2G_GO_PROF): refill / control / vote / each primitive,
requesters per executed primitive.  usage: go_prof.py [nreads] [paired 0|1] [index base]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from hisat2_amd import api, synth
nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
paired = len(sys.argv) > 2 and sys.argv[2] == "1"
if len(sys.argv) > 3:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build_bench_index as BB
    total = int(float(sys.argv[3]))
    base = BB.index_base(total); contigs = BB.genome(total)
else:
    base, contigs = bench.build_index(os.path.join(ROOT, ".bench_cache"), 4_900_000)
ix = api.Index(base)
names = [str(i) for i in range(nreads)]
if paired:
    m1, m2 = synth.make_pairs(contigs, nreads, 101, bench.SEED + 7, sub_rate=0.005)
    c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
    st = api.Stream(ix, max_reads=nreads, max_bases=c1.size)
    st.set_reads(c1, o1); st.set_read_names(names); st.set_mates(c2, o2, names)
    run = st.align_pairs_run
else:
    reads, _ = synth.make_reads(contigs, nreads, 101, bench.SEED + 1000, sub_rate=0.005)
    codes, offs = synth.flatten_reads(reads)
    st = api.Stream(ix, max_reads=nreads, max_bases=codes.size)
    st.set_reads(codes, offs); st.set_read_names(names)
    run = st.align_run
for _ in range(3):
    run()
st.sync()
c = st.counters()
print("align %.3f ms (main kernel %.3f ms) aligned %d overflow %d second-pass %d  ranks/read %.1f steps/read %.1f" % (
    c.ms_align, c.ms_align_kernel, c.n_aligned, c.n_overflow, c.n_second_pass, c.n_rank / nreads, c.n_sa_steps / nreads))
L = api.lib()
if hasattr(L, "h2g_go_prof"):
    v = (C.c_ulonglong * 80)()
    L.h2g_go_prof.argtypes = [C.c_void_p, C.c_void_p]
    if L.h2g_go_prof(st.h, v) == 0 and v[47]:
        names = "NONE PSEARCH GCOORDS EXTEND LSEARCH LCOORDS GSEARCH COMBINE ADJUST ADJMEMBER SW FINISH".split()
        tot = sum(v[k] for k in range(0, 16))
        print("trips %d  avg slots per trip %.1f  total wave-ticks %d" % (v[47], v[46] / max(1, v[47]), tot))
        for k, nm in ((0, "refill"), (1, "control"), (2, "vote")):
            print("  %-10s %5.1f %%" % (nm, 100.0 * v[k] / tot))
        for op in range(1, 12):
            if v[3 + op]:
                print("  %-10s %5.1f %%   executions: avg %.1f of 64 lanes" % (names[op], 100.0 * v[3 + op] / tot, v[20 + op] / max(1, v[32 + op])))
        print("  control time by the ring the trip popped (0 = new reads):", {names[op] if op else "FETCH": "%.1f %%" % (100.0 * v[64 + op] / tot) for op in range(0, 12) if v[64 + op]})
        print("  executions:", {names[op]: int(v[32 + op]) for op in range(1, 12) if v[32 + op]})
