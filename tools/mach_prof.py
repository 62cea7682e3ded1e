#!/usr/bin/env python3
"""Wave-level time split of the general machine's pass over the reads the fast pass hands on (library built with -DH2G_GO_PROF, tools/build_prof_lib.sh,
loaded through H2G_LIB): one run on its own on the repeat-structured / graph leg, then h2g_go_prof of that run's machine pass.
usage: H2G_LIB=hisat2_amd/csrc/obj_prof/libh2g_prof.so mach_prof.py rep|graph|rnd GENOME_BP [pairs]"""
import ctypes as C, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from hisat2_amd import api, synth
import build_bench_index as BB

kind, glen = sys.argv[1], int(float(sys.argv[2]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
cache = os.path.join(ROOT, ".bench_cache")
if kind == "rep":
    base = os.path.join(cache, f"rep{glen}_s{bench.SEED}", "g")
    contigs = synth.make_repeat_genome(BB.contig_lens(glen), bench.SEED + 77)
    if not os.path.exists(base + ".8.ht2"):
        os.makedirs(os.path.dirname(base), exist_ok=True)
        synth.write_fasta(base + ".fa", contigs)
        subprocess.run([os.path.join(bench.REF, "hisat2-build-s"), "-q", "-p", str(BB.usable_cpus()), base + ".fa", base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.remove(base + ".fa")
    m1, m2 = synth.make_pairs(contigs, n, 101, bench.SEED + 78, sub_rate=0.005)
elif kind == "graph":
    import build_graph_bench_index as GB
    base, info = GB.build(glen, 250, cache=cache)
    contigs = BB.genome(glen)
    alt = synth.apply_snps(contigs, GB.variants(glen, 250, contigs), names=GB.names(glen))
    m1, m2 = synth.make_pairs(alt, n, 101, bench.SEED + 79, frag_mean=300, frag_sd=30, sub_rate=0.005)
else:
    base, contigs = bench.small_index(cache, glen)
    m1, m2 = synth.make_pairs(contigs, n, 101, bench.SEED + 7, sub_rate=0.005)
c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
names = [str(i) for i in range(n)]
ix = api.Index(base, device=0)
st = api.Stream(ix, max_reads=n, max_bases=c1.size)
st.set_reads(c1, o1); st.set_read_names(names); st.set_mates(c2, o2, names)
for _ in range(2):
    st.align_pairs_run(); st.sync()
c = st.counters()
print("one run on its own: fast kernel %.2f ms, machine pass %.2f ms (with its second pass %.2f ms), handed on %d, second pass %d" % (c.ms_fast_kernel, c.ms_align_kernel, c.ms_align, c.n_fast_bail, c.n_second_pass))
L = api.lib()
v = (C.c_ulonglong * 80)()
L.h2g_go_prof.argtypes = [C.c_void_p, C.c_void_p]
if L.h2g_go_prof(st.h, v) == 0 and v[47]:
    ops = "NONE PSEARCH GCOORDS EXTEND LSEARCH LCOORDS GSEARCH COMBINE ADJUST ADJMEMBER SW FINISH".split()
    tot = sum(v[k] for k in range(0, 16))
    print("machine pass: wave-trips %d, slots per trip %.2f, wave-ticks %d (shader clock, ~2.4 GHz: %.1f wave-ms)" % (v[47], sum(v[20 + op] for op in range(1, 12)) / max(1, v[47]), tot, tot / 2.4e6))
    print("  %-12s %5.1f %%" % ("pop+load/new", 100.0 * v[0] / tot))
    print("  %-12s %5.1f %%" % ("control+push", 100.0 * v[1] / tot))
    for op in range(1, 12):
        if v[3 + op]:
            print("  %-12s %5.1f %%   executions %d: avg %.2f of 64 lanes, %.1f us per execution" % (ops[op], 100.0 * v[3 + op] / tot, v[32 + op], v[20 + op] / max(1, v[32 + op]), v[3 + op] / max(1, v[32 + op]) / 2400.0))
    print("  control time by the ring popped:", {ops[op] if op else "FETCH": "%.1f %%" % (100.0 * v[64 + op] / tot) for op in range(0, 12) if v[64 + op]})
else:
    print("no profile (library without -DH2G_GO_PROF?)")
st.close(); ix.close()
