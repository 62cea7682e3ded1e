#!/usr/bin/env python3
"""Short GPU run for rocprofv3 passes: 3 launches of the fused seed stage on the bench workload and one launch
of each Occ-rank variant at GRCh38 scale.  (bench.py is the timed harness; this only keeps profiler passes short.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from hisat2_amd import api, synth  # noqa: E402

nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 26
base, contigs = bench.build_index(os.path.join(ROOT, ".bench_cache"), 4_900_000)
reads, truth = synth.make_reads(contigs, nreads, 101, bench.SEED + 1000, sub_rate=0.005)
codes, offs = synth.flatten_reads(reads)
ix = api.Index(base)
st = api.Stream(ix, max_reads=nreads, max_bases=codes.size)
st.set_reads(codes, offs)
p = st.seed_params(True)
for _ in range(3):
    st.seed_extend_run(p)
st.sync()
st.set_read_names([str(i) for i in range(nreads)])
for _ in range(2):
    st.align_run()
st.sync()
c = st.counters()
print("align: %.3f ms, n_rank %d n_side %d n_sa_steps %d aligned %d overflow %d" % (c.ms_align, c.n_rank, c.n_side, c.n_sa_steps, c.n_aligned, c.n_overflow))
print("seed stage: search %.3f ms, resolve+extend %.3f ms, n_side %d n_sa_steps %d" % (c.ms_search, c.ms_resolve_extend, c.n_side, c.n_sa_steps))
rix = api.Index(synth_sides=15_300_000, seed=bench.SEED)
rst = api.Stream(rix)
for v in (0, 1, 2):
    ms, ck = rst.rank_synth(nq, bench.SEED, variant=v, repeats=1)
    print("rank variant %d: %.3f ms  %.1f GB/s" % (v, ms, nq * 64 / ms / 1e6))
rst.close(); rix.close()
gix = api.Index(synth_sides=7_650_000, seed=bench.SEED, graph=True)
gst = api.Stream(gix)
for v in (0, 1):
    ms, ck = gst.rank_synth(nq, bench.SEED, variant=v, repeats=1)
    print("graph rank variant %d: %.3f ms  %.1f GB/s" % (v, ms, nq * 128 / ms / 1e6))
gst.close(); gix.close()
# Smith-Waterman: 65536 problems framed around the true positions
nsw = min(65536, nreads)
swq = [api.SwQuery(i, int(truth[i][2]), int(truth[i][0]), int(truth[i][1]), -20, i + 1) for i in range(nsw)]
res, ms = st.sw_align(swq, repeats=2)
print("sw: %d problems %.3f ms (fill + backtrace kernels), found %d" % (nsw, ms, sum(1 for r in res if r.found)))
# paired-end go()
npairs = nreads // 2
m1, m2 = synth.make_pairs(contigs, npairs, 101, bench.SEED + 77, sub_rate=0.005)
c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
qn = [str(i) for i in range(npairs)]
pst = api.Stream(ix, max_reads=npairs, max_bases=c1.size)
pst.set_reads(c1, o1); pst.set_read_names(qn); pst.set_mates(c2, o2, qn)
for _ in range(2):
    pst.align_pairs_run()
pst.sync()
pc = pst.counters()
print("pairs: %.3f ms for %d pairs, concordant %d" % (pc.ms_align, npairs, pc.n_aligned))
pst.close()
# go() on the SNP-graph index of the bench genome (built/cached by bench.py)
gbase = os.path.join(ROOT, ".bench_cache", "rnd4900000_s%d_snp" % bench.SEED, "g")
if os.path.exists(gbase + ".8.ht2"):
    var = synth.make_snps(contigs, bench.SEED + 5, every=250, names=["ecoli_substitute"])
    alt = synth.apply_snps(contigs, var, names=["ecoli_substitute"])
    greads, _ = synth.make_reads(alt, nreads, 101, bench.SEED + 4242, sub_rate=0.005)
    gc, go = synth.flatten_reads(greads)
    gix2 = api.Index(gbase)
    gst2 = api.Stream(gix2, max_reads=nreads, max_bases=gc.size)
    gst2.set_reads(gc, go); gst2.set_read_names([str(i) for i in range(nreads)])
    for _ in range(2):
        gst2.align_run()
    gst2.sync()
    c3 = gst2.counters()
    print("graph go(): %.3f ms for %d reads, aligned %d, n_rank %d" % (c3.ms_align, nreads, c3.n_aligned, c3.n_rank))
