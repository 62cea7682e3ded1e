#!/usr/bin/env python3
"""k_align timing under the scheduling knobs (env H2G_ALIGN_SORT, H2G_ALIGN_OCC); prints a result checksum so that
variants can be compared for identical output."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from hisat2_amd import api, synth
nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
glen = int(sys.argv[2]) if len(sys.argv) > 2 else 4_900_000
import time
t0 = time.time()
base, contigs = bench.build_index(os.path.join(ROOT, ".bench_cache") if glen < 10_000_000 else "/tmp/h2g_bigidx", glen)
print('index ready in %.1f s' % (time.time() - t0), flush=True)
sub = float(os.environ.get('H2G_SUB', '0.005'))
reads, _ = synth.make_reads(contigs, nreads, 101, bench.SEED + 1000, sub_rate=sub)
if os.environ.get('H2G_ONLY_ERR'):
    # keep only reads that differ from the genome (>= 1 substitution): redraw with a guaranteed substitution
    rng = np.random.default_rng(5)
    pos = rng.integers(0, 101, size=nreads)
    reads[np.arange(nreads), pos] = (reads[np.arange(nreads), pos] + 1 + rng.integers(0, 3, size=nreads)) & 3
codes, offs = synth.flatten_reads(reads)
ix = api.Index(base); st = api.Stream(ix, max_reads=nreads, max_bases=codes.size)
st.set_reads(codes, offs); st.set_read_names([str(i) for i in range(nreads)])
for _ in range(3):
    st.align_run()
st.sync()
c = st.counters()
res, aln = st.align_fetch(0, 20000)
allres, _ = st.align_fetch(with_alignments=False)
ck = zlib.crc32(allres.tobytes()) ^ zlib.crc32(bytes(aln))
p = st.seed_params(True)
for _ in range(2):
    st.seed_extend_run(p)
st.sync()
c2 = st.counters()
print("genome %d: seed stage search %.3f ms resolve+extend %.3f ms; align ranks/read %.1f sa steps/read %.1f sides/read %.1f" % (
    glen, c2.ms_search, c2.ms_resolve_extend, c.n_rank / nreads, c.n_sa_steps / nreads, c.n_side / nreads))
print("sub %s only_err %s" % (os.environ.get("H2G_SUB","0.005"), os.environ.get("H2G_ONLY_ERR","0")), end="  ")
print("SORT=%s OCC=%s: align total %.3f ms (kernel %.3f ms)  aligned %d overflow %d  crc %08x" % (
    os.environ.get("H2G_ALIGN_SORT", "1"), os.environ.get("H2G_ALIGN_OCC", "2"), c.ms_align, c.ms_align_kernel, c.n_aligned, c.n_overflow, ck))
