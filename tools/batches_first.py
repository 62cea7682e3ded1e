#!/usr/bin/env python3
"""What does a resident batch's FIRST run cost?  B batches are filled (batch 0 run once first, as bench.py does), then every run is timed on its own (sync after each):
runs 0 .. 3B - 1 over batches 0 .. B - 1 in turn.  usage: batches_first.py GENOME_BP [pairs=1000000] [B=10]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from hisat2_amd import api, synth
import build_bench_index as BB

glen = int(float(sys.argv[1])); n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000; B = int(sys.argv[3]) if len(sys.argv) > 3 else 10
base = BB.build(glen, cache=os.path.join(ROOT, ".bench_cache"))
contigs = BB.genome(glen)
ix = api.Index(base, device=0)
st = api.Stream(ix, max_reads=n, max_bases=n * 101)
for b in range(B):
    m1, m2 = synth.make_pairs(contigs, n, 101, bench.SEED + 7 + 100003 * b, sub_rate=0.005)
    c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
    names = [str(b * n + i) for i in range(n)]
    st.select_batch(b)
    st.set_reads(c1, o1); st.set_read_names(names); st.set_mates(c2, o2, names)
    if b == 0:
        st.align_pairs_run(); st.sync()
rows = []
for k in range(3 * B):
    t0 = time.perf_counter()
    st.select_batch(k % B); st.align_pairs_run(); st.sync()
    dt = time.perf_counter() - t0
    c = st.counters()
    rows.append({"run": k, "batch": k % B, "ms": round(dt * 1e3, 2), "fast_ms": round(float(c.ms_fast_kernel), 2), "machine_ms": round(float(c.ms_align_kernel), 2), "total_ms": round(float(c.ms_align), 2)})
for r in rows:
    print(json.dumps(r), flush=True)
st.close(); ix.close()
