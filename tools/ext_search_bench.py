#!/usr/bin/env python3
"""LDS-staged local-index search (h2g_ext_search) vs the all-HBM kernel: 1 M localGFMSearch queries as hybridSearch_recur issues
them (one per read, in the local index under the read), on the E. coli-size index and on a larger genome if given.
usage: ext_search_bench.py [genome_bases]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import bench
from hisat2_amd import api, synth
res = {}
cases = [("ecoli_size_4.9Mbp", None)]
if len(sys.argv) > 1:
    cases.append(("genome_%s" % sys.argv[1], int(float(sys.argv[1]))))
for tag, total in cases:
    if total is None:
        base, contigs = bench.small_index(os.path.join(ROOT, ".bench_cache"), 4_900_000)
    else:
        import build_bench_index as BB
        base = BB.build(total); contigs = BB.genome(total)
    n = 1_000_000
    reads, truth = synth.make_reads(contigs, n, 101, 4711, sub_rate=0.005)
    codes, offs = synth.flatten_reads(reads)
    ix = api.Index(base); st = api.Stream(ix, max_reads=n, max_bases=codes.size)
    st.set_reads(codes, offs)
    L = api.lib()
    lens = np.array([len(c) for c in contigs])
    q = (api.ExtSearchQuery * n)()
    rng = np.random.default_rng(3)
    rdoff = rng.integers(30, 101, size=n)
    first = {}
    for i in range(n):
        ci, pos, fw = int(truth[i][0]), int(truth[i][1]), int(truth[i][2])
        key = (ci, pos // 56320)
        if key not in first:
            first[key] = L.h2g_local_index_of(ix.h, ci, pos)
        q[i].read = i; q[i].rdoff = int(rdoff[i]); q[i].lidx = first[key]; q[i].maxHitLen = 0xffff; q[i].fw = fw; q[i].uniqueStop = 1
    out = {}
    for stage_min in (0, 8, 64):
        st.ext_search(q, stage_min=stage_min)
        hits, s = st.ext_search(q, stage_min=stage_min)
        out["stage_min_%d" % stage_min] = {"lds_hit_rate": s.n_staged / max(1, s.n_local), "buckets": int(s.n_buckets), "workgroup_buckets_staged": int(s.n_buckets_staged),
                                           "lds_bytes_staged": int(s.lds_bytes_staged), "ms_staged_kernel": float(s.ms_staged), "ms_hbm_kernel": float(s.ms_hbm),
                                           "ms_total": float(s.ms_staged + s.ms_hbm), "sides_read": int(sum(h.nside for h in hits)), "found": int(sum(1 for h in hits if h.nelt > 0))}
    res[tag] = {"queries": n, "local_indexes": int(ix.info.nLocal), **out}
    st.close(); ix.close()
print(json.dumps(res))
