#!/usr/bin/env python3
"""Occ-rank micro-kernel (2^28 uniform queries over GRCh38-scale synthetic sides) at several occupancies and lanes-per-side variants.
usage: r05_rank_occupancy.py [log2 queries = 28]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from hisat2_amd import api
nq = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
for graph, nsides in ((False, 15_300_000), (True, 7_650_000)):
    ix = api.Index(synth_sides=nsides, seed=bench.SEED, device=0, graph=graph)
    st = api.Stream(ix)
    ref = None
    for v, name in ((0, "full occupancy"), (6, "16 waves per CU"), (7, "8 waves per CU"), (8, "4 waves per CU"), (9, "2 waves per CU")) if not graph else ((0, "full occupancy"),):
        st.rank_synth(nq, bench.SEED, variant=v, repeats=1)
        ms, ck = st.rank_synth(nq, bench.SEED, variant=v, repeats=3)
        ref = ck if ref is None else ref
        gbs = nq * (128 if graph else 64) / (ms * 1e-3) / 1e9
        print(json.dumps({"sides": "graph 128 B" if graph else "linear 64 B", "launch": name, "ms": round(ms, 3), "GB/s": round(gbs, 1), "frac_of_8TBs": round(gbs / 8000, 4), "G_queries_per_s": round(nq / ms / 1e6, 2), "checksum_equal": ck == ref}), flush=True)
    st.close(); ix.close()
