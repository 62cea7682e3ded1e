#!/usr/bin/env python3
"""The Occ-rank micro-kernel alone, for a `rocprofv3 --pmc FETCH_SIZE` pass: 2^28 uniform-random rank queries on 0.98 GB of synthetic sides,
one launch per variant (linear: k_rank_v0 / v1 / v2, 64 B sides; graph: k_rank_g0 / g1, 128 B sides).  The algorithmic bytes per launch are
known exactly (queries x side bytes), so FETCH_SIZE of these launches calibrates the counter for scattered 64 B and 128 B lines
(MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern").  usage: rank_pmc.py [log2 queries = 28]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hisat2_amd import api
q = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
out = {}
for graph, nsides in ((False, 15_300_000), (True, 7_650_000)):
    ix = api.Index(synth_sides=nsides, seed=20260927, device=0, graph=graph)
    st = api.Stream(ix)
    for v in ((0, 1) if graph else (0, 1, 2)):
        ms, ck = st.rank_synth(q, 20260927, variant=v, repeats=1)
        out[("k_rank_g%d" if graph else "k_rank_v%d") % v] = {"queries": q, "side_bytes": 128 if graph else 64, "algorithmic_bytes": q * (128 if graph else 64), "ms": ms,
                                                             "GB/s": q * (128 if graph else 64) / (ms * 1e-3) / 1e9}
    st.close(); ix.close()
print(json.dumps(out))
