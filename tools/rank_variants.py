#!/usr/bin/env python3
"""The Occ-rank micro-kernels on the synthetic side arrays of bench.py (2^28 queries over 0.98 GB): variant -> ms, GB/s, checksum; and the 2^20 sampled outputs of variant 0
against the oracle's mapLF.  usage: rank_variants.py [log2 queries = 28]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from hisat2_amd import api
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
out = {}
for graph, nsides in ((False, 15_300_000), (True, 7_650_000)):
    rix = api.Index(synth_sides=nsides, seed=bench.SEED, device=0, graph=graph)
    rst = api.Stream(rix)
    for v in ((0, 1) if graph else (0, 10, 1, 2, 6, 7, 8, 11, 12)):
        rst.rank_synth(n, bench.SEED, variant=v, repeats=1)
        runs = []
        for _ in range(3):
            ms, ck = rst.rank_synth(n, bench.SEED, variant=v, repeats=3)
            runs.append(round(n * (128 if graph else 64) / (ms * 1e-3) / 1e9, 1))
        out[("graph_" if graph else "linear_") + str(v)] = {"GB/s": runs, "frac_of_8TBs": round(max(runs) / 8000.0, 4), "checksum": int(ck)}
    if not graph and n >= (1 << 24):
        import rank_synth_check as RC
        import h2o_py as HO
        stride = n >> 20
        rst.rank_synth(n, bench.SEED, variant=0, repeats=1)
        got = rst.rank_synth_sample(stride, 1 << 20)
        want = RC.sampled_expect(HO.load(), nsides, bench.SEED, n, stride, 1 << 20)
        out["sampled_vs_oracle_mapLF"] = {"samples": int(len(want)), "differing": int((want != got).sum())}
    rst.close(); rix.close()
print(json.dumps(out, indent=1))
