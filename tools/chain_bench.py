#!/usr/bin/env python3
"""Chains of dependent rank queries (libh2g's measurement kernel k_rank_chain, h2g_rank_chain_bench): what several chains in flight per LANE are worth at
the occupancy of the compact-state pass.  The search loops of go() are chains — the row of step i + 1 comes out of the rank of step i — and the pass runs
512 lanes per CU (256 VGPRs: two waves per SIMD), one chain each.  For synthetic linear (64 B) and graph (128 B) sides at GRCh38 scale (0.98 GB), the same
number of chains is walked (a) at full occupancy, one chain per lane; (b) one 512-thread workgroup per CU (dynamic LDS bounds it as the pass's staging area
does), with 1, 2, 4, 8 chains per lane.  Reported: steps per second, the algorithmic GB/s (64 or 128 B per step), the checksum (must not depend on the split).
Then whole graph LF steps (k_glf_chain) on a real SNP-graph index: 1, 2, 4 walks per lane, one after the other or stage by stage.

usage: chain_bench.py [chains=2^21] [steps=64] [graph index base (default: a 32 Mbp SNP graph built here; "none": skip the graph LF walks)] [compact]   -> one JSON line
(compact: label -> [ms, GB/s or G LF steps/s] only — the form bench.py's extras carry as their last key)"""
import ctypes as C, json, sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hisat2_amd import api


def main():
    nchains = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    f = api.lib().h2g_rank_chain_bench
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    f.restype = C.c_int
    out = {"chains": nchains, "steps": steps}
    for graph, nsides, name, side in ((False, 15_300_000, "linear_64B", 64), (True, 7_650_000, "graph_128B", 128)):
        ix = api.Index(synth_sides=nsides, seed=20260927, device=0, graph=graph)
        st = api.Stream(ix)
        rows = {}
        sums = set()
        # occupancy of the pass: 512 lanes per CU x 256 CUs = 131072 lanes in flight; chains beyond that queue up behind them
        for label, block, lds, per_lane in (("full_occupancy_1", 256, 0, 1), ("pass_occupancy_1", 512, 96 * 1024, 1), ("pass_occupancy_2", 512, 96 * 1024, 2),
                                            ("pass_occupancy_4", 512, 96 * 1024, 4), ("pass_occupancy_8", 512, 96 * 1024, 8), ("full_occupancy_4", 256, 0, 4)):
            ms, cs = C.c_float(0), C.c_uint64(0)
            rc = f(st.h, nchains, per_lane, steps, block, lds, 12345, 3, C.byref(ms), C.byref(cs))
            if rc != 0:
                rows[label] = {"error": rc}
                continue
            per_s = nchains * steps / (ms.value * 1e-3)
            rows[label] = {"block": block, "lds_bytes": lds, "chains_per_lane": per_lane, "ms": round(ms.value, 3), "steps_per_s": per_s, "GB/s": per_s * side / 1e9,
                           "frac_of_8TBs": per_s * side / 8e12, "checksum": cs.value}
            sums.add(cs.value)
        rows["checksums_equal"] = len(sums) == 1
        out[name] = rows
        st.close(); ix.close()
    # whole graph LF steps (rank, rank_M, select_F: two or three dependent lines) on a REAL SNP-graph index (built here: ~15 s for 32 Mbp), walks of `steps` steps:
    # C walks per lane one after the other (what a lane of the pass does, C times) against C walks stage by stage (h2g_graph_staged.h)
    glen = int(float(os.environ.get("H2G_CHAIN_GRAPH_GENOME", "32e6")))
    compact = len(sys.argv) > 4 and sys.argv[4] == "compact"
    if len(sys.argv) > 3 and sys.argv[3] == "none":               # no graph index at hand and no time to build one: the rank chains only
        ok = all(out[k]["checksums_equal"] for k in ("linear_64B", "graph_128B"))
        if compact:
            out = {"chains": nchains, "steps": steps, "checksums_equal": ok,
                   **{k: {lab: ([r["ms"], round(r["GB/s"], 1)] if "ms" in r else r) for lab, r in out[k].items() if isinstance(r, dict)} for k in ("linear_64B", "graph_128B")}}
        print(json.dumps(out))
        return 0 if ok else 1
    if len(sys.argv) > 3 and sys.argv[3] not in ("", "-"):
        base, glen = sys.argv[3], None
    else:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import build_graph_bench_index as GB
        base, _ = GB.build(glen, 250)
    f2 = api.lib().h2g_glf_chain_bench
    f2.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    f2.restype = C.c_int
    ix = api.Index(base, device=0)
    st = api.Stream(ix)
    rows, sums = {"genome": glen, "index_device_bytes": int(ix.info.device_bytes)}, set()
    for occ, block, lds in (("pass_occupancy", 512, 96 * 1024), ("full_occupancy", 256, 0)):
        for per_lane in (1, 2, 4):
            for staged in ((0,) if per_lane == 1 else (0, 1)):
                ms, cs = C.c_float(0), C.c_uint64(0)
                rc = f2(st.h, nchains, per_lane, steps, block, lds, 12345, staged, 3, C.byref(ms), C.byref(cs))
                label = "%s_%d_%s" % (occ, per_lane, "staged" if staged else "one_after_the_other")
                if rc != 0:
                    rows[label] = {"error": rc}
                    continue
                rows[label] = {"ms": round(ms.value, 3), "lf_steps_per_s": nchains * steps / (ms.value * 1e-3), "checksum": cs.value}
                sums.add(cs.value)
    rows["checksums_equal"] = len(sums) == 1
    out["graph_lf_walks"] = rows
    st.close(); ix.close()
    ok = all(out[k]["checksums_equal"] for k in ("linear_64B", "graph_128B", "graph_lf_walks"))
    if compact:
        c = {"chains": nchains, "steps": steps, "checksums_equal": ok}
        for k in ("linear_64B", "graph_128B"):
            c[k] = {lab: ([r["ms"], round(r["GB/s"], 1)] if "ms" in r else r) for lab, r in out[k].items() if isinstance(r, dict)}
        c["graph_lf_walks_G_steps_per_s"] = {lab: ([r["ms"], round(r["lf_steps_per_s"] / 1e9, 3)] if "ms" in r else r) for lab, r in out["graph_lf_walks"].items() if isinstance(r, dict)}
        c["graph_lf_index_bytes"] = out["graph_lf_walks"]["index_device_bytes"]
        out = c
    print(json.dumps(out))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
