#!/usr/bin/env python3
"""Where does the compact-state pass pay on a graph index?  SNP-graph indexes over the seeded genome at several sizes (tools/build_graph_bench_index.py;
staged ones are used as they are, the others are built here), N pairs from the alternate haplotype, and the steady-state step of queued runs with the
fast pass on and off (h2g_stream_tune "fast"), plus a few machine-pass shares ("mach_div") with it on.  One JSON line per size.

usage: graph_scale.py SIZES(bp, comma separated) [pairs=1000000] [mach_divs=400]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from hisat2_amd import api, synth
import build_bench_index as BB, build_graph_bench_index as GB


def tune(st, key, v):
    f = api.lib().h2g_stream_tune
    f.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
    assert f(st.h, key.encode(), v) == 0


def steady(st, steps=5):
    for _ in range(3):
        st.align_pairs_run()
    st.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        st.align_pairs_run()
    st.sync()
    dt = (time.perf_counter() - t0) / steps
    st.align_pairs_run(); st.sync()
    c = st.counters()
    return {"ms_per_step": round(dt * 1e3, 2), "fast_kernel_ms": round(float(c.ms_fast_kernel), 2), "machine_pass_ms": round(float(c.ms_align_kernel), 2), "handed_on": int(c.n_fast_bail),
            "concordant": int(c.n_aligned)}


def main():
    sizes = [int(float(x)) for x in sys.argv[1].split(",")]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    divs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [400]
    for glen in sizes:
        t0 = time.time()
        base, info = GB.build(glen, 250)
        contigs = BB.genome(glen)
        alt = synth.apply_snps(contigs, GB.variants(glen, 250, contigs), names=GB.names(glen))
        m1, m2 = synth.make_pairs(alt, n, 101, 20260925 + 79, frag_mean=300, frag_sd=30, sub_rate=0.005)
        c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
        names = [str(i) for i in range(n)]
        ix = api.Index(base, device=0)
        st = api.Stream(ix, max_reads=n, max_bases=c1.size)
        st.set_reads(c1, o1); st.set_read_names(names); st.set_mates(c2, o2, names)
        out = {"genome": glen, "pairs": n, "index_device_bytes": int(ix.info.device_bytes), "built_s": None if info is None else round(info["build_seconds"])}
        tune(st, "fast", 0)
        out["machine_only"] = steady(st)
        tune(st, "fast", 1)
        for d in divs:
            tune(st, "mach_div", d)
            out["fast_mach_div_%d" % d] = steady(st)
        out["seconds"] = round(time.time() - t0)
        print(json.dumps(out), flush=True)
        st.close(); ix.close()


if __name__ == "__main__":
    main()
