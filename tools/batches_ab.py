#!/usr/bin/env python3
"""One resident batch run K times against B distinct resident batches (h2g_stream_select_batch), same stream, same process, alternating: the steady-state step of queued runs.
usage: batches_ab.py GENOME_BP [pairs=1000000] [B=10] [rounds=3]   -> one JSON line per measurement"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from hisat2_amd import api, synth
import build_bench_index as BB


def main():
    glen = int(float(sys.argv[1]))
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    cache = os.path.join(ROOT, ".bench_cache")
    base = BB.build(glen, cache=cache)
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
    for b in range(B):                          # every batch once: nothing left to allocate
        st.select_batch(b); st.align_pairs_run()
    st.sync()
    K = 40
    for r in range(rounds):
        for nb in (1, B):
            for k in range(10):
                st.select_batch(k % nb); st.align_pairs_run()
            st.sync()
            t0 = time.perf_counter()
            for k in range(K):
                st.select_batch(k % nb); st.align_pairs_run()
            st.sync()
            dt = (time.perf_counter() - t0) / K
            c = st.counters()
            print(json.dumps({"round": r, "distinct_batches": nb, "ms_per_step": round(dt * 1e3, 3), "reads_per_s": round(2 * n / dt), "fast_kernel_ms_last": round(float(c.ms_fast_kernel), 2),
                              "handed_on_last": int(c.n_fast_bail)}), flush=True)
    st.close(); ix.close()


if __name__ == "__main__":
    main()
