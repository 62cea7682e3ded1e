#!/usr/bin/env python3
"""How many machine passes in flight?  (the hand-on path co-bounds the step.)  One index, one resident batch, one stream; the steady-state step
of queued runs for several (mstreams, mach_total, fast_reserve, mach_div) settings of h2g_stream_tune, with a checksum of every result record per setting
(it must not move) and the kernels' own times.  One JSON line per setting.

usage: queued_steps.py rep|rnd|graph GENOME_BP [pairs=1000000] [settings "M,total,reserve,div[,orphan[,drain_grid]];..."]"""
import ctypes as C, json, os, subprocess, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from hisat2_amd import api, synth
import build_bench_index as BB

ALN_DT = np.dtype([("fw", "<u4"), ("tidx", "<u4"), ("toff", "<u4"), ("len", "<u4"), ("trim5", "<u4"), ("trim3", "<u4"), ("nedits", "<u4"), ("spl", "<u4"),
                   ("score", "<i8"), ("edits", [("pos", "<u4"), ("chr", "u1"), ("qchr", "u1"), ("type", "u1"), ("pad", "u1"), ("snp", "<u4")], 32)])


def aln_crc(arr, n):
    a = np.frombuffer(arr, dtype=ALN_DT, count=n).copy()
    keep = np.arange(32)[None, :] < a["nedits"][:, None]
    for f in ("pos", "chr", "qchr", "type", "pad", "snp"):
        a["edits"][f][~keep] = 0
    return zlib.crc32(a.tobytes())


def tune(st, key, v):
    f = api.lib().h2g_stream_tune
    f.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
    assert f(st.h, key.encode(), v) == 0, key


def main():
    kind = sys.argv[1]
    glen = int(float(sys.argv[2]))
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
    settings = sys.argv[4] if len(sys.argv) > 4 else "2,96,-1,0;4,96,-1,0;6,96,-1,0;8,96,-1,0;4,128,-1,0;4,64,-1,0;4,96,0,0;4,96,32,0"
    cache = os.path.join(ROOT, ".bench_cache")
    t0 = time.time()
    if kind == "rep":
        d = os.path.join(cache, f"rep{glen}_s{bench.SEED}")
        base = os.path.join(d, "g")
        contigs = synth.make_repeat_genome(BB.contig_lens(glen), bench.SEED + 77)
        if not os.path.exists(base + ".8.ht2"):
            os.makedirs(d, exist_ok=True)
            synth.write_fasta(base + ".fa", contigs)
            subprocess.run([os.path.join(bench.REF, "hisat2-build-s"), "-q", "-p", str(BB.usable_cpus()), base + ".fa", base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            os.remove(base + ".fa")
        m1, m2 = synth.make_pairs(contigs, n, 101, bench.SEED + 78, sub_rate=0.005)
    elif kind == "graph":
        import build_graph_bench_index as GB
        base, info = GB.build(glen, 250, cache=cache)
        contigs = BB.genome(glen)
        alt = synth.apply_snps(contigs, GB.variants(glen, 250, contigs), names=GB.names(glen))
        m1, m2 = synth.make_pairs(alt, n, 101, bench.SEED + 79, frag_mean=300, frag_sd=30, sub_rate=0.005)
    else:
        if glen < 10_000_000:
            base, contigs = bench.small_index(cache, glen)
        else:
            base = BB.build(glen, cache=cache)
            contigs = BB.genome(glen)
        m1, m2 = synth.make_pairs(contigs, n, 101, bench.SEED + 7, sub_rate=0.005)
    print(json.dumps({"kind": kind, "genome": glen, "pairs": n, "index_and_reads_s": round(time.time() - t0, 1)}), flush=True)
    c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
    names = [str(i) for i in range(n)]
    ix = api.Index(base, device=0)
    st = api.Stream(ix, max_reads=n, max_bases=c1.size)
    st.set_reads(c1, o1); st.set_read_names(names); st.set_mates(c2, o2, names)
    first_ck = None
    for sset in settings.split(";"):
        f_ = [int(x) for x in sset.split(",")]
        M, total, reserve, div = f_[:4]
        orphan, dgrid = (f_[4] if len(f_) > 4 else -1), (f_[5] if len(f_) > 5 else 32)
        tune(st, "mate_handover", f_[6] if len(f_) > 6 else -1)      # (the end of the batch: h2g_stream_tune "orphan" -1 = the default policy, 0 = off; "drain_grid")
        tune(st, "mstreams", M); tune(st, "mach_total", total); tune(st, "fast_reserve", reserve); tune(st, "mach_div", div); tune(st, "orphan", orphan); tune(st, "drain_grid", dgrid)
        for _ in range(M + 2):                      # (each machine stream allocates its workspace on its first pass)
            st.align_pairs_run()
        st.sync()
        K = 3 * M + 6
        t1 = time.perf_counter()
        for _ in range(K):
            st.align_pairs_run()
        st.sync()
        dt = (time.perf_counter() - t1) / K
        st.align_pairs_run(); st.sync()
        c = st.counters()
        res, a1, o1_, a2, o2_ = st.align_pairs_fetch_dense()
        rb = np.frombuffer(bytes(res), dtype=np.uint8).reshape(n, -1)
        ck = aln_crc(a1, int(o1_[n])) ^ aln_crc(a2, int(o2_[n]))
        del a1, a2, res
        if first_ck is None:
            first_ck = ck
        print(json.dumps({"mstreams": M, "mach_total": total, "fast_reserve": reserve, "mach_div": div, "orphan": orphan, "drain_grid": dgrid, "mate_handover": (f_[6] if len(f_) > 6 else -1), "drain_ms_solo": round(float(c.ms_drain_kernel), 2), "adopted": int(c.n_adopted), "ms_per_step": round(dt * 1e3, 2), "reads_per_s": round(2 * n / dt),
                          "fast_kernel_ms_solo": round(float(c.ms_fast_kernel), 2), "machine_pass_ms_solo": round(float(c.ms_align_kernel), 2), "handed_on": int(c.n_fast_bail),
                          "second_pass": int(c.n_second_pass), "flagged": int(c.n_overflow), "concordant": int(c.n_aligned), "records_crc": "%08x" % ck, "crc_same_as_first": ck == first_ck}), flush=True)
    st.close(); ix.close()


if __name__ == "__main__":
    main()
