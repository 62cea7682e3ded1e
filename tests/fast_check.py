"""The fast path of go() (hisat2_amd/csrc/h2g_fast.h) against the general machine, both instantiated on the host (tests/emul):
every read / pair runs through both; what the fast path completes must equal the machine's result bit for bit."""
import ctypes as C
import os

import numpy as np

from h2gemu_py import Emu

BAIL_REASONS = ["none", "input", "longpool", "subsample", "coords", "nghits", "edits", "depth", "localhits", "gsearch", "nres", "searched",
                "redundant", "mate", "npairs", "partial", "straddle", "other", "indel", "tail", "iedges", "gwalk"]    # == h2g_fast.h FB_* (FB_COUNT entries: tests/test_fast_path_cpu.py checks it)


def fast_check(base, reads1, reads2=None, names=None, quals=None, options=(), variant=""):
    """-> dict(completed, mismatching, bails {reason: n}, bad [read ids], done flags)"""
    e = Emu(base, variant)
    if options:
        from h2gemu_align import set_options
        set_options(e, 0, options)
    n = len(reads1)
    codes = np.concatenate(reads1).astype(np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads1])]).astype(np.uint32)
    e.set_reads(codes, offs, quals)
    names = names or [str(i) for i in range(n)]
    nb = "".join(names).encode()
    noffs = np.concatenate([[0], np.cumsum([len(q) for q in names])]).astype(np.uint32)
    stats = np.zeros(2 + len(BAIL_REASONS) + 2, dtype=np.uint64)
    bad = np.zeros(64, dtype=np.uint32)
    done = np.zeros(n, dtype=np.uint8)
    vp = C.c_void_p
    e.L.h2gemu_fast_check.argtypes = [vp, vp, vp, C.c_char_p, vp, C.c_char_p, vp, vp, vp, C.c_uint32, vp]
    if reads2 is not None:
        c2 = np.concatenate([np.concatenate(reads2).astype(np.uint8), np.zeros(8, np.uint8)])
        o2 = np.concatenate([[0], np.cumsum([len(r) for r in reads2])]).astype(np.uint32)
        e.L.h2gemu_fast_check(e.h, c2.ctypes.data, o2.ctypes.data, nb, noffs.ctypes.data, nb, noffs.ctypes.data, stats.ctypes.data, bad.ctypes.data, 64, done.ctypes.data)
    else:
        e.L.h2gemu_fast_check(e.h, None, None, nb, noffs.ctypes.data, None, None, stats.ctypes.data, bad.ctypes.data, 64, done.ctypes.data)
    nbad = int(stats[1])
    return {"n": n, "completed": int(stats[0]), "mismatching": nbad, "bails": {BAIL_REASONS[k]: int(stats[2 + k]) for k in range(len(BAIL_REASONS)) if stats[2 + k]},
            "bad": [int(x) for x in bad[:min(nbad, 64)]], "done": done}


def pairs_overflow_check(base, reads1, reads2, slots, ovf_cap):
    """paired go() on the host machine with `slots` record rows per mate and an overflow area of ovf_cap records (MachOut::ovf)
    -> dict(in_area, mismatching, flagged, area_records)"""
    e = Emu(base)
    n = len(reads1)
    codes = np.concatenate(reads1).astype(np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads1])]).astype(np.uint32)
    e.set_reads(codes, offs, None)
    names = [str(i) for i in range(n)]
    nb = "".join(names).encode()
    noffs = np.concatenate([[0], np.cumsum([len(q) for q in names])]).astype(np.uint32)
    c2 = np.concatenate([np.concatenate(reads2).astype(np.uint8), np.zeros(8, np.uint8)])
    o2 = np.concatenate([[0], np.cumsum([len(r) for r in reads2])]).astype(np.uint32)
    stats = np.zeros(4, dtype=np.uint64)
    vp = C.c_void_p
    e.L.h2gemu_pairs_overflow_check.argtypes = [vp, vp, vp, C.c_char_p, vp, C.c_char_p, vp, C.c_uint32, C.c_uint32, vp]
    e.L.h2gemu_pairs_overflow_check(e.h, c2.ctypes.data, o2.ctypes.data, nb, noffs.ctypes.data, nb, noffs.ctypes.data, slots, ovf_cap, stats.ctypes.data)
    return {"n": n, "in_area": int(stats[0]), "mismatching": int(stats[1]), "flagged": int(stats[2]), "area_records": int(stats[3])}


if __name__ == "__main__":
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import bench
    from hisat2_amd import synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    mode = sys.argv[2] if len(sys.argv) > 2 else "pairs"
    base, contigs = bench.build_index(os.path.join(ROOT, ".bench_cache"), 4_900_000)
    if mode == "pairs":
        m1, m2 = synth.make_pairs(contigs, n, 101, bench.SEED + 2000, frag_mean=300, frag_sd=30, sub_rate=0.005)
        r = fast_check(base, m1, m2)
    else:
        reads, _ = synth.make_reads(contigs, n, 101, bench.SEED + 1000, sub_rate=0.005)
        r = fast_check(base, reads)
    r.pop("done")
    print(r)
