#!/usr/bin/env python3
"""Memory-access profile of the go() machine on the bench workload (host instantiation, default device workspace layout).
usage: run.py [nreads] [pairs]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from hisat2_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
paired = len(sys.argv) > 2 and sys.argv[2] == "pairs"
subprocess.run(["make", "-s", "-C", HERE], check=True)
L = C.CDLL(os.path.join(HERE, "libh2gemu_memprof.so"))
base, contigs = bench.build_index(os.path.join(ROOT, ".bench_cache"), 4_900_000)
vp = C.c_void_p
h = vp()
L.h2gemu_load.argtypes = [C.c_char_p, C.POINTER(vp)]
assert L.h2gemu_load(base.encode(), C.byref(h)) == 0
L.h2gemu_set_reads.argtypes = [vp, vp, vp, vp, C.c_size_t]
names = [str(i) for i in range(n)]
nb = "".join(names).encode(); noffs = np.concatenate([[0], np.cumsum([len(q) for q in names])]).astype(np.uint32)
outs = np.zeros(n * 4096, dtype=np.uint8); recs = np.zeros(n * 32 * 512, dtype=np.uint8)
if not paired:
    reads, _ = synth.make_reads(contigs, n, 101, bench.SEED + 1000, sub_rate=0.005)
    codes, offs = synth.flatten_reads(reads)
    L.h2gemu_set_reads(h, codes.ctypes.data, offs.ctypes.data, None, n)
    L.h2gemu_align.argtypes = [vp, C.c_uint32, C.c_char_p, vp, vp, vp]
    L.h2gemu_align(h, 1, nb, noffs.ctypes.data, outs.ctypes.data, recs.ctypes.data)
else:
    m1, m2 = synth.make_pairs(contigs, n, 101, bench.SEED + 2000, frag_mean=300, frag_sd=30, sub_rate=0.005)
    c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
    L.h2gemu_set_reads(h, c1.ctypes.data, o1.ctypes.data, None, n)
    recs2 = np.zeros(n * 32 * 512, dtype=np.uint8)
    L.h2gemu_align_pairs.argtypes = [vp, C.c_uint32, vp, vp, C.c_char_p, vp, C.c_char_p, vp, vp, vp, vp]
    L.h2gemu_align_pairs(h, 1, c2.ctypes.data, o2.ctypes.data, nb, noffs.ctypes.data, nb, noffs.ctypes.data, outs.ctypes.data, recs.ctypes.data, recs2.ctypes.data)
L.h2gemu_memprof_report.argtypes = [C.c_uint32]
sys.stdout.flush()
L.h2gemu_memprof_report(n)
L.mach_pctrace_close()
