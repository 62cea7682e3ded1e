#!/usr/bin/env python3
"""host-side SAM formatting rate (no GPU): pairs through the host instantiation of go() once, then h2g_sam_format_paired timed.
usage: sam_fmt_perf.py [npairs] [threads]"""
import ctypes as C, os, sys, time, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzz_pairs as F, sam_lines as SL, sam_util as SU
from hisat2_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tmp = tempfile.mkdtemp(prefix="h2fmt")
contigs = synth.make_genome([300000, 120000, 60000], 5, n_gaps=2, gap_len=300, repeats=6, repeat_len=500)
synth.write_fasta(os.path.join(tmp, "g.fa"), contigs)
base = os.path.join(tmp, "g")
subprocess.run([os.path.join(ROOT, "oracle", "_ref", "hisat2-build-s"), "-q", os.path.join(tmp, "g.fa"), base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
m1, m2 = synth.make_pairs(contigs, n, 101, 6, frag_mean=300, frag_sd=30, sub_rate=0.005)
q = [str(i) for i in range(n)]
outs, r1, r2 = F.emu_pairs(base, m1, m2, q, q)
res = (api.PairResult * n)(); a1 = (api.AlnRes * (n * api.PAIR_RES_CAP))(); a2 = (api.AlnRes * (n * api.PAIR_RES_CAP))()
C.memmove(res, outs, C.sizeof(res))
for i in range(n):
    for m, (src, dst) in enumerate(((r1, a1), (r2, a2))):
        for k in range(min(outs[i].nres[m], api.PAIR_RES_CAP)):
            C.memmove(C.byref(dst[i * api.PAIR_RES_CAP + k]), C.byref(src[i * SU.AL_MAX_RESULTS + k]), C.sizeof(api.AlnRes))
L = SL.load_sam_lib()
L.h2g_sam_set_threads.argtypes = [C.c_void_p, C.c_int]
h = C.c_void_p(); assert L.h2g_sam_open(base.encode(), C.byref(h)) == 0
L.h2g_sam_set_threads(h, threads)
c1, o1 = SL.flat(list(m1)); c2, o2 = SL.flat(list(m2)); nb, no = SL.flat_names(q)
cap = 1400 * n + 4096; buf = C.create_string_buffer(cap); used = C.c_size_t(0)
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    rc = L.h2g_sam_format_paired(h, c1.ctypes.data, o1.ctypes.data, None, nb, no.ctypes.data, c2.ctypes.data, o2.ctypes.data, None, nb, no.ctypes.data, n, res, a1, a2, 5, buf, cap, C.byref(used))
    best = min(best, time.perf_counter() - t0)
    assert rc == 0, rc
import hashlib
print("pairs %d threads %d: %.3f s best of 5 = %.2f us per line, %d bytes, md5 %s" % (n, threads, best, best / (2 * n) * 1e6, used.value, hashlib.md5(buf.raw[:used.value]).hexdigest()))
