#!/usr/bin/env python3
"""Development aid: run chosen reads of a fuzz case (tmp dir of fuzz_align/fuzz_pairs) through a -DH2G_TRACE build of the
host instantiation.  usage: dbg_read.py <tmpdir> <read index> [...]"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
so = os.path.join(HERE, "emul", "libh2gemu_trace.so")
subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-DH2G_TRACE", "-w", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "emul", "h2g_emul.cpp")], check=True)
import ctypes as C
import h2gemu_py
_orig = C.CDLL
C.CDLL = lambda p, *a, **k: _orig(so if p.endswith("libh2gemu.so") else p, *a, **k)
import numpy as np
import sam_util as SU
from h2gemu_align import emu_align
tmp = sys.argv[1]
idx = [int(x) for x in sys.argv[2:]]
seqs = {}
name = None
for ln in open(os.path.join(tmp, "r.fa")):
    if ln[0] == ">": name = ln[1:].strip()
    else: seqs[name] = ln.strip()
code = {"A": 0, "C": 1, "G": 2, "T": 3, "N": 4}
refnames, want = SU.parse_sam(os.path.join(tmp, "ref.sam"))
for i in idx:
    s = seqs[str(i)]
    rd = np.array([code[c] for c in s], dtype=np.uint8)
    sys.stderr.write(f"==== read {i} {s}\n"); sys.stderr.flush()
    sites = None
    if os.path.exists(os.path.join(tmp, "ss.txt")):   # fuzz_spliced with known splice sites
        from hisat2_amd import api
        sites = api.read_splice_site_file(os.path.join(tmp, "ss.txt"), refnames)
    outs, recs = emu_align(os.path.join(tmp, "g"), [rd], [str(i)], bowtie2_dp=int(os.environ.get("DP", "0")), no_spliced=int(os.environ.get("NOSPLICED", "1")), splice_sites=sites)
    got = SU.render(outs, recs, refnames, [rd], [str(i)])
    print(" GOT ", got[str(i)], "\n WANT", want[str(i)])
