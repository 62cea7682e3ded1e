#!/usr/bin/env python3
"""Development aid (build container only): primitive-level comparison on a fuzz case — runs ref_probe (psearch / coords /
extend / adjust) on the first N reads of a fuzz_align tmp dir and checks the host instantiation of the device functions
against it with the parity_cases checkers.  usage: probe_fuzz.py <tmpdir> <nreads> [cmd ...]"""
import gzip, os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import parity_cases as PC
from h2gemu_py import Emu
tmp, n = sys.argv[1], int(sys.argv[2])
cmds = sys.argv[3:] or ["psearch", "coords", "extend", "adjust"]
code = {"A": 0, "C": 1, "G": 2, "T": 3, "N": 4}
sub = os.path.join(tmp, "sub.fa")
seqs = []
with open(os.path.join(tmp, "r.fa")) as f, open(sub, "w") as o:
    for i, ln in enumerate(f):
        if i >= 2 * n: break
        o.write(ln)
        if ln[0] != ">": seqs.append(np.array([code[c] for c in ln.strip()], dtype=np.uint8))
gd = tempfile.mkdtemp(prefix="h2probe")
probe = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "ref_probe")
for c in cmds:
    out = subprocess.run([probe, c, os.path.join(tmp, "g"), sub, "1"], check=True, capture_output=True, text=True).stdout
    with gzip.open(os.path.join(gd, f"probe_g1s_{c}.txt.gz"), "wt") as f: f.write(out)
e = Emu(os.path.join(tmp, "g"))
codes = np.concatenate(seqs); offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint32)
e.set_reads(codes, offs)
for c in cmds:
    try:
        if c == "psearch": print(c, PC.check_graph_fm_search(e, gd))
        if c == "coords": print(c, PC.check_graph_coords(e, gd, "probe_g1s_coords.txt.gz"))
        if c == "extend": print(c, PC.check_graph_extend(e, gd))
        if c == "adjust": print(c, PC.check_graph_adjust(e, gd, "probe_g1s_adjust.txt.gz"))
    except AssertionError as ex:
        print(c, "MISMATCH", str(ex)[:1500])
print("probe outputs in", gd)
