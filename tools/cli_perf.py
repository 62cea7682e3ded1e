#!/usr/bin/env python3
"""command-line throughput on a multi-batch input: cli_perf.py npairs [exe ...]  (E. coli-size cached index, --no-spliced-alignment;
H2G_CLI_GENOME=<bases> takes the staged / built GRCh38-profile index of that size instead)"""
import os, sys, time, subprocess, tempfile, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import bench
from hisat2_amd import synth
n = int(sys.argv[1])
exes = sys.argv[2:] or [os.path.join(ROOT, "hisat2_amd", "hisat2-align-amd")]
if os.environ.get("H2G_CLI_GENOME"):
    import build_bench_index as BB
    base, total, how = bench.headline_index(os.path.join(ROOT, ".bench_cache"), int(float(os.environ["H2G_CLI_GENOME"])))
    contigs = BB.genome(total)
else:
    base, contigs = bench.build_index(os.path.join(ROOT, ".bench_cache"), 4_900_000)
t0 = time.time()
m1, m2 = synth.make_pairs(contigs, n, 101, bench.SEED + 7, sub_rate=0.005)
tmp = tempfile.mkdtemp(prefix="h2cli")
f1, f2 = os.path.join(tmp, "r1.fa"), os.path.join(tmp, "r2.fa")
synth.write_reads_fasta(f1, m1); synth.write_reads_fasta(f2, m2)
print("input ready in %.1f s" % (time.time() - t0), flush=True)
def body_md5(path):
    h = hashlib.md5()
    for l in open(path, "rb"):
        if not l.startswith(b"@"):
            h.update(l)
    return h.hexdigest()


variants = [(os.path.basename(e), e, {}, True) for e in exes]
for name, exe, env, check in variants:
    for dest in (os.path.join(tmp, "o.sam"), "/dev/null"):
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-f", "--no-spliced-alignment", "-p", "32", "-x", base, "-1", f1, "-2", f2, "-S", dest], env=dict(os.environ, H2G_CLI_TIMING="1", **env), capture_output=True, text=True)
        dt = time.perf_counter() - t0
        h = body_md5(dest) if (check and dest != "/dev/null") else ""
        print("%-22s -> %-9s rc %d wall %.2f s = %.2f M reads/s %s | %s" % (name, "file" if dest != "/dev/null" else dest, r.returncode, dt, 2 * n / dt / 1e6, h, [l for l in r.stderr.splitlines() if l.startswith("time:") or l.startswith("index load:")]), flush=True)
