#!/usr/bin/env python3
"""End-to-end timing of hisat2-align-amd vs hisat2-align-s on the bench genome: N x 101 bp reads, FASTA file -> SAM file.
usage: cli_timing.py [nreads=1000000] [host threads=16]"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from hisat2_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
threads = sys.argv[2] if len(sys.argv) > 2 else "16"
base, contigs = bench.build_index(os.path.join(ROOT, ".bench_cache"), 4_900_000)
reads, _ = synth.make_reads(contigs, n, 101, bench.SEED + 1000, sub_rate=0.005)
tmp = tempfile.mkdtemp(prefix="h2cli")
fa = os.path.join(tmp, "r.fa")
synth.write_reads_fasta(fa, reads)
out = {"reads": n, "host_threads": int(threads)}
cli = os.path.join(ROOT, "hisat2_amd", "hisat2-align-amd")
for rep in range(2):
    t0 = time.perf_counter()
    r = subprocess.run([cli, "-f", "--no-spliced-alignment", "-p", threads, "-x", base, "-U", fa, "-S", os.path.join(tmp, "amd.sam")], capture_output=True, text=True,
                       env=dict(os.environ, H2G_CLI_TIMING="1"))
    out["amd_wall_s"] = time.perf_counter() - t0
    out["amd_timing"] = r.stderr.strip().splitlines()[-1]
ref = os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")
t0 = time.perf_counter()
subprocess.run([ref, "-f", "--no-spliced-alignment", "--reorder", "-p", threads, "-x", base, "-U", fa, "-S", os.path.join(tmp, "ref.sam")], check=True,
               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
out["ref_wall_s"] = time.perf_counter() - t0
a = [l for l in open(os.path.join(tmp, "amd.sam")) if l[0] != "@"]
b = [l for l in open(os.path.join(tmp, "ref.sam")) if l[0] != "@"]
out["sam_lines"] = len(b)
out["sam_lines_differing"] = sum(1 for x, y in zip(a, b) if x != y) + abs(len(a) - len(b))
out["amd_reads_per_s"] = n / out["amd_wall_s"]; out["ref_reads_per_s"] = n / out["ref_wall_s"]
print(json.dumps(out))
