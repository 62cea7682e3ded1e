#!/usr/bin/env python3
"""Builds the SNP-GRAPH benchmark index of bench.py's graph legs into .bench_cache/ (BASELINE configs[3] shape): the seeded
uniform-random genome of build_bench_index.py at TOTAL bases, a seeded variant about every EVERY bases (86 % single-base, 7 % deletions,
7 % insertions of 1-3 bases: hisat2_amd/synth.make_snps), indexed by the reference's own builder with --snp.  Prints the build's wall
time and peak memory (the real GRCh38+SNP index needs 160-200 GB to build, MANUAL.markdown:1857; this is the stand-in that fits the box).

usage: build_graph_bench_index.py [TOTAL=256e6] [EVERY=250] [threads]"""
import os, resource, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from hisat2_amd import synth
import build_bench_index as BB


def graph_base(total, every, cache=None):
    cache = cache or os.path.join(ROOT, ".bench_cache")
    return os.path.join(cache, f"grch38sim{total}_s{BB.SEED}_snp{every}", "g")


def names(total):
    return ["chr%d" % (i + 1) for i in range(len(BB.contig_lens(total)))]


def variants(total, every, contigs=None):
    contigs = contigs if contigs is not None else BB.genome(total)
    return synth.make_snps(contigs, BB.SEED + 5, every=every, names=names(total))


def build(total, every=250, threads=None, cache=None):
    base = graph_base(total, every, cache)
    if all(os.path.exists(f"{base}.{k}.ht2") for k in range(1, 9)):
        return base, None
    os.makedirs(os.path.dirname(base), exist_ok=True)
    builder = os.path.join(ROOT, "oracle", "_ref", "hisat2-build-s")
    contigs = BB.genome(total)
    var = variants(total, every, contigs)
    synth.write_fasta(base + ".fa", contigs, names=names(total))
    synth.write_snps(base + ".snp", var)
    t0 = time.time()
    subprocess.run([builder, "-q", "-p", str(threads or BB.usable_cpus()), "--snp", base + ".snp", base + ".fa", base + ".tmp"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dt = time.time() - t0
    for k in range(1, 9):
        os.replace(f"{base}.tmp.{k}.ht2", f"{base}.{k}.ht2")
    os.remove(base + ".fa")
    peak = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1e6
    info = {"genome_bases": total, "variants": len(var), "build_seconds": dt, "builder_peak_rss_GB": peak, "threads": threads or BB.usable_cpus(),
            "index_bytes": sum(os.path.getsize(f"{base}.{k}.ht2") for k in range(1, 9))}
    import json
    json.dump(info, open(base + ".build.json", "w"))
    return base, info


if __name__ == "__main__":
    total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 256_000_000
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 250
    thr = int(sys.argv[3]) if len(sys.argv) > 3 else None
    print(build(total, every, thr))
