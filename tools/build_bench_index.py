#!/usr/bin/env python3
"""Builds the GRCh38-SCALE benchmark index of bench.py into .bench_cache/ (BASELINE configs[2] shape): a seeded
uniform-random genome of TOTAL bases in 24 human-profile contigs, indexed by the reference's own builder
(oracle/_ref/hisat2-build-s; index construction is outside the hot path).  The cache directory travels to the GPU
box with the repo snapshot, so the driver's bench run does not pay the ~20-60 min build.

usage: build_bench_index.py [TOTAL=3.1e9] [threads]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from hisat2_amd import synth

SEED = 20260925 + 38
PROFILE = [248, 242, 198, 190, 181, 171, 159, 145, 138, 133, 135, 133, 114, 107, 102, 90, 83, 80, 58, 64, 46, 50, 156, 57]


def contig_lens(total):
    prof = np.array(PROFILE, dtype=np.float64)
    return [int(x) for x in (prof / prof.sum() * total).astype(np.int64)]


def index_base(total, cache=None):
    cache = cache or os.path.join(ROOT, ".bench_cache")
    return os.path.join(cache, f"grch38sim{total}_s{SEED}")


def genome(total):
    return synth.make_genome(contig_lens(total), SEED)


def have(base):
    return all(os.path.exists(f"{base}.{k}.ht2") for k in range(1, 9))


def usable_cpus():
    """CPUs this process can actually run on: the affinity mask capped by the cgroup CPU quota.  The reference's builder slows down when it
    is given more threads than that (200 Mbp on 8 CPUs: 61 s with -p 8, 115 s with -p 32), and the GPU boxes show 256 CPUs under a 16-CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 64))


def build(total, threads=None, timeout=None, cache=None):
    base = index_base(total, cache)
    if have(base):
        return base
    os.makedirs(os.path.dirname(base), exist_ok=True)
    builder = os.path.join(ROOT, "oracle", "_ref", "hisat2-build-s")
    if not os.path.exists(builder):
        raise RuntimeError("oracle/_ref/hisat2-build-s is missing")
    fa = base + ".fa"
    synth.write_fasta(fa, genome(total))
    threads = threads or usable_cpus()
    try:
        subprocess.run([builder, "-q", "-p", str(threads), fa, base + ".tmp"], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=timeout)
        for k in range(1, 9):
            os.replace(f"{base}.tmp.{k}.ht2", f"{base}.{k}.ht2")
    finally:
        for p in [fa] + [f"{base}.tmp.{k}.ht2" for k in range(1, 9)]:
            if os.path.exists(p):
                os.remove(p)
    return base


if __name__ == "__main__":
    total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
    thr = int(sys.argv[2]) if len(sys.argv) > 2 else None
    t0 = time.time()
    b = build(total, thr)
    print(b, "built in %.0f s" % (time.time() - t0), {k: os.path.getsize(f"{b}.{k}.ht2") for k in range(1, 9)})
