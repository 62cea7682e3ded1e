#!/usr/bin/env python3
"""Large GPU fuzz (development aid): go() through the C ABI vs the reference binary on fresh genomes — linear and SNP-graph
indexes, single-end and paired-end.  usage: gpu_fuzz.py [scale]"""
import os, sys, functools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_align as FA
import fuzz_pairs as FP
import test_gpu_align as TA
import test_gpu_pairs as TP
from hisat2_amd import api
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tot = 0
for case in [dict(seed=901, nreads=60000 * scale, rdlen=101, sub=0.01, indel=0.001, nrate=0.001),
             dict(seed=902, nreads=40000 * scale, rdlen=101, sub=0.01, indel=0.001, nrate=0.001, snps=250),
             dict(seed=903, nreads=30000 * scale, rdlen=101, sub=0.02, indel=0.003, nrate=0.0, snps=80),
             dict(seed=904, nreads=20000 * scale, rdlen=150, sub=0.01, indel=0.002, nrate=0.001, snps=150, lens=(400000, 150000), repeats=40),
             dict(seed=905, nreads=20000 * scale, rdlen=101, sub=0.02, indel=0.006, nrate=0.001, extra=("--bowtie2-dp", "2"), bowtie2_dp=2),
             dict(seed=921, nreads=40000 * scale, rdlen=101, sub=0.02, indel=0.004, nrate=0.002, snps=40),
             dict(seed=922, nreads=40000 * scale, rdlen=151, sub=0.015, indel=0.003, nrate=0.001, snps=60, lens=(500000, 200000), repeats=60),
             dict(seed=924, nreads=20000 * scale, rdlen=250, sub=0.01, indel=0.002, nrate=0.001, snps=100),
             dict(seed=928, nreads=20000 * scale, rdlen=150, sub=0.02, indel=0.006, nrate=0.002, snps=30, extra=("--bowtie2-dp", "2"), bowtie2_dp=2)]:
    dp = case.pop("bowtie2_dp", 0)
    bad, _ = FA.run_case(verbose=2, backend=functools.partial(TA._backend, bowtie2_dp=dp), **case)
    tot += bad
for snps, case in [(0, dict(seed=911, npairs=40000 * scale, rdlen=101, sub=0.01)), (200, dict(seed=912, npairs=30000 * scale, rdlen=101, sub=0.01)),
                   (100, dict(seed=913, npairs=20000 * scale, rdlen=125, sub=0.02, frag_mean=350, frag_sd=120)),
                   (40, dict(seed=931, npairs=30000 * scale, rdlen=101, sub=0.02)),
                   (60, dict(seed=932, npairs=20000 * scale, rdlen=150, sub=0.015, frag_mean=400, frag_sd=150))]:
    FP.SNPS = snps
    bad, _ = FP.run_case(verbose=2, backend=TP._backend, stride=api.PAIR_RES_CAP, **case)
    tot += bad
print("TOTAL mismatching:", tot)
