#!/usr/bin/env python3
"""Random option sets against the live reference on the host instantiation of this round's device sources (the emulator, default configuration): single-end and paired cases on fresh
genomes, linear and SNP-graph indexes.  usage: option_campaign.py [cases per kind] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzz_align as FA
import fuzz_pairs as FP

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9100
tot = {"se_cases": 0, "se_bad": 0, "pe_cases": 0, "pe_bad": 0}
for k in range(n):
    rng = np.random.default_rng(s0 + k)
    extra = []
    if rng.random() < 0.5: extra += ["-k", str(int(rng.integers(1, 9)))]
    if rng.random() < 0.3: extra += ["--secondary"]
    if rng.random() < 0.4: extra += ["--score-min", "L,0,%g" % float(rng.choice([-0.2, -0.4, -0.6, -1.0]))]
    if rng.random() < 0.3: extra += ["--mp", "%d,%d" % (int(rng.integers(3, 7)), int(rng.integers(1, 3)))]
    if rng.random() < 0.3: extra += ["--rdg", "%d,%d" % (int(rng.integers(3, 8)), int(rng.integers(1, 4)))]
    if rng.random() < 0.2: extra += ["--no-softclip"]
    dp = int(rng.choice([0, 0, 0, 1, 2]))
    if dp: extra += ["--bowtie2-dp", str(dp)]
    snps = int(rng.choice([0, 0, 150]))
    rdlen = int(rng.choice([50, 76, 101, 125, 150]))
    bad, _ = FA.run_case(s0 + k, 1500, rdlen, float(rng.choice([0.005, 0.02, 0.04])), float(rng.choice([0.0, 0.002, 0.006])), 0.001, extra=tuple(extra), verbose=1, bowtie2_dp=dp, snps=snps,
                         fastq=bool(rng.random() < 0.4))
    tot["se_cases"] += 1; tot["se_bad"] += bad
for k in range(n):
    rng = np.random.default_rng(s0 + 500 + k)
    opts = []
    if rng.random() < 0.4: opts += ["-k", str(int(rng.integers(1, 7)))]
    if rng.random() < 0.3: opts += ["-X", str(int(rng.choice([400, 600, 1500])))]
    if rng.random() < 0.3: opts += ["--score-min", "L,0,%g" % float(rng.choice([-0.2, -0.5, -1.0]))]
    if rng.random() < 0.2: opts += ["--no-mixed"] if False else []
    FP.OPTS = tuple(opts)
    FP.SNPS = int(rng.choice([0, 0, 200]))
    bad, _ = FP.run_case(verbose=1, seed=s0 + 500 + k, npairs=1200, rdlen=int(rng.choice([76, 101, 125])), sub=float(rng.choice([0.005, 0.02, 0.04])))
    tot["pe_cases"] += 1; tot["pe_bad"] += bad
print("total:", tot)
