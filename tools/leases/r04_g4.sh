#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD:$PWD/tests
python - <<'PY'
import os, sys, tempfile, subprocess, json
sys.path.insert(0, "tests")
import numpy as np
from hisat2_amd import synth
import fast_stress as FS
tmp = tempfile.mkdtemp(prefix="h2fg")
contigs = synth.make_genome([1500000, 400000, 100000], 73, n_gaps=3, gap_len=300, repeats=80, repeat_len=600)
var = synth.make_snps(contigs, 82, every=250)
fa = os.path.join(tmp, "g.fa"); synth.write_fasta(fa, contigs); synth.write_snps(os.path.join(tmp, "g.snp"), var)
base = os.path.join(tmp, "g")
subprocess.run(["oracle/_ref/hisat2-build-s", "-q", "-p", "16", "--snp", os.path.join(tmp, "g.snp"), fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
alt = synth.apply_snps(contigs, var)
m1, m2 = synth.make_pairs(alt, 100000, 101, 74, frag_mean=300, frag_sd=40, sub_rate=0.005)
reads, _ = synth.make_reads(alt, 100000, 101, 75, sub_rate=0.005)
npz = os.path.join(tmp, "reads.npz")
np.savez(npz, m1=np.stack(m1), m2=np.stack(m2), reads=np.asarray(reads))
r = FS.run(base, npz, runs=1, sizes=("all",))
for c in r["cases"]:
    print(c["kind"], c["n"], "fast", c["fast"], "handed_on", c["handed_on"], "differing", c["differing"])
    for f in c["first"]: print("   ", json.dumps(f, default=str)[:1500])
PY
