#!/usr/bin/env python3
"""Parity of the bench's own GRCh38-size workload WITHOUT a GPU: pairs [first, first + n) of rank 0's bench batch (bench.py: synth.make_pairs over the
seeded 3.1 Gbp genome, seed SEED + 7) through the host instantiation of the device sources (tests/emul) and through oracle/_ref/hisat2-align-s on the
staged .bench_cache index, compared pair by pair (FLAG, RNAME, POS, CIGAR, AS:i of every line, in order); then the same pairs through the fast pass and the general machine
(tests/fast_check.py: bit for bit).  Needs the staged index (build_bench_index.py)
and ~15 GB of memory.  usage: grch38_parity_cpu.py [n=20000] [first=0] [genome=3.1e9] [sub_rate=0.005: another rate draws another batch, seed SEED + 8]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import build_bench_index as BB
import bench
import fuzz_pairs as F
import pe_sink as PS
import sam_util as SU
from hisat2_amd import synth


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    total = int(float(sys.argv[3])) if len(sys.argv) > 3 else 3_100_000_000
    base = BB.index_base(total)
    assert BB.have(base), "stage the index first: tools/build_bench_index.py"
    t0 = time.time()
    contigs = BB.genome(total)
    sub = float(sys.argv[4]) if len(sys.argv) > 4 else 0.005
    m1, m2 = synth.make_pairs(contigs, 1_000_000, 101, bench.SEED + (7 if sub == 0.005 else 8), sub_rate=sub)   # 0.005: exactly rank 0's batch of the default run (seed SEED + 7 + 1000 * rank)
    m1, m2 = m1[first:first + n], m2[first:first + n]
    del contigs
    print("reads ready %.0f s" % (time.time() - t0), flush=True)
    tmp = tempfile.mkdtemp(prefix="h2g38")
    f1, f2 = os.path.join(tmp, "1.fa"), os.path.join(tmp, "2.fa")
    synth.write_reads_fasta(f1, m1, start_id=first); synth.write_reads_fasta(f2, m2, start_id=first)
    sam = os.path.join(tmp, "ref.sam")
    opts = os.environ.get("H2G_PARITY_OPTS", "").split()          # extra reference options for both sides, e.g. "-k 10 --secondary"; --bowtie2-dp N is taken out
    dp = 0
    if "--bowtie2-dp" in opts:
        k = opts.index("--bowtie2-dp"); dp = int(opts[k + 1]); del opts[k:k + 2]
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s"), "-f", "-p", "8", "--reorder", "--no-spliced-alignment", "-x", base, "-1", f1, "-2", f2, "-S", sam]
                   + opts + (["--bowtie2-dp", str(dp)] if dp else []), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    refnames, want = F.parse_pe_sam(sam)
    print("reference done %.0f s" % (time.time() - t0), flush=True)
    q = [str(first + i) for i in range(n)]
    F.OPTS, F.DP = tuple(opts), dp
    outs, r1, r2 = F.emu_pairs(base, m1, m2, q, q)
    print("emulator done %.0f s" % (time.time() - t0), flush=True)
    bad = ovf = ncon = 0
    khits = int(opts[opts.index("-k") + 1]) if "-k" in opts else 5          # (--sensitive raises a -k below 10 and leaves the linear default of 5 alone: hisat2.cpp:1891-1907, :3903)
    if "--sensitive" in opts and "-k" in opts and khits < 10:
        khits = 10
    for i in range(n):
        got = PS.finish_pair(outs[i], r1, r2, i * SU.AL_MAX_RESULTS, refnames, (m1[i], m2[i]), khits=khits, secondary="--secondary" in opts)
        w = want[q[i]]
        ncon += 1 if (w[0][0] & 2) else 0
        ovf += 1 if outs[i].overflow else 0
        if got != w:
            bad += 1
            if bad <= 5:
                print(" pair", q[i], "ovf%d" % outs[i].overflow, "\n   GOT ", got, "\n   WANT", w)
    res = {"genome": total, "options": " ".join(opts) + (" --bowtie2-dp %d" % dp if dp else ""), "sub_rate": sub, "pairs": n, "first": first, "concordant_in_reference": ncon, "pairs_differing": bad, "flagged_overflow": ovf}
    del outs, r1, r2
    # the fast pass against the general machine on the same pairs (both on the host): what the pass completes must be the machine's result bit for bit
    import fast_check as FC
    fc = FC.fast_check(base, [m1[i] for i in range(n)], [m2[i] for i in range(n)], names=q, options=("--no-spliced-alignment",) + tuple(opts)) if not dp and "--secondary" not in opts else {"completed": 0, "mismatching": 0, "bails": {"(go_run does not route --bowtie2-dp / --secondary runs through the fast pass)": n}}
    res["fast_pass"] = {"completed": fc["completed"], "mismatching_the_machine": fc["mismatching"], "handed_on": fc["bails"]}
    res["seconds"] = round(time.time() - t0)
    print(res)
    return 1 if bad or fc["mismatching"] else 0


if __name__ == "__main__":
    sys.exit(main())
