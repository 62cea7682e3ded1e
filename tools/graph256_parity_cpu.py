#!/usr/bin/env python3
"""The bench's graph256_pe batch (bench.py: SNP-graph index over the seeded 256 Mbp genome, a variant every ~250 bp, pairs from the alternate haplotype,
seed SEED + 79) WITHOUT a GPU: pairs [first, first + n) through the host instantiation of the device sources and through oracle/_ref/hisat2-align-s on the staged
index, pair by pair; then through the graph form of the fast pass and the general machine (libh2gemu_g.so), bit for bit.
usage: graph256_parity_cpu.py [n=20000] [first=0] [genome=256e6] [every=250]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import build_bench_index as BB, build_graph_bench_index as GB
import bench
import fuzz_pairs as F
import pe_sink as PS
import sam_util as SU
import fast_check as FC
from hisat2_amd import synth


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    glen = int(float(sys.argv[3])) if len(sys.argv) > 3 else 256_000_000
    every = int(sys.argv[4]) if len(sys.argv) > 4 else 250
    base = GB.graph_base(glen, every)
    assert all(os.path.exists(f"{base}.{k}.ht2") for k in range(1, 9)), "stage the index first: tools/build_graph_bench_index.py"
    t0 = time.time()
    contigs = BB.genome(glen)
    alt = synth.apply_snps(contigs, GB.variants(glen, every, contigs), names=GB.names(glen))
    m1, m2 = synth.make_pairs(alt, 1_000_000, 101, bench.SEED + 79, frag_mean=300, frag_sd=30, sub_rate=0.005)     # the leg's batch
    m1, m2 = m1[first:first + n], m2[first:first + n]
    del contigs, alt
    tmp = tempfile.mkdtemp(prefix="h2g256")
    f1, f2 = os.path.join(tmp, "1.fa"), os.path.join(tmp, "2.fa")
    synth.write_reads_fasta(f1, m1, start_id=first); synth.write_reads_fasta(f2, m2, start_id=first)
    sam = os.path.join(tmp, "ref.sam")
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s"), "-f", "-p", "8", "--reorder", "--no-spliced-alignment", "-x", base, "-1", f1, "-2", f2, "-S", sam],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    refnames, want = F.parse_pe_sam(sam)
    print("reference done %.0f s" % (time.time() - t0), flush=True)
    q = [str(first + i) for i in range(n)]
    F.SNPS = every                                                   # (fuzz_pairs: graph index -> -k 10)
    outs, r1, r2 = F.emu_pairs(base, m1, m2, q, q)
    print("emulator done %.0f s" % (time.time() - t0), flush=True)
    bad = ovf = ncon = 0
    for i in range(n):
        got = PS.finish_pair(outs[i], r1, r2, i * SU.AL_MAX_RESULTS, refnames, (m1[i], m2[i]), khits=10, secondary=False)
        w = want[q[i]]
        ncon += 1 if (w[0][0] & 2) else 0
        ovf += 1 if outs[i].overflow else 0
        if got != w:
            bad += 1
            if bad <= 5:
                print(" pair", q[i], "ovf%d" % outs[i].overflow, "\n   GOT ", got, "\n   WANT", w)
    res = {"genome": glen, "every": every, "pairs": n, "first": first, "concordant_in_reference": ncon, "pairs_differing": bad, "flagged_overflow": ovf}
    del outs, r1, r2
    fc = FC.fast_check(base, [m1[i] for i in range(n)], [m2[i] for i in range(n)], names=q, options=("--no-spliced-alignment",), variant="g")
    res["fast_pass"] = {"completed": fc["completed"], "mismatching_the_machine": fc["mismatching"], "handed_on": fc["bails"]}
    res["seconds"] = round(time.time() - t0)
    print(res)
    return 1 if bad or fc["mismatching"] else 0


if __name__ == "__main__":
    sys.exit(main())
