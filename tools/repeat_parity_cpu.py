#!/usr/bin/env python3
"""The bench's repeat_pe batch (bench.py repeat_leg: synth.make_repeat_genome over 256 Mbp, linear index, pairs seeded SEED + 78) WITHOUT a GPU: pairs
[first, first + n) through the host instantiation of the device sources and through oracle/_ref/hisat2-align-s (the index is built into .bench_cache if it is
not staged), pair by pair; then fast pass against general machine, bit for bit.  usage: repeat_parity_cpu.py [n=20000] [first=0] [genome=256e6]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import build_bench_index as BB
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
    d = os.path.join(ROOT, ".bench_cache", f"rep{glen}_s{bench.SEED}")
    base = os.path.join(d, "g")
    t0 = time.time()
    contigs = synth.make_repeat_genome(BB.contig_lens(glen), bench.SEED + 77)
    if not os.path.exists(base + ".8.ht2"):
        os.makedirs(d, exist_ok=True)
        synth.write_fasta(base + ".fa", contigs)
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "hisat2-build-s"), "-q", "-p", str(BB.usable_cpus()), base + ".fa", base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.remove(base + ".fa")
        print("index built %.0f s" % (time.time() - t0), flush=True)
    m1, m2 = synth.make_pairs(contigs, 1_000_000, 101, bench.SEED + 78, sub_rate=0.005)     # the leg's batch
    m1, m2 = m1[first:first + n], m2[first:first + n]
    del contigs
    tmp = tempfile.mkdtemp(prefix="h2rep")
    f1, f2 = os.path.join(tmp, "1.fa"), os.path.join(tmp, "2.fa")
    synth.write_reads_fasta(f1, m1, start_id=first); synth.write_reads_fasta(f2, m2, start_id=first)
    sam = os.path.join(tmp, "ref.sam")
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s"), "-f", "-p", "8", "--reorder", "--no-spliced-alignment", "-x", base, "-1", f1, "-2", f2, "-S", sam],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    refnames, want = F.parse_pe_sam(sam)
    print("reference done %.0f s" % (time.time() - t0), flush=True)
    q = [str(first + i) for i in range(n)]
    outs, r1, r2 = F.emu_pairs(base, m1, m2, q, q)
    print("emulator done %.0f s" % (time.time() - t0), flush=True)
    bad = ovf = ncon = 0
    for i in range(n):
        got = PS.finish_pair(outs[i], r1, r2, i * SU.AL_MAX_RESULTS, refnames, (m1[i], m2[i]), khits=5, secondary=False)
        w = want[q[i]]
        ncon += 1 if (w[0][0] & 2) else 0
        ovf += 1 if outs[i].overflow else 0
        if got != w:
            bad += 1
            if bad <= 5:
                print(" pair", q[i], "ovf%d" % outs[i].overflow, "\n   GOT ", got, "\n   WANT", w)
    res = {"genome": glen, "pairs": n, "first": first, "concordant_in_reference": ncon, "pairs_differing": bad, "flagged_overflow": ovf}
    del outs, r1, r2
    fc = FC.fast_check(base, [m1[i] for i in range(n)], [m2[i] for i in range(n)], names=q, options=("--no-spliced-alignment",))
    res["fast_pass"] = {"completed": fc["completed"], "mismatching_the_machine": fc["mismatching"], "handed_on": fc["bails"]}
    res["seconds"] = round(time.time() - t0)
    print(res)
    return 1 if bad or fc["mismatching"] else 0


if __name__ == "__main__":
    sys.exit(main())
