#!/usr/bin/env python3
"""The bench's ecoli_se batch (bench.py extras: configs[1], the E. coli-size seeded substitute, 1 M x 101 bp single-end reads, seed SEED + 1000) WITHOUT a GPU:
reads [first, first + n) through the host instantiation of the device sources and through oracle/_ref/hisat2-align-s on the staged index, read by read (FLAG, RNAME,
POS, CIGAR, AS:i of every line); then fast pass against general machine.  usage: ecoli_se_parity_cpu.py [n=1000000] [first=0]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
import sam_util as SU
import fast_check as FC
from h2gemu_align import emu_align
from hisat2_amd import synth


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    base, contigs = bench.small_index(os.path.join(ROOT, ".bench_cache"), 4_900_000)
    reads, _ = synth.make_reads(contigs, 1_000_000, 101, bench.SEED + 1000, sub_rate=0.005)      # the leg's batch
    reads = reads[first:first + n]
    tmp = tempfile.mkdtemp(prefix="h2ecoli")
    rfa = os.path.join(tmp, "r.fa")
    synth.write_reads_fasta(rfa, reads, start_id=first)
    sam = os.path.join(tmp, "ref.sam")
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s"), "-f", "-p", "8", "--reorder", "--no-spliced-alignment", "-x", base, "-U", rfa, "-S", sam],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    refnames, want = SU.parse_sam(sam)
    print("reference done %.0f s" % (time.time() - t0), flush=True)
    q = [str(first + i) for i in range(n)]
    rl = [reads[i] for i in range(n)]
    outs, recs = emu_align(base, rl, q)
    got = SU.render(outs, recs, refnames, rl, q)
    bad = sum(1 for x in q if got[x] != want[x])
    naln = sum(1 for x in q if want[x][0][0] != 4)
    ovf = sum(1 for i in range(n) if outs[i].overflow)
    res = {"reads": n, "first": first, "aligned_in_reference": naln, "reads_differing": bad, "flagged_overflow": ovf}
    del outs, recs, got
    fc = FC.fast_check(base, rl, names=q, options=("--no-spliced-alignment",))
    res["fast_pass"] = {"completed": fc["completed"], "mismatching_the_machine": fc["mismatching"], "handed_on": fc["bails"]}
    res["seconds"] = round(time.time() - t0)
    print(res)
    return 1 if bad or fc["mismatching"] else 0


if __name__ == "__main__":
    sys.exit(main())
