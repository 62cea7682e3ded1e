#!/usr/bin/env python3
"""--haplotype (development fuzzer, build container only): a SNP-graph index built with --haplotype; reads come from donor
genomes that carry, per SNP cluster, either one of the index's haplotypes or an arbitrary subset of the cluster's SNPs.  With
--haplotype the reference only walks ALT combinations some haplotype carries (alignWithALTs_recur hi_aligner.h:2898-2996,
:3251-3331); the host instantiation of the same device source must agree read for read.
usage: fuzz_haplotype.py <seed> <nreads> [every] [spliced 0/1]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import sam_util as SU  # noqa: E402
from h2gemu_align import emu_align  # noqa: E402
from hisat2_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def make_haplotypes(snps, rng, window=70, with_file=True):
    """clusters of SNPs less than `window` apart -> (haplotype lines, clusters); a haplotype = a non-empty subset of a cluster"""
    clusters, cur = [], []
    for s in snps:
        if cur and (s[2] != cur[-1][2] or s[3] - cur[-1][3] >= window):
            clusters.append(cur); cur = []
        cur.append(s)
    if cur:
        clusters.append(cur)
    lines, per = [], []
    for c in clusters:
        hs = []
        for _ in range(int(rng.integers(1, 4))):
            pick = [s for s in c if rng.random() < 0.6] or [c[int(rng.integers(0, len(c)))]]
            if pick not in hs:
                hs.append(pick)
        per.append(hs)
        for pick in hs:
            last = pick[-1]
            right = last[3] + (int(last[4]) - 1 if last[1] == "deletion" else 0)
            lines.append("ht%d\t%s\t%d\t%d\t%s" % (len(lines), pick[0][2], pick[0][3], right, ",".join(s[0] for s in pick)))
    return lines, clusters, per


def real_example():
    """the reference's example: 1 Mbp of chr22 and its 3.5 k dbSNP variants (tests/golden)"""
    import gzip
    import fuzz_real
    gold = os.path.join(HERE, "golden")
    snps = []
    for ln in gzip.open(os.path.join(gold, "example_22_20-21M.snp.gz"), "rt"):
        f = ln.split()
        snps.append((f[0], f[1], "chr1", int(f[3]), f[4]))
    return [fuzz_real.real_contig()], snps


def run_case(seed, nreads, every=40, rdlen=101, sub=0.004, verbose=6, extra=(), use=True, glen=200000, ht_file=True, real=False):
    tmp = tempfile.mkdtemp(prefix="h2hap")
    rng = np.random.default_rng(seed)
    if real:
        contigs, snps = real_example()
    else:
        contigs = [rng.integers(0, 4, size=glen, dtype=np.uint8), rng.integers(0, 4, size=glen // 3, dtype=np.uint8)]
        snps = synth.make_snps(contigs, seed + 5, every=every)
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs)
    synth.write_snps(os.path.join(tmp, "g.snp"), snps)
    lines, clusters, per = make_haplotypes(snps, rng)
    with open(os.path.join(tmp, "g.haplotype"), "w") as f:
        f.write("\n".join(lines) + "\n")
    base = os.path.join(tmp, "g")
    cmd = [os.path.join(REF, "hisat2-build-s"), "-q", "--snp", os.path.join(tmp, "g.snp")]
    if ht_file:                                        # without the file every SNP is a haplotype of its own (gfm.h:1645)
        cmd += ["--haplotype", os.path.join(tmp, "g.haplotype")]
    subprocess.run(cmd + [fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # donors: per cluster one of its haplotypes (consistent) / nothing / an arbitrary subset (possibly carried by no haplotype)
    donors = []
    for d in range(6):
        chosen = []
        for c, hs in zip(clusters, per):
            r = rng.random()
            if d < 3:
                chosen += hs[int(rng.integers(0, len(hs)))] if r < 0.8 else []
            else:
                chosen += [s for s in c if rng.random() < 0.5]
        donors.append(synth.apply_snps(contigs, chosen))
    per_d = nreads // len(donors)
    parts = [synth.make_reads(dn, per_d if k else nreads - per_d * (len(donors) - 1), rdlen, seed + 11 + k, sub_rate=sub, indel_rate=0.0, n_rate=0.0)[0] for k, dn in enumerate(donors)]
    reads = np.concatenate(parts)
    rfa = os.path.join(tmp, "r.fa")
    synth.write_reads_fasta(rfa, reads)
    opts = (["--haplotype"] if use else []) + list(extra)
    spliced = "--no-spliced-alignment" not in opts and "--spliced" in opts
    opts = [o for o in opts if o != "--spliced"]
    mode = ["--no-temp-splicesite"] if spliced else ["--no-spliced-alignment"]
    sam = os.path.join(tmp, "ref.sam")
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", "1", "-x", base, "-U", rfa, "-S", sam] + mode + opts,
                   check=True, stdout=subprocess.DEVNULL, stderr=open(os.path.join(tmp, "ref.err"), "w"))
    sam0 = os.path.join(tmp, "ref_nohap.sam")
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", "1", "-x", base, "-U", rfa, "-S", sam0] + mode + [o for o in opts if o != "--haplotype"],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    refnames, want = SU.parse_sam(sam)
    _, want0 = SU.parse_sam(sam0)
    qnames = [str(i) for i in range(len(reads))]
    rl = [reads[i] for i in range(len(reads))]
    outs, recs = emu_align(base, rl, qnames, no_spliced=0 if spliced else 1, options=opts)
    got = SU.render(outs, recs, refnames, rl, qnames)
    bad = 0
    for q in qnames:
        if got[q] != want[q]:
            bad += 1
            if bad <= verbose:
                print(" read", q, "\n   GOT ", got[q], "\n   WANT", want[q], "\n   (without --haplotype)", want0[q])
    delta = sum(1 for q in qnames if want[q] != want0[q])
    print(f"seed {seed} n {len(reads)} every {every}: haplotypes {len(lines)}  mismatching {bad}  overflow {sum(1 for o in outs if o.overflow)}  "
          f"reads --haplotype changes in the reference: {delta}  tmp {tmp}")
    return bad, tmp


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    extra = ("--spliced",) if len(sys.argv) > 4 and sys.argv[4] == "1" else ()
    sys.exit(1 if run_case(seed, n, every, extra=extra, real=bool(os.environ.get("H2G_FUZZ_REAL")))[0] else 0)
