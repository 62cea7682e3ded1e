#!/usr/bin/env python3
"""Development fuzzer (build container only): host instantiation of the go() state machine vs the real reference
binary on freshly generated genomes/reads.  usage: fuzz_align.py <seed> <nreads> <rdlen> <sub> <indel> <nrate> [genome spec]"""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import sam_util as SU  # noqa: E402
from h2gemu_align import emu_align  # noqa: E402
from hisat2_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def run_case(seed, nreads, rdlen, sub, indel, nrate, lens=(300000, 120000, 60000), repeats=6, gaps=2, verbose=8, extra=(), backend=None, bowtie2_dp=0, snps=0, fastq=False):
    tmp = tempfile.mkdtemp(prefix="h2fuzz")
    contigs = synth.make_genome(list(lens), seed, n_gaps=gaps, gap_len=300, repeats=repeats, repeat_len=500)
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs)
    base = os.path.join(tmp, "g")
    src = contigs
    if snps:   # graph index: seeded variants every ~`snps` bp, reads drawn from the alternate haplotype
        var = synth.make_snps(contigs, seed + 5, every=snps)
        synth.write_snps(os.path.join(tmp, "g.snp"), var)
        subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", "--snp", os.path.join(tmp, "g.snp"), fa, base], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        src = synth.apply_snps(contigs, var)
    else:
        subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads, _ = synth.make_reads(src, nreads, rdlen, seed + 1, sub_rate=sub, indel_rate=indel, n_rate=nrate)
    rfa = os.path.join(tmp, "r.fa")
    synth.write_reads_fasta(rfa, reads)
    quals = None
    if fastq:   # FASTQ input with seeded random qualities: the mismatch / N penalties become quality-dependent (scoring.h:206-260)
        import numpy as np
        rng = np.random.default_rng(seed + 9)
        quals = (33 + rng.choice(np.array([2, 8, 15, 20, 25, 30, 37, 40], dtype=np.uint8), size=reads.shape)).astype(np.uint8)
        rfa = os.path.join(tmp, "r.fq")
        with open(rfa, "wb") as f:
            txt = synth._ALPHA[reads]
            for i in range(nreads):
                f.write(b"@%d\n" % i + txt[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n")
    sam = os.path.join(tmp, "ref.sam")
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-q" if fastq else "-f", "-p", "1", "--no-spliced-alignment", "-x", base, "-U", rfa, "-S", sam] + list(extra),
                   check=True, stdout=subprocess.DEVNULL, stderr=open(os.path.join(tmp, "ref.err"), "w"))
    refnames, want = SU.parse_sam(sam)
    qnames = [str(i) for i in range(nreads)]
    if backend is None:
        outs, recs = emu_align(base, [reads[i] for i in range(nreads)], qnames, bowtie2_dp=bowtie2_dp, quals=None if quals is None else quals.reshape(-1),
                               options=list(extra))
        got = SU.render(outs, recs, refnames, [reads[i] for i in range(nreads)], qnames)
    else:   # backend(base, reads, qnames) -> (outs with .overflow/.depth, rendered dict)
        kw = {}
        if quals is not None:
            kw["quals"] = quals.reshape(-1)
        opts = list(extra)
        if opts:
            kw["options"] = opts
        outs, got = backend(base, reads, qnames, refnames, **kw)
    bad = ovf = setbad = 0
    maxdep = 0
    for i, q in enumerate(qnames):
        ovf += 1 if outs[i].overflow else 0
        ovfbits = ovfbits | outs[i].overflow if 'ovfbits' in dir() else outs[i].overflow
        maxdep = max(maxdep, outs[i].depth)
        if got[q] != want[q]:
            bad += 1
            if sorted((f & ~256, r, p, c, a) for f, r, p, c, a in got[q]) != sorted((f & ~256, r, p, c, a) for f, r, p, c, a in want[q]):
                setbad += 1
            if bad <= verbose:
                print(" read", q, ("ovf%d" % outs[i].overflow) if outs[i].overflow else "", "\n   GOT ", got[q], "\n   WANT", want[q])
    naln = sum(1 for q in qnames if want[q][0][0] != 4)
    print(f"seed {seed} n {nreads} len {rdlen} sub {sub} indel {indel} N {nrate}: aligned(ref) {naln}  mismatching {bad} (set-level {setbad})  overflow {ovf}  max depth {maxdep}  tmp {tmp}")
    return bad, tmp


if __name__ == "__main__":
    a = sys.argv[1:]
    dp = int(a[6]) if len(a) > 6 else 0
    snps = int(a[7]) if len(a) > 7 else 0
    run_case(int(a[0]), int(a[1]), int(a[2]), float(a[3]), float(a[4]), float(a[5]), extra=(("--bowtie2-dp", str(dp)) if dp else ()), bowtie2_dp=dp,
             snps=snps)
