#!/usr/bin/env python3
"""Development fuzzer for spliced alignment (the reference's default mode, here with --no-temp-splicesite so that every read is
independent): a random genome with planted GT..AG (and some non-canonical) introns, reads drawn from the spliced "transcripts",
host instantiation of the go() machine vs the real reference binary.  usage: fuzz_spliced.py <seed> <nreads> [sub]"""
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


def rich():
    """H2G_FUZZ_RICH: three contigs (the planted genes on the second), N runs, duplicated genes, processed pseudogenes"""
    return bool(os.environ.get("H2G_FUZZ_RICH"))



def contig_names(contigs):
    """the contig with the planted introns is always called chr1 (the site files name it); RICH puts it second"""
    return ["lead", "chr1", "tail"] if len(contigs) == 3 else None


def enrich(g, introns, rng):
    """RICH: N runs inside some exons, copies of whole genes (exon-intron-exon) on a leading contig, processed pseudogenes
    (exon-exon, the intron removed) on a trailing one — spliced and contiguous placements compete for the same reads"""
    g = g.copy()
    for k in range(5, len(introns) - 1, 9):                 # 30 Ns in the middle of a long exon
        e0, e1 = introns[k][1], introns[k + 1][0]
        if e1 - e0 > 400:
            m = (e0 + e1) // 2
            g[m:m + 30] = 4
    lead = [rng.integers(0, 4, size=3000, dtype=np.uint8)]
    tail = [rng.integers(0, 4, size=3000, dtype=np.uint8)]
    for k in range(2, len(introns) - 1, 5):
        a, b = introns[k]
        if b - a <= 1200 and k % 2 == 0:
            lead += [g[a - 150:b + 150].copy(), rng.integers(0, 4, size=500, dtype=np.uint8)]
        else:
            tail += [g[a - 120:a].copy(), g[b:b + 120].copy(), rng.integers(0, 4, size=500, dtype=np.uint8)]
    lead.insert(len(lead) // 2, np.full(200, 4, dtype=np.uint8))         # an N gap: two fragments in one contig
    return [np.concatenate(lead), g, np.concatenate(tail)]


def revcomp(r):
    return np.where(r > 3, 4, 3 - r)[::-1].astype(np.uint8)


def make_case(seed, nreads, rdlen=101, sub=0.005, glen=400000, nintrons=400, alt_fn=None, indel=0.0):
    rdlen = int(os.environ.get("H2G_FUZZ_RDLEN", rdlen))
    multi = float(os.environ.get("H2G_FUZZ_MULTI", "0"))   # > 0: that fraction of the exons is 20..90 bp long and the reads are drawn from
    rng = np.random.default_rng(seed)                      # the spliced transcript, so one read crosses two or three junctions
    g = rng.integers(0, 4, size=glen, dtype=np.uint8)
    # plant introns: [a, b) with GT at a and AG at b-2 (canonical); a few GC..AG / AT..AC / random (non-canonical)
    introns = []
    pos = 2000
    while len(introns) < nintrons and pos < glen - 12000:
        L = int(rng.choice([60, 90, 150, 400, 1200, 5000, 9000]))
        a, b = pos, pos + L
        kind = int(rng.integers(0, 10))
        if kind < 7:
            g[a:a + 2] = [2, 3]; g[b - 2:b] = [0, 2]
        elif kind == 7:
            g[a:a + 2] = [2, 1]; g[b - 2:b] = [0, 2]
        elif kind == 8:
            g[a:a + 2] = [0, 3]; g[b - 2:b] = [0, 1]
        introns.append((a, b))
        pos = b + (int(rng.integers(20, 90)) if multi > 0 and rng.random() < multi else int(rng.integers(150, 900)))
    extra = None
    if rich():
        extra = enrich(g, introns, rng)
        g = extra[1]
    ref_g = g
    if alt_fn is not None:
        g = alt_fn(ref_g)                          # reads come from an alternate haplotype of the same length
    reads = np.zeros((nreads, rdlen), dtype=np.uint8)
    if multi > 0:                                            # transcript = the genome with every planted intron spliced out
        keep = np.ones(glen, dtype=bool)
        for a, b in introns:
            keep[a:b] = False
    for i in range(nreads):
        if multi > 0 and alt_fn is None:
            if rng.random() < 0.15:
                s = int(rng.integers(1000, introns[-1][1]))
                r = g[s:s + rdlen].copy()
            else:
                k = int(rng.integers(0, len(introns)))
                s = introns[k][0] - int(rng.integers(1, rdlen))
                r = g[s:s + 60000][keep[s:s + 60000]][:rdlen].copy()
            m = rng.random(rdlen) < sub
            r = np.where(m & (r < 4), (r + rng.integers(1, 4, size=rdlen)) & 3, r).astype(np.uint8)
            if rng.random() < 0.5:
                r = revcomp(r)
            reads[i] = r
            continue
        a, b = introns[int(rng.integers(0, len(introns)))]
        left = int(rng.integers(8, rdlen - 8)) if rng.random() < 0.8 else int(rng.integers(1, rdlen))   # bases before the intron
        if rng.random() < 0.15:                      # unspliced read nearby
            s = a - rdlen - int(rng.integers(0, 50))
            r = g[s:s + rdlen].copy()
        else:
            r = np.concatenate([g[a - left:a], g[b:b + rdlen - left]])
        m = rng.random(rdlen) < sub
        r = np.where(m & (r < 4), (r + rng.integers(1, 4, size=rdlen)) & 3, r).astype(np.uint8)
        if indel > 0 and rng.random() < indel * rdlen:          # one short insertion or deletion somewhere in the read
            at, k = int(rng.integers(5, rdlen - 5)), int(rng.integers(1, 3))
            if rng.random() < 0.5:
                r = np.concatenate([r[:at], rng.integers(0, 4, size=k, dtype=np.uint8), r[at:]])[:rdlen]
            else:
                r = np.concatenate([r[:at], r[at + k:], rng.integers(0, 4, size=k, dtype=np.uint8)])[:rdlen]
        if rng.random() < 0.5:
            r = revcomp(r)
        reads[i] = r
    return ([extra[0], ref_g, extra[2]] if extra else [ref_g]), reads, introns


def known_sites(introns, seed, frac):
    """a --known-splicesite-infile for a fraction of the planted introns plus some sites that do not exist in the reads"""
    rng = np.random.default_rng(seed + 77)
    sites = [(int(rich()), a - 1, b, "+") for a, b in introns if rng.random() < frac]
    for a, b in introns[::7]:
        sites.append((int(rich()), a - 1 - int(rng.integers(3, 40)), b + int(rng.integers(3, 40)), "-" if rng.random() < 0.5 else "+"))
    return sites


def run_case(seed, nreads, sub=0.005, verbose=8, backend=None, extra=(), known=0.0, indel=0.0):
    tmp = tempfile.mkdtemp(prefix="h2spl")
    snps = int(os.environ.get("H2G_FUZZ_SNPS", "0"))      # > 0: SNP-graph index with a seeded variant every ~snps bp
    var = []

    def alt(g):                                           # every single-base variant applied (indels stay index-only)
        var.extend(synth.make_snps([g], seed + 5, every=snps))
        return synth.apply_snps([g], [v for v in var if v[1] == "single"])[0]
    contigs, reads, introns = make_case(seed, nreads, sub=sub, alt_fn=alt if snps else None, indel=indel)
    sites = None
    if known > 0:
        sites = known_sites(introns, seed, known)
        with open(os.path.join(tmp, "ss.txt"), "w") as f:
            for t, l, r, d in sites:
                f.write("chr1\t%d\t%d\t%s\n" % (l, r, d))
        extra = list(extra) + ["--known-splicesite-infile", os.path.join(tmp, "ss.txt")]
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs, names=contig_names(contigs))
    base = os.path.join(tmp, "g")
    if snps:
        synth.write_snps(os.path.join(tmp, "g.snp"), var)
        subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", "--snp", os.path.join(tmp, "g.snp"), fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    else:
        subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rfa = os.path.join(tmp, "r.fa")
    synth.write_reads_fasta(rfa, reads)
    sam = os.path.join(tmp, "ref.sam")
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", "1", "--no-temp-splicesite", "-x", base, "-U", rfa, "-S", sam] + list(extra),
                   check=True, stdout=subprocess.DEVNULL, stderr=open(os.path.join(tmp, "ref.err"), "w"))
    refnames, want = SU.parse_sam(sam)
    qnames = [str(i) for i in range(nreads)]
    rl = [reads[i] for i in range(nreads)]
    if backend is None:
        eopts = [o for o in extra if o != "--known-splicesite-infile" and not str(o).endswith("ss.txt")]
        outs, recs = emu_align(base, rl, qnames, no_spliced=0, options=eopts, splice_sites=sites)
        got = SU.render(outs, recs, refnames, rl, qnames)
    else:
        outs, got = backend(base, reads, qnames, refnames, options=list(extra))
    bad = 0
    nspl = sum(1 for q in qnames if any("N" in r[3] for r in want[q]))
    for q in qnames:
        if got[q] != want[q]:
            bad += 1
            if bad <= verbose:
                print(" read", q, "\n   GOT ", got[q], "\n   WANT", want[q])
    print(f"seed {seed} n {nreads} sub {sub}: spliced(ref) {nspl}  mismatching {bad}  overflow {sum(1 for o in outs if o.overflow)}  tmp {tmp}")
    return bad, tmp


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    sub = float(sys.argv[3]) if len(sys.argv) > 3 else 0.005
    known = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    sys.exit(1 if run_case(seed, n, sub, known=known)[0] else 0)
