#!/usr/bin/env python3
"""Spliced alignment on REAL sequence (development fuzzer): the reference's example contig (1 Mbp of chr22, tests/golden) with
introns chosen at GT..AG pairs that exist in it, reads drawn from the resulting transcript (short exons: two or three junctions per
read), optionally over the dbSNP variants of the example (SNP-graph / --ss --exon index).  Installs itself as fuzz_spliced.make_case
so that every harness (fuzz_spliced, temp_splice, fuzz_tran) runs on it.  usage: fuzz_real.py <seed> <nreads> <mode>, mode in
notemp | temp | tran | snp | snptemp | pairs | snppairs"""
import gzip
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import fuzz_spliced as F  # noqa: E402

GOLD = os.path.join(HERE, "golden")


def real_contig():
    seq = b"".join(l.strip() for l in gzip.open(os.path.join(GOLD, "example_22_20-21M.fa.gz"), "rb") if not l.startswith(b">")).upper()
    code = np.full(256, 4, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    return code[np.frombuffer(seq, dtype=np.uint8)]


def make_case(seed, nreads, rdlen=101, sub=0.005, alt_fn=None, indel=0.0, **_):
    rdlen = int(os.environ.get("H2G_FUZZ_RDLEN", rdlen))
    rng = np.random.default_rng(seed)
    g = real_contig()
    n = len(g)
    gt = np.flatnonzero((g[:-1] == 2) & (g[1:] == 3))                  # donors: GT at a
    ag = set(np.flatnonzero((g[:-1] == 0) & (g[1:] == 2)).tolist())      # acceptors: AG at b-2
    introns, pos = [], 3000
    while pos < n - 20000 and len(introns) < 400:
        k = np.searchsorted(gt, pos)
        if k >= len(gt):
            break
        a = int(gt[k])
        b = None
        for L in rng.permutation([70, 110, 180, 420, 1300, 4800, 9000]):
            for d in range(0, 60):                                       # the nearest AG at about that length
                if (a + int(L) + d - 2) in ag:
                    b = a + int(L) + d
                    break
            if b:
                break
        if b is None or (g[a - 30:b + 30] > 3).any():
            pos = a + 50
            continue
        introns.append((a, b))
        pos = b + (int(rng.integers(25, 90)) if rng.random() < 0.4 else int(rng.integers(150, 1500)))
    src = alt_fn(g) if alt_fn is not None else g
    keep = np.ones(n, dtype=bool)
    for a, b in introns:
        keep[a:b] = False
    reads = np.zeros((nreads, rdlen), dtype=np.uint8)
    for i in range(nreads):
        while True:
            if rng.random() < 0.2:
                s = int(rng.integers(1000, n - 2 * rdlen))
                r = src[s:s + rdlen].copy()
            else:
                a = introns[int(rng.integers(0, len(introns)))][0]
                s = a - int(rng.integers(1, rdlen))
                r = src[s:s + 60000][keep[s:s + 60000]][:rdlen].copy()
            if len(r) == rdlen and (r < 4).all():
                break
        m = rng.random(rdlen) < sub
        r = np.where(m, (r + rng.integers(1, 4, size=rdlen)) & 3, r).astype(np.uint8)
        if rng.random() < 0.5:
            r = F.revcomp(r)
        reads[i] = r
    return [g], reads, introns


def make_pairs_case(seed, npairs, rdlen=101, sub=0.005, frag_mean=280, frag_sd=40, **_):
    """fuzz_spliced_pairs.make_case on the real contig: fragments of the transcript, --fr mates"""
    rdlen = int(os.environ.get("H2G_FUZZ_RDLEN", rdlen))
    contigs, _, introns = make_case(seed, 1, rdlen=rdlen)
    g = contigs[0]
    rng = np.random.default_rng(seed + 3)
    keep = np.ones(len(g), dtype=bool)
    for a, b in introns:
        keep[a:b] = False
    tx = g[keep]
    m1 = np.zeros((npairs, rdlen), dtype=np.uint8)
    m2 = np.zeros((npairs, rdlen), dtype=np.uint8)
    for i in range(npairs):
        while True:
            fl = max(rdlen, int(rng.normal(frag_mean, frag_sd)))
            s = int(rng.integers(0, len(tx) - fl))
            f = tx[s:s + fl].copy()
            if (f < 4).all():
                break
        m = rng.random(fl) < sub
        f = np.where(m, (f + rng.integers(1, 4, size=fl)) & 3, f).astype(np.uint8)
        if rng.random() < 0.5:
            f = F.revcomp(f)
        m1[i] = f[:rdlen]
        m2[i] = F.revcomp(f)[:rdlen]
    return [g], m1, m2, introns


def install():
    F.make_case = make_case
    import fuzz_spliced_pairs as FP
    FP.make_case = make_pairs_case


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    mode = sys.argv[3] if len(sys.argv) > 3 else "notemp"
    install()
    if mode in ("snp", "snptemp"):
        os.environ["H2G_FUZZ_SNPS"] = "300"
    if mode == "tran":
        os.environ["H2G_FUZZ_TRAN"] = "1"
    if mode in ("pairs", "snppairs"):
        import fuzz_spliced_pairs as FP
        if mode == "snppairs":
            os.environ["H2G_FUZZ_SNPS"] = "300"
        bad = FP.run_case(seed, n, 0.005, known=0.4, show=3)[0]
    elif mode in ("notemp", "snp"):
        bad = F.run_case(seed, n, 0.005, known=0.5, verbose=3)[0]
    else:
        import temp_splice as T
        bad = T.run_case(seed, n, P=2, show=3)[0]
    sys.exit(1 if bad else 0)
