"""Seeded synthetic genomes and reads.

The reference's own simulator (hisat2_simulate_reads.py) crashes under Python 3 for
arbitrary contig names, so workloads are generated here: uniform-random genomes (optionally
with N-gaps and planted repeats) and reads drawn from them with i.i.d. substitutions
(default 0.5 %/base), optional short indels, strand ~ Bernoulli(1/2).  Everything is a pure
function of the seed so the GPU box regenerates byte-identical inputs.
"""
from __future__ import annotations

import numpy as np

_ALPHA = np.frombuffer(b"ACGTN", dtype=np.uint8)
_COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


def make_genome(contig_lens, seed, n_gaps=0, gap_len=500, repeats=0, repeat_len=400):
    """Return list of uint8 arrays (codes 0..4) — one per contig."""
    rng = np.random.default_rng(seed)
    contigs = []
    for L in contig_lens:
        g = rng.integers(0, 4, size=L, dtype=np.uint8)
        for _ in range(repeats):
            if L > 4 * repeat_len:
                a = int(rng.integers(0, L - repeat_len))
                b = int(rng.integers(0, L - repeat_len))
                g[b:b + repeat_len] = g[a:a + repeat_len]
        for _ in range(n_gaps):
            if L > 4 * gap_len:
                a = int(rng.integers(gap_len, L - 2 * gap_len))
                g[a:a + gap_len] = 4
        contigs.append(g)
    return contigs


def make_repeat_genome(contig_lens, seed, alu_frac=0.10, line_frac=0.15, n_tandem_per_mbp=0.8, n_segdup_per_100mbp=12):
    """A genome with HUMAN-LIKE repeat structure (what a uniform-random genome lacks: VERDICT r3 item 5), codes 0..3 per contig:
    * interspersed families — 6 short (~300 bp, Alu-like: many copies) and 6 long (1-6 kbp consensus, LINE-like: copies 5'-truncated to a
      random suffix) — every copy mutated from its family's consensus at a per-copy divergence drawn from U(8 %, 20 %), random strand,
      pasted at a random position; `alu_frac` / `line_frac` of the bases end up in them (human: ~10 % / ~17 %);
    * tandem arrays: a 5-170 bp unit repeated 20-400 times, 2-6 % divergence between units;
    * segmental duplications: 10-100 kbp segments copied elsewhere (also to other contigs) at 1-3 % divergence.
    A pure function of the seed."""
    rng = np.random.default_rng(seed)
    contigs = [rng.integers(0, 4, size=L, dtype=np.uint8) for L in contig_lens]
    total = sum(contig_lens)
    cum = np.cumsum([0] + list(contig_lens))

    def mutate(seq, div):
        out = seq.copy()
        m = rng.random(len(out)) < div
        out[m] = (out[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
        return out

    def paste(seq):
        ci = int(np.searchsorted(cum, rng.integers(0, total), side="right") - 1)
        g = contigs[ci]
        if len(seq) + 2 >= len(g):
            return
        at = int(rng.integers(0, len(g) - len(seq)))
        g[at:at + len(seq)] = seq if rng.random() < 0.5 else (3 - seq[::-1])

    short = [rng.integers(0, 4, size=int(rng.integers(280, 320)), dtype=np.uint8) for _ in range(6)]
    long_ = [rng.integers(0, 4, size=int(rng.integers(1000, 6000)), dtype=np.uint8) for _ in range(6)]
    placed = 0
    while placed < alu_frac * total:
        c = short[int(rng.integers(0, len(short)))]
        paste(mutate(c, rng.uniform(0.08, 0.20))); placed += len(c)
    placed = 0
    while placed < line_frac * total:
        c = long_[int(rng.integers(0, len(long_)))]
        keep = int(rng.integers(300, len(c) + 1))                 # 5'-truncated: a suffix of the consensus
        paste(mutate(c[len(c) - keep:], rng.uniform(0.08, 0.20))); placed += keep
    for _ in range(int(n_tandem_per_mbp * total / 1e6)):
        unit = rng.integers(0, 4, size=int(rng.integers(5, 171)), dtype=np.uint8)
        n = int(rng.integers(20, 401))
        paste(np.concatenate([mutate(unit, rng.uniform(0.02, 0.06)) for _ in range(n)])[:40000])
    for _ in range(max(1, int(n_segdup_per_100mbp * total / 1e8))):
        ci = int(np.searchsorted(cum, rng.integers(0, total), side="right") - 1)
        g = contigs[ci]
        L = int(min(rng.integers(10_000, 100_001), len(g) // 4))
        if L < 1000:
            continue
        a = int(rng.integers(0, len(g) - L))
        paste(mutate(g[a:a + L], rng.uniform(0.01, 0.03)))
    return contigs


def make_snps(contigs, seed, every=250, names=None):
    """Seeded variant set for a graph index (hisat2-build --snp): list of
    (id, type, chrom, pos, data) with type in single|deletion|insertion, >= 12 bp apart, away from Ns."""
    rng = np.random.default_rng(seed)
    out, sid = [], 0
    for ci, g in enumerate(contigs):
        L = len(g)
        if L < 200:
            continue
        pos = np.sort(rng.choice(np.arange(50, L - 50), size=max(1, L // every), replace=False))
        last = -100
        name = names[ci] if names else f"chr{ci + 1}"
        for p in pos:
            p = int(p)
            if p - last < 12 or (g[p - 6:p + 8] > 3).any():
                continue
            last = p
            sid += 1
            t = rng.random()
            if t < 0.86:
                alt = (int(g[p]) + int(rng.integers(1, 4))) & 3
                out.append((f"rs{sid}", "single", name, p, "ACGT"[alt]))
            elif t < 0.93:
                out.append((f"rs{sid}", "deletion", name, p, str(int(rng.integers(1, 4)))))
            else:
                ins = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(1, 4))))
                out.append((f"rs{sid}", "insertion", name, p, ins))
    return out


def write_snps(path, snps):
    with open(path, "w") as f:
        for s in snps:
            f.write("\t".join(map(str, s)) + "\n")


def apply_snps(contigs, snps, names=None):
    """Alternate haplotype carrying every variant of `snps` (they never overlap)."""
    idx = {(names[i] if names else f"chr{i + 1}"): i for i in range(len(contigs))}
    per = {i: [] for i in range(len(contigs))}
    for s in snps:
        per[idx[s[2]]].append(s)
    out = []
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    for i, g in enumerate(contigs):
        parts, cur = [], 0
        for _, typ, _, p, data in sorted(per[i], key=lambda x: x[3]):
            if typ == "single":
                parts.append(g[cur:p]); parts.append(np.array([code[data]], dtype=np.uint8)); cur = p + 1
            elif typ == "deletion":
                parts.append(g[cur:p]); cur = p + int(data)
            else:
                parts.append(g[cur:p]); parts.append(np.array([code[c] for c in data], dtype=np.uint8))
                cur = p
        parts.append(g[cur:])
        out.append(np.concatenate(parts))
    return out


def write_fasta(path, contigs, names=None, width=60):
    with open(path, "wb") as f:
        for i, g in enumerate(contigs):
            name = names[i] if names else f"chr{i + 1}"
            f.write(b">" + name.encode() + b"\n")
            s = _ALPHA[g]
            nfull = len(s) // width
            if nfull:   # all full lines in one vectorised write (a GRCh38-size genome has 52 M lines)
                body = np.empty((nfull, width + 1), dtype=np.uint8)
                body[:, :width] = s[:nfull * width].reshape(nfull, width)
                body[:, width] = 10
                f.write(body.tobytes())
            if len(s) > nfull * width:
                f.write(s[nfull * width:].tobytes() + b"\n")


def make_reads(contigs, n, rdlen, seed, sub_rate=0.005, indel_rate=0.0, n_rate=0.0):
    """Draw n reads of length rdlen.  Returns (codes[n, rdlen] uint8, truth[n,3] = contig,pos,fw)."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    ok = lens > rdlen + 8
    w = np.where(ok, lens - rdlen - 8, 0).astype(np.float64)
    w /= w.sum()
    ci = rng.choice(len(contigs), size=n, p=w)
    reads = np.empty((n, rdlen), dtype=np.uint8)
    truth = np.empty((n, 3), dtype=np.int64)
    for c in range(len(contigs)):
        idx = np.nonzero(ci == c)[0]
        if idx.size == 0:
            continue
        g = contigs[c]
        pos = rng.integers(0, lens[c] - rdlen - 8, size=idx.size)
        win = g[pos[:, None] + np.arange(rdlen + 8)[None, :]]
        # redraw windows containing N (bounded retries)
        for _ in range(8):
            bad = (win[:, :rdlen] > 3).any(axis=1)
            if not bad.any():
                break
            pos[bad] = rng.integers(0, lens[c] - rdlen - 8, size=int(bad.sum()))
            win[bad] = g[pos[bad][:, None] + np.arange(rdlen + 8)[None, :]]
        r = win[:, :rdlen].copy()
        if indel_rate > 0:
            for k in np.nonzero(rng.random(idx.size) < indel_rate * rdlen)[0]:
                p = int(rng.integers(10, rdlen - 10))
                ln = int(rng.integers(1, 4))
                if rng.random() < 0.5:   # deletion from read (ref gap in read)
                    r[k, p:] = win[k, p + ln:p + ln + rdlen - p]
                else:                    # insertion into read
                    ins = rng.integers(0, 4, size=ln, dtype=np.uint8)
                    r[k, p + ln:] = win[k, p:rdlen - ln]
                    r[k, p:p + ln] = ins
        sub = rng.random((idx.size, rdlen)) < sub_rate
        shift = rng.integers(1, 4, size=(idx.size, rdlen), dtype=np.uint8)
        r = np.where(sub & (r < 4), (r + shift) & 3, r).astype(np.uint8)
        if n_rate > 0:
            r = np.where(rng.random((idx.size, rdlen)) < n_rate, np.uint8(4), r)
        fw = rng.random(idx.size) < 0.5
        rc = _COMP[r[:, ::-1]]
        r = np.where(fw[:, None], r, rc)
        reads[idx] = r
        truth[idx, 0] = c
        truth[idx, 1] = pos
        truth[idx, 2] = fw
    return reads, truth


def make_pairs(contigs, n, rdlen, seed, frag_mean=300, frag_sd=30, sub_rate=0.005):
    """--fr pairs: mate1 fw at fragment start, mate2 rc at fragment end (or the mirror)."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    fl = np.clip(np.rint(rng.normal(frag_mean, frag_sd, size=n)), max(150, rdlen), 600).astype(np.int64)
    w = np.where(lens > 700, lens - 700, 0).astype(np.float64)
    w /= w.sum()
    ci = rng.choice(len(contigs), size=n, p=w)
    m1 = np.empty((n, rdlen), dtype=np.uint8)
    m2 = np.empty((n, rdlen), dtype=np.uint8)
    for c in range(len(contigs)):
        idx = np.nonzero(ci == c)[0]
        if idx.size == 0:
            continue
        g = contigs[c]
        pos = rng.integers(0, lens[c] - 700, size=idx.size)
        a = g[pos[:, None] + np.arange(rdlen)[None, :]]
        b = g[(pos + fl[idx] - rdlen)[:, None] + np.arange(rdlen)[None, :]]
        for arr in (a, b):
            sub = rng.random(arr.shape) < sub_rate
            shift = rng.integers(1, 4, size=arr.shape, dtype=np.uint8)
            arr[...] = np.where(sub & (arr < 4), (arr + shift) & 3, arr)
        brc = _COMP[b[:, ::-1]]
        flip = rng.random(idx.size) < 0.5
        m1[idx] = np.where(flip[:, None], brc, a)
        m2[idx] = np.where(flip[:, None], a, brc)
    return m1, m2


def write_reads_fasta(path, reads, start_id=0):
    n, L = reads.shape
    with open(path, "wb") as f:
        txt = _ALPHA[reads]
        for i in range(n):
            f.write(b">%d\n" % (start_id + i))
            f.write(txt[i].tobytes() + b"\n")


def flatten_reads(reads):
    """(codes flat uint8, offsets uint32[n+1]) layout used by the C-ABI."""
    n, L = reads.shape
    offs = (np.arange(n + 1, dtype=np.uint64) * L).astype(np.uint32)
    return np.ascontiguousarray(reads.reshape(-1)), offs
