#!/usr/bin/env python3
"""Development fuzzer for the paired go() (host instantiation) vs the real reference binary."""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import pe_sink as PS  # noqa: E402
import sam_util as SU  # noqa: E402
from h2gemu_py import Emu  # noqa: E402
from hisat2_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


DP = int(os.environ.get("H2G_FUZZ_DP", "0"))   # --bowtie2-dp for both sides
OPTS = ()   # extra reference command-line options (scoring / reporting), applied to both sides
SNPS = int(os.environ.get("H2G_FUZZ_SNPS", "0"))   # > 0: graph index with a seeded variant every ~SNPS bp, pairs from the alt haplotype


def emu_pairs(base, m1, m2, q1, q2):
    e = Emu(base)
    e.L.h2gemu_set_bowtie2_dp.argtypes = [C.c_void_p, C.c_uint32]
    e.L.h2gemu_set_bowtie2_dp(e.h, DP)
    if OPTS:
        from h2gemu_align import set_options
        set_options(e, DP, OPTS)
    n, L = m1.shape
    c1, o1 = synth.flatten_reads(m1)
    c2, o2 = synth.flatten_reads(m2)
    e.set_reads(c1, o1)
    nb1 = "".join(q1).encode(); no1 = np.concatenate([[0], np.cumsum([len(q) for q in q1])]).astype(np.uint32)
    nb2 = "".join(q2).encode(); no2 = np.concatenate([[0], np.cumsum([len(q) for q in q2])]).astype(np.uint32)
    outs = (PS.PairOut * n)()
    r1 = (SU.AlnRec * (n * SU.AL_MAX_RESULTS))()
    r2 = (SU.AlnRec * (n * SU.AL_MAX_RESULTS))()
    vp = C.c_void_p
    e.L.h2gemu_align_pairs.argtypes = [vp, C.c_uint32, vp, vp, C.c_char_p, vp, C.c_char_p, vp, vp, vp, vp]
    e.L.h2gemu_align_pairs(e.h, 1, c2.ctypes.data, o2.ctypes.data, nb1, no1.ctypes.data, nb2, no2.ctypes.data, outs, r1, r2)
    return outs, r1, r2


def parse_pe_sam(path):
    names, recs = [], {}
    for line in open(path):
        if line.startswith("@"):
            if line.startswith("@SQ"):
                names.append(line.split("\t")[1][3:])
            continue
        t = line.rstrip("\n").split("\t")
        a = None
        for x in t[11:]:
            if x.startswith("AS:i:"):
                a = int(x[5:])
        recs.setdefault(t[0], []).append((int(t[1]), t[2], int(t[3]), t[5], a))
    return names, recs


def _flip(m1, m2):      # every 4th mate 2 reverse-complemented: same-strand pairs => discordant (YT:Z:DP)
    m2 = m2.copy()
    m2[::4] = np.where(m2[::4, ::-1] < 4, 3 - m2[::4, ::-1], 4)
    return m1, m2


def _nmask(m1, m2):     # N-filtered mates: the other mate goes through initRead (mate 2: rightendonly, hi_aligner.h:3993); N tails
    m1 = m1.copy(); m2 = m2.copy()
    m1[::3] = 4
    m2[1::7] = 4
    m1[5::11, 50:] = 4
    return m1, m2


MUTATORS = {"flip": _flip, "nmask": _nmask}


def run_case(seed, npairs, rdlen, sub, lens=(300000, 120000, 60000), repeats=6, gaps=2, frag_mean=300, frag_sd=30, verbose=6,
             backend=None, stride=16, mutate=None):
    tmp = tempfile.mkdtemp(prefix="h2pe")
    contigs = synth.make_genome(list(lens), seed, n_gaps=gaps, gap_len=300, repeats=repeats, repeat_len=500)
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs)
    base = os.path.join(tmp, "g")
    src = contigs
    if SNPS:
        var = synth.make_snps(contigs, seed + 5, every=SNPS)
        synth.write_snps(os.path.join(tmp, "g.snp"), var)
        subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", "--snp", os.path.join(tmp, "g.snp"), fa, base], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        src = synth.apply_snps(contigs, var)
    else:
        subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    m1, m2 = synth.make_pairs(src, npairs, rdlen, seed + 1, frag_mean=frag_mean, frag_sd=frag_sd, sub_rate=sub)
    if mutate is not None:   # a name in MUTATORS or a callable (m1, m2) -> (m1, m2)
        m1, m2 = (MUTATORS[mutate] if isinstance(mutate, str) else mutate)(m1, m2)
    f1, f2 = os.path.join(tmp, "r1.fa"), os.path.join(tmp, "r2.fa")
    synth.write_reads_fasta(f1, m1)
    synth.write_reads_fasta(f2, m2)
    sam = os.path.join(tmp, "ref.sam")
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", "1", "--no-spliced-alignment", "-x", base, "-1", f1, "-2", f2, "-S", sam] + (["--bowtie2-dp", str(DP)] if DP else []) + list(OPTS),
                   check=True, stdout=subprocess.DEVNULL, stderr=open(os.path.join(tmp, "ref.err"), "w"))
    refnames, want = parse_pe_sam(sam)
    # -k as the sink sees it (hisat2.cpp:336 default 10, :1891-1907 the presets raise a smaller one, :3903 no -k seen: 5 on a linear index, 10 on a graph)
    saw_k = "-k" in OPTS
    khits = int(OPTS[OPTS.index("-k") + 1]) if saw_k else 10
    if "--sensitive" in OPTS and khits < 10:
        khits, saw_k = 10, True
    elif "--very-sensitive" in OPTS and "--sensitive" not in OPTS and khits < 30:
        khits, saw_k = 30, True
    if not saw_k:
        khits = 10 if SNPS else 5
    secondary = "--secondary" in OPTS
    q = [str(i) for i in range(npairs)]
    outs, r1, r2 = (backend or emu_pairs)(base, m1, m2, q, q)
    bad = ovf = setbad = 0
    ncon = 0
    for i in range(npairs):
        got = PS.finish_pair(outs[i], r1, r2, i * (stride if backend else SU.AL_MAX_RESULTS), refnames, (m1[i], m2[i]), khits=khits, secondary=secondary)
        w = want[q[i]]
        ncon += 1 if (w[0][0] & 2) else 0
        ovf += 1 if outs[i].overflow else 0
        if got != w:
            bad += 1
            if sorted(got, key=str) != sorted(w, key=str):
                setbad += 1
            if bad <= verbose:
                print(" pair", i, ("ovf%d" % outs[i].overflow) if outs[i].overflow else "", "\n   GOT ", got, "\n   WANT", w)
    print(f"PE seed {seed} n {npairs} len {rdlen} sub {sub}: concordant(ref) {ncon}  mismatching {bad} (set-level {setbad})  overflow {ovf}  tmp {tmp}")
    return bad, tmp


if __name__ == "__main__":
    a = sys.argv[1:]
    run_case(int(a[0]), int(a[1]), int(a[2]), float(a[3]))
