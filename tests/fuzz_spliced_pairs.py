#!/usr/bin/env python3
"""Development fuzzer for paired spliced alignment (--no-temp-splicesite): fragments drawn from the spliced transcript of
fuzz_spliced's genome, mate 2 reverse-complemented; host instantiation of the go() machine + the SAM formatter vs the lines of
the real reference binary.  usage: fuzz_spliced_pairs.py <seed> <npairs> [sub]"""
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
import sam_lines as SL  # noqa: E402
import sam_util as SU  # noqa: E402
import pe_sink as PS  # noqa: E402
from h2gemu_py import Emu  # noqa: E402
from hisat2_amd import api, synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def make_case(seed, npairs, rdlen=101, sub=0.005, glen=400000, nintrons=300, frag_mean=280, frag_sd=40):
    import fuzz_spliced as FS
    rdlen = int(os.environ.get("H2G_FUZZ_RDLEN", rdlen))
    multi = float(os.environ.get("H2G_FUZZ_MULTI", "0"))   # that fraction of the exons is 20..90 bp long
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=glen, dtype=np.uint8)
    introns = []
    pos = 2000
    while len(introns) < nintrons and pos < glen - 12000:
        L = int(rng.choice([60, 90, 150, 400, 1200, 5000, 9000]))
        a, b = pos, pos + L
        kind = int(rng.integers(0, 10))
        if kind < 8:
            g[a:a + 2] = [2, 3]; g[b - 2:b] = [0, 2]
        elif kind == 8:
            g[a:a + 2] = [2, 1]; g[b - 2:b] = [0, 2]
        introns.append((a, b))
        pos = b + (int(rng.integers(20, 90)) if multi > 0 and rng.random() < multi else int(rng.integers(120, 700)))
    extra = None
    if FS.rich():                                    # three contigs, N runs, gene copies, processed pseudogenes (fuzz_spliced.enrich)
        extra = FS.enrich(g, introns, rng)
        g = extra[1]
    keep = np.ones(glen, dtype=bool)
    for a, b in introns:
        keep[a:b] = False
    tx = g[keep]                                     # the transcript: every intron removed
    m1 = np.zeros((npairs, rdlen), dtype=np.uint8)
    m2 = np.zeros((npairs, rdlen), dtype=np.uint8)
    for i in range(npairs):
        fl = max(rdlen, int(rng.normal(frag_mean, frag_sd)))
        s = int(rng.integers(0, len(tx) - fl))
        f = tx[s:s + fl].copy()
        m = rng.random(fl) < sub
        f = np.where(m & (f < 4), (f + rng.integers(1, 4, size=fl)) & 3, f).astype(np.uint8)
        if rng.random() < 0.5:
            f = FS.revcomp(f)
        m1[i] = f[:rdlen]
        m2[i] = FS.revcomp(f)[:rdlen]
        if rng.random() < float(os.environ.get("H2G_FUZZ_CHIMERA", "0")):    # mate 2 from elsewhere (discordant pairs), sometimes same strand
            s2 = int(rng.integers(0, len(tx) - rdlen))
            m2[i] = tx[s2:s2 + rdlen] if rng.random() < 0.5 else FS.revcomp(tx[s2:s2 + rdlen])
    return ([extra[0], g, extra[2]] if extra else [g]), m1, m2, introns


def emu_pairs(base, m1, m2, q1, q2, options=(), splice_sites=None):
    e = Emu(base)
    from h2gemu_align import set_options, set_splice_sites
    set_options(e, 0, list(options))
    if splice_sites:
        set_splice_sites(e, splice_sites)
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
    e.L.h2gemu_align_pairs(e.h, 0, c2.ctypes.data, o2.ctypes.data, nb1, no1.ctypes.data, nb2, no2.ctypes.data, outs, r1, r2)
    return outs, r1, r2


def run_case(seed, npairs, sub=0.005, extra=(), show=6, known=0.0, novel_out=False):
    tmp = tempfile.mkdtemp(prefix="h2splpe")
    contigs, m1, m2, introns = make_case(seed, npairs, sub=sub)
    sites, sopt = None, []
    if known > 0:
        import fuzz_spliced as FS
        sites = FS.known_sites(introns, seed, known)
        with open(os.path.join(tmp, "ss.txt"), "w") as f:
            for t, l, r, d in sites:
                f.write("chr1\t%d\t%d\t%s\n" % (l, r, d))
        sopt = ["--known-splicesite-infile", os.path.join(tmp, "ss.txt")]
    fa = os.path.join(tmp, "g.fa")
    import fuzz_spliced as FS
    synth.write_fasta(fa, contigs, names=FS.contig_names(contigs))
    base = os.path.join(tmp, "g")
    snps = int(os.environ.get("H2G_FUZZ_SNPS", "0"))      # > 0: SNP-graph index (reads stay on the reference haplotype)
    if snps:
        synth.write_snps(os.path.join(tmp, "g.snp"), synth.make_snps(contigs, seed + 5, every=snps, names=FS.contig_names(contigs)))
        subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", "--snp", os.path.join(tmp, "g.snp"), fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    else:
        subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f1, f2 = os.path.join(tmp, "r1.fa"), os.path.join(tmp, "r2.fa")
    synth.write_reads_fasta(f1, m1)
    synth.write_reads_fasta(f2, m2)
    sam = os.path.join(tmp, "ref.sam")
    nopt = ["--novel-splicesite-outfile", os.path.join(tmp, "ref.ss")] if novel_out else []
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", "1", "--no-temp-splicesite", "-x", base, "-1", f1, "-2", f2, "-S", sam] + list(extra) + sopt + nopt,
                   check=True, stdout=subprocess.DEVNULL, stderr=open(os.path.join(tmp, "ref.err"), "w"))
    q = [str(i) for i in range(npairs)]
    outs, r1, r2 = emu_pairs(base, m1, m2, q, q, options=extra, splice_sites=sites)
    n = npairs
    res = (api.PairResult * n)()
    C.memmove(res, outs, C.sizeof(res))
    # every record the machine reported, back to back (the dense layout): a pair whose mate reports more than H2G_PAIR_RES_CAP alignments is
    # flagged and re-run with more room on the device; the host instantiation already holds them all
    cnt = [[min(int(outs[i].nres[m]), SU.AL_MAX_RESULTS) for i in range(n)] for m in range(2)]
    offs = [np.concatenate([[0], np.cumsum(cnt[m])]).astype(np.uint64) for m in range(2)]
    a1 = (api.AlnRes * max(1, int(offs[0][-1])))()
    a2 = (api.AlnRes * max(1, int(offs[1][-1])))()
    for i in range(n):
        for m, (src, dst) in enumerate(((r1, a1), (r2, a2))):
            for k in range(cnt[m][i]):
                C.memmove(C.byref(dst[int(offs[m][i]) + k]), C.byref(src[i * SU.AL_MAX_RESULTS + k]), C.sizeof(api.AlnRes))
    khits = int(extra[extra.index("-k") + 1]) if "-k" in extra else (10 if snps else 5)
    nopt = ["--novel-splicesite-outfile", os.path.join(tmp, "our.ss")] if novel_out else []
    got = SL.format_paired(SL.load_sam_lib(), base, list(m1), list(m2), q, q, res, a1, a2, khits, options=list(extra) + sopt + nopt, dense=(offs[0], offs[1]))
    want = SL.body_lines(sam)
    from test_sam_lines import diff_lines
    bad = diff_lines(got, want, show=show)
    if novel_out and open(os.path.join(tmp, "our.ss")).read() != open(os.path.join(tmp, "ref.ss")).read():
        print("novel-splicesite-outfile differs:", os.path.join(tmp, "our.ss"), os.path.join(tmp, "ref.ss"))
        bad += 1
    nspl = sum(1 for l in want if "N" in l.split("\t")[5])
    ovf = sum(1 for o in outs if o.overflow)
    print(f"seed {seed} pairs {npairs} sub {sub}: spliced lines(ref) {nspl}  differing lines {bad}  overflow {ovf}  summary {'same' if SL.LAST_SUMMARY == open(os.path.join(tmp, 'ref.err')).read() else 'DIFFERENT'}  tmp {tmp}")
    return bad, tmp


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    sub = float(sys.argv[3]) if len(sys.argv) > 3 else 0.005
    known = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    sys.exit(1 if run_case(seed, n, sub, known=known)[0] else 0)
