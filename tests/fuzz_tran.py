#!/usr/bin/env python3
"""Development fuzzer for `--ss / --exon` indexes (the reference's _tran indexes: splice sites as graph edges + known sites):
fuzz_spliced's genome and reads, an index built with half of the planted introns as splice-site ALTs (optionally SNPs too),
host instantiation vs the real reference binary.  usage: fuzz_tran.py <seed> <nreads> [sub] [snps_every]"""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import fuzz_spliced as F  # noqa: E402
import sam_util as SU  # noqa: E402
from h2gemu_align import emu_align  # noqa: E402
from hisat2_amd import synth  # noqa: E402


def build(tmp, contigs, introns, seed, snps=0, every=2):
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs, names=__import__('fuzz_spliced').contig_names(contigs))
    with open(os.path.join(tmp, "ss.txt"), "w") as f:
        for a, b in introns[::every]:
            f.write("chr1\t%d\t%d\t+\n" % (a - 1, b))
    with open(os.path.join(tmp, "exon.txt"), "w") as f:
        prev = 0
        for a, b in introns:
            f.write("chr1\t%d\t%d\n" % (max(prev, a - 400), a - 1)); prev = b
    base = os.path.join(tmp, "g")
    cmd = [os.path.join(F.REF, "hisat2-build-s"), "-q", "--ss", os.path.join(tmp, "ss.txt"), "--exon", os.path.join(tmp, "exon.txt")]
    if snps:
        synth.write_snps(os.path.join(tmp, "g.snp"), synth.make_snps(contigs, seed + 5, every=snps, names=__import__('fuzz_spliced').contig_names(contigs)))
        cmd += ["--snp", os.path.join(tmp, "g.snp")]
    subprocess.run(cmd + [fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return base


def run_case(seed, nreads, sub=0.01, snps=0, verbose=6, extra=(), lines=False):
    tmp = tempfile.mkdtemp(prefix="h2tran")
    contigs, reads, introns = F.make_case(seed, nreads, sub=sub)
    base = build(tmp, contigs, introns, seed, snps)
    rfa = os.path.join(tmp, "r.fa")
    synth.write_reads_fasta(rfa, reads)
    sam = os.path.join(tmp, "ref.sam")
    subprocess.run([os.path.join(F.REF, "hisat2-align-s"), "-f", "-p", "1", "--no-temp-splicesite", "-x", base, "-U", rfa, "-S", sam] + list(extra), check=True,
                   stdout=subprocess.DEVNULL, stderr=open(os.path.join(tmp, "ref.err"), "w"))
    refnames, want = SU.parse_sam(sam)
    q = [str(i) for i in range(nreads)]
    rl = [reads[i] for i in range(nreads)]
    outs, recs = emu_align(base, rl, q, no_spliced=0, options=list(extra))
    got = SU.render(outs, recs, refnames, rl, q)
    if lines:                                     # complete SAM lines + the alignment summary
        import sam_lines as SL
        from test_sam_lines import diff_lines
        res, aln = SL.emu_to_abi(outs, recs)
        gl = SL.format_unpaired(SL.load_sam_lib(), base, rl, q, res, aln, options=list(extra))
        nbad = diff_lines(gl, SL.body_lines(sam), show=verbose)
        same = SL.LAST_SUMMARY == open(os.path.join(tmp, "ref.err")).read()
        print(f"seed {seed} n {nreads} sub {sub} snps {snps}: differing lines {nbad}  summary {'same' if same else 'DIFFERENT'}  overflow {sum(1 for o in outs if o.overflow)}  tmp {tmp}")
        return nbad + (0 if same else 1), tmp
    bad = [x for x in q if got[x] != want[x]]
    for x in bad[:verbose]:
        print(" read", x, "\n   GOT ", got[x], "\n   WANT", want[x])
    print(f"seed {seed} n {nreads} sub {sub} snps {snps}: spliced(ref) {sum(1 for x in q if any('N' in r[3] for r in want[x]))}  mismatching {len(bad)}  overflow {sum(1 for o in outs if o.overflow)}  tmp {tmp}")
    return len(bad), tmp


def run_pairs(seed, npairs, sub=0.01, snps=0, show=4, extra=()):
    """paired reads over the same kind of index, complete SAM lines"""
    import ctypes as C
    import fuzz_spliced_pairs as FP
    import sam_lines as SL
    from hisat2_amd import api
    from test_sam_lines import diff_lines
    tmp = tempfile.mkdtemp(prefix="h2tranpe")
    contigs, m1, m2, introns = FP.make_case(seed, npairs, sub=sub)
    base = build(tmp, contigs, introns, seed, snps)
    f1, f2 = os.path.join(tmp, "r1.fa"), os.path.join(tmp, "r2.fa")
    synth.write_reads_fasta(f1, m1)
    synth.write_reads_fasta(f2, m2)
    sam = os.path.join(tmp, "ref.sam")
    subprocess.run([os.path.join(F.REF, "hisat2-align-s"), "-f", "-p", "1", "--no-temp-splicesite", "-x", base, "-1", f1, "-2", f2, "-S", sam] + list(extra),
                   check=True, stdout=subprocess.DEVNULL, stderr=open(os.path.join(tmp, "ref.err"), "w"))
    q = [str(i) for i in range(npairs)]
    outs, r1, r2 = FP.emu_pairs(base, m1, m2, q, q, options=extra)
    n = npairs
    res = (api.PairResult * n)()
    a1 = (api.AlnRes * (n * api.PAIR_RES_CAP))()
    a2 = (api.AlnRes * (n * api.PAIR_RES_CAP))()
    C.memmove(res, outs, C.sizeof(res))
    for i in range(n):
        for m, (src, dst) in enumerate(((r1, a1), (r2, a2))):
            for k in range(min(outs[i].nres[m], api.PAIR_RES_CAP)):
                C.memmove(C.byref(dst[i * api.PAIR_RES_CAP + k]), C.byref(src[i * SU.AL_MAX_RESULTS + k]), C.sizeof(api.AlnRes))
    got = SL.format_paired(SL.load_sam_lib(), base, list(m1), list(m2), q, q, res, a1, a2, 10, options=list(extra))
    bad = diff_lines(got, SL.body_lines(sam), show=show)
    same = SL.LAST_SUMMARY == open(os.path.join(tmp, "ref.err")).read()
    print(f"seed {seed} pairs {npairs} sub {sub} snps {snps}: differing lines {bad}  summary {'same' if same else 'DIFFERENT'}  overflow {sum(1 for o in outs if o.overflow)}  tmp {tmp}")
    return bad + (0 if same else 1), tmp


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    sub = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
    snps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    sys.exit(1 if run_case(seed, n, sub, snps)[0] else 0)
