#!/usr/bin/env python3
"""Temporary splice sites (the reference's default spliced mode) through the host instantiation: the wave scheme of
hisat2-align-amd restated in Python.  Reads are run in waves of W = 1000 * P ids (P = the reference's -p, hisat2.cpp:3687); after
a wave the junctions of the lines it printed join the database (per site the smallest read id), and read r only sees sites of
reads <= r - W — what `hisat2 -p P --reorder` computes.  usage: temp_splice.py <seed> <nreads> [P]"""
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
from h2gemu_align import emu_align  # noqa: E402
from hisat2_amd import api, synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def wave_run(base, reads, names, P, formatter):
    """formatter(lo, hi, outs, recs, sites_array, nsites, W) -> (lines, novel api.SpliceSite list); returns all lines"""
    W = 1000 * P if P > 1 else 0                    # -p 1: window 0, every read sees all the reads before it (hisat2.cpp:3687)
    step = W if W else 1
    db = {}                                          # (tidx, left, right, dir) -> smallest read id, in first-seen order
    lines = []
    for lo in range(0, len(reads), step):
        hi = min(len(reads), lo + step)
        arr = (api.SpliceSite * max(1, len(db)))()
        for k, ((t, l, r, d), rid) in enumerate(db.items()):
            arr[k].tidx, arr[k].left, arr[k].right, arr[k].readid, arr[k].dir, arr[k].fromfile, arr[k].known = t, l, r, rid, d, 0, 0
        outs, recs = emu_align(base, reads[lo:hi], names[lo:hi], no_spliced=0, splice_sites=(arr, len(db)) if db else None, window=W, rdid_base=lo)
        got, novel = formatter(lo, hi, outs, recs, arr, len(db), W)
        lines += got
        for s in novel:
            key = (s.tidx, s.left, s.right, s.dir)
            if key not in db or s.readid < db[key]:
                db[key] = min(s.readid, db.get(key, s.readid))
    return lines, db


def format_wave(base, reads, names, lo, hi, outs, recs, arr, nsites, W):
    L = SL.load_sam_lib()
    res, aln = SL.emu_to_abi(outs, recs)
    h = C.c_void_p()
    assert L.h2g_sam_open(base.encode(), C.byref(h)) == 0
    L.h2g_sam_set_splice_sites.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]
    L.h2g_sam_set_splice_sites(h, arr, nsites, W)
    L.h2g_sam_collect_novel_sites.argtypes = [C.c_void_p, C.c_int]
    L.h2g_sam_collect_novel_sites(h, 1)
    L.h2g_sam_set_first_read_id.argtypes = [C.c_void_p, C.c_uint64]
    L.h2g_sam_set_first_read_id(h, lo)
    codes, offs = SL.flat(reads[lo:hi])
    nb, noffs = SL.flat_names(names[lo:hi])
    n = hi - lo
    cap = 900 * n * 6 + 4096
    buf = C.create_string_buffer(cap)
    used = C.c_size_t(0)
    rc = L.h2g_sam_format_unpaired(h, codes.ctypes.data, offs.ctypes.data, None, nb, noffs.ctypes.data, n, res, aln, buf, cap, C.byref(used))
    assert rc == 0
    L.h2g_sam_take_novel_sites.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.h2g_sam_take_novel_sites.restype = C.c_size_t
    k = L.h2g_sam_take_novel_sites(h, None, 0)
    out = (api.SpliceSite * max(1, k))()
    L.h2g_sam_take_novel_sites(h, out, k)
    L.h2g_sam_close(h)
    return buf.raw[:used.value].decode().splitlines(), [out[i] for i in range(k)]


def run_case(seed, nreads, P=2, sub=0.005, show=6):
    import fuzz_spliced as F
    from test_sam_lines import diff_lines
    tmp = tempfile.mkdtemp(prefix="h2tmpss")
    contigs, reads, introns = F.make_case(seed, nreads, sub=sub)
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs)
    base = os.path.join(tmp, "g")
    snps = int(os.environ.get("H2G_FUZZ_SNPS", "0"))      # > 0: SNP-graph index (reads stay on the reference haplotype)
    if os.environ.get("H2G_FUZZ_TRAN"):                   # --ss / --exon index over every third planted intron (+ SNPs)
        import fuzz_tran
        fuzz_tran.build(tmp, contigs, introns, seed, snps, every=3)
    elif snps:
        synth.write_snps(os.path.join(tmp, "g.snp"), synth.make_snps(contigs, seed + 5, every=snps))
        subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", "--snp", os.path.join(tmp, "g.snp"), fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    else:
        subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rfa = os.path.join(tmp, "r.fa")
    synth.write_reads_fasta(rfa, reads)
    sam = os.path.join(tmp, "ref.sam")
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", str(P), "--reorder", "-x", base, "-U", rfa, "-S", sam],
                   check=True, stdout=subprocess.DEVNULL, stderr=open(os.path.join(tmp, "ref.err"), "w"))
    names = [str(i) for i in range(nreads)]
    rl = [reads[i] for i in range(nreads)]
    got, db = wave_run(base, rl, names, P, lambda lo, hi, o, r, a, k, W: format_wave(base, rl, names, lo, hi, o, r, a, k, W))
    want = SL.body_lines(sam)
    bad = diff_lines(got, want, show=show)
    # how much the shared database matters: the same reads without it
    sam2 = os.path.join(tmp, "ref_notemp.sam")
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", str(P), "--reorder", "--no-temp-splicesite", "-x", base, "-U", rfa, "-S", sam2],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    delta = sum(1 for a, b in zip(want, SL.body_lines(sam2)) if a != b)
    print(f"seed {seed} n {nreads} -p {P}: differing lines {bad}; database {len(db)} sites; lines the database changes in the reference: {delta}; tmp {tmp}")
    return bad, tmp


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    sys.exit(1 if run_case(seed, n, P)[0] else 0)
