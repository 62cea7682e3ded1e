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


def wave_run(base, reads, names, P, formatter, file_sites=(), rank=0, world=1, exchange=None):
    """formatter(lo, hi, outs, recs, sites_array, nsites, W) -> (lines, novel api.SpliceSite list); returns all lines.
    file_sites: (tidx, left, right, '+'/'-') of a --known-splicesite-infile — always visible, never replaced by a read's.
    world > 1: this process runs shard `rank` of every wave (hisat2_amd.shard.shard_range inside the wave) and exchange(rows) returns
    every rank's new junctions (shard.all_gather_junctions); the lines returned are this rank's, wave by wave: [(wave, lines), ...]"""
    from hisat2_amd import shard
    W = 1000 * P if P > 1 else 0                    # -p 1: window 0, every read sees all the reads before it (hisat2.cpp:3687)
    step = W if W else 1
    db = {}                                          # (tidx, left, right, dir) -> smallest read id, in first-seen order
    fixed = {}
    for t, l, r, d in file_sites:
        fixed.setdefault((t, l, r, 2 if d == "+" else 3), None)
    lines = []
    for wave, wlo in enumerate(range(0, len(reads), step)):
        whi = min(len(reads), wlo + step)
        lo, hi = wlo, whi
        if world > 1:
            a, b = shard.shard_range(whi - wlo, rank, world)
            lo, hi = wlo + a, wlo + b
        nall = len(fixed) + len(db)
        arr = (api.SpliceSite * max(1, nall))()
        for k, (t, l, r, d) in enumerate(fixed):
            arr[k].tidx, arr[k].left, arr[k].right, arr[k].readid, arr[k].dir, arr[k].fromfile, arr[k].known = t, l, r, 0, d, 1, 1
        for k, ((t, l, r, d), rid) in enumerate(db.items(), len(fixed)):
            arr[k].tidx, arr[k].left, arr[k].right, arr[k].readid, arr[k].dir, arr[k].fromfile, arr[k].known = t, l, r, rid, d, 0, 0
        outs, recs = emu_align(base, reads[lo:hi], names[lo:hi], no_spliced=0, splice_sites=(arr, nall) if nall else None, window=W, rdid_base=lo)
        got, novel = formatter(lo, hi, outs, recs, arr, nall, W)
        if world > 1:
            lines.append((wave, got))
            rows = exchange(np.array([(s.tidx, s.left, s.right, s.dir, s.readid) for s in novel], dtype=np.int64).reshape(-1, 5))
            novel = [api.SpliceSite(tidx=int(r[0]), left=int(r[1]), right=int(r[2]), dir=int(r[3]), readid=int(r[4])) for r in rows]
        else:
            lines += got
        for s in novel:
            key = (s.tidx, s.left, s.right, s.dir)
            if key in fixed:
                continue
            if key not in db or s.readid < db[key]:
                db[key] = min(s.readid, db.get(key, s.readid))
    return lines, db


def novel_text(h):
    """--novel-splicesite-outfile of the formatter handle h (SpliceSiteDB::print)"""
    L = SL.load_sam_lib()
    L.h2g_sam_novel_splice_sites_text.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.h2g_sam_novel_splice_sites_text.restype = C.c_size_t
    need = L.h2g_sam_novel_splice_sites_text(h, None, 0)
    buf = C.create_string_buffer(need + 1)
    L.h2g_sam_novel_splice_sites_text(h, buf, need)
    return buf.raw[:need].decode()


def format_wave(base, reads, names, lo, hi, outs, recs, arr, nsites, W, handle=None):
    """handle: one formatter for every wave (its site statistics then cover the whole run); None opens one per wave"""
    L = SL.load_sam_lib()
    res, aln = SL.emu_to_abi(outs, recs)
    h = handle or C.c_void_p()
    if handle is None:
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
    if handle is None:
        L.h2g_sam_close(h)
    return buf.raw[:used.value].decode().splitlines(), [out[i] for i in range(k)]


def run_case(seed, nreads, P=2, sub=0.005, show=6, known=0.0, ref_opts=()):
    """known: fraction of the planted introns given as --known-splicesite-infile.  ref_opts: reference options that must not change
    the wave scheme's result (e.g. --no-temp-splicesite next to --novel-splicesite-outfile and a known file: the database is still
    written and read, hisat2.cpp:4092-4093)"""
    import fuzz_spliced as F
    from test_sam_lines import diff_lines
    tmp = tempfile.mkdtemp(prefix="h2tmpss")
    contigs, reads, introns = F.make_case(seed, nreads, sub=sub)
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs, names=F.contig_names(contigs))
    base = os.path.join(tmp, "g")
    snps = int(os.environ.get("H2G_FUZZ_SNPS", "0"))      # > 0: SNP-graph index (reads stay on the reference haplotype)
    if os.environ.get("H2G_FUZZ_TRAN"):                   # --ss / --exon index over every third planted intron (+ SNPs)
        import fuzz_tran
        fuzz_tran.build(tmp, contigs, introns, seed, snps, every=3)
    elif snps:
        synth.write_snps(os.path.join(tmp, "g.snp"), synth.make_snps(contigs, seed + 5, every=snps, names=F.contig_names(contigs)))
        subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", "--snp", os.path.join(tmp, "g.snp"), fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    else:
        subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rfa = os.path.join(tmp, "r.fa")
    synth.write_reads_fasta(rfa, reads)
    sam = os.path.join(tmp, "ref.sam")
    file_sites, sopt = (), []
    if known > 0:
        file_sites = F.known_sites(introns, seed, known)
        with open(os.path.join(tmp, "ss.txt"), "w") as f:
            for t, l, r, d in file_sites:
                f.write("chr1\t%d\t%d\t%s\n" % (l, r, d))
        sopt = ["--known-splicesite-infile", os.path.join(tmp, "ss.txt")]
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", str(P), "--reorder", "--novel-splicesite-outfile", os.path.join(tmp, "ref.ss"), "-x", base, "-U", rfa, "-S", sam] + sopt + list(ref_opts),
                   check=True, stdout=subprocess.DEVNULL, stderr=open(os.path.join(tmp, "ref.err"), "w"))
    names = [str(i) for i in range(nreads)]
    rl = [reads[i] for i in range(nreads)]
    L = SL.load_sam_lib()
    h = C.c_void_p()
    assert L.h2g_sam_open(base.encode(), C.byref(h)) == 0
    got, db = wave_run(base, rl, names, P, lambda lo, hi, o, r, a, k, W: format_wave(base, rl, names, lo, hi, o, r, a, k, W, handle=h), file_sites=file_sites)
    want = SL.body_lines(sam)
    bad = diff_lines(got, want, show=show)
    ss_got, ss_want = novel_text(h).splitlines(), open(os.path.join(tmp, "ref.ss")).read().splitlines()
    L.h2g_sam_close(h)
    if ss_got != ss_want:                            # --novel-splicesite-outfile: the same sites in the same order
        extra, missing = sorted(set(ss_got) - set(ss_want)), sorted(set(ss_want) - set(ss_got))
        print(f"novel-splicesite-outfile differs: {len(ss_got)} lines vs {len(ss_want)}; only here {extra[:4]}; only in the reference {missing[:4]}")
        bad += 1
    # how much the shared database matters: the same reads without it
    sam2 = os.path.join(tmp, "ref_notemp.sam")
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", str(P), "--reorder", "--no-temp-splicesite", "-x", base, "-U", rfa, "-S", sam2] + sopt,
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    delta = sum(1 for a, b in zip(want, SL.body_lines(sam2)) if a != b)
    print(f"seed {seed} n {nreads} -p {P}: differing lines {bad}; database {len(db)} sites; lines the database changes in the reference: {delta}; tmp {tmp}")
    return bad, tmp


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    sys.exit(1 if run_case(seed, n, P)[0] else 0)
