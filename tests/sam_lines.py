"""Full-line SAM comparison helpers: run the C++ SAM emitter (include/h2g_sam.h) over alignment records and diff every
non-header line against the reference's SAM text."""
import ctypes as C
import os

import numpy as np

from hisat2_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


LAST_SUMMARY = None   # alignment summary text of the last format_* call (h2g_sam_summary)
LAST_HEADER = ""


def load_sam_lib(path=None):
    L = C.CDLL(path or os.environ.get("H2G_SAM_LIB") or os.path.join(ROOT, "hisat2_amd", "libh2g.so"))
    L.h2g_sam_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.h2g_sam_close.argtypes = [C.c_void_p]
    L.h2g_sam_format_unpaired.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.h2g_sam_format_paired.argtypes = [C.c_void_p] + [C.c_void_p] * 10 + [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t,
                                                                             C.POINTER(C.c_size_t)]
    L.h2g_sam_format_paired_dense.argtypes = [C.c_void_p] + [C.c_void_p] * 10 + [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                               C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.h2g_sam_header.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    L.h2g_sam_header.restype = C.c_size_t
    L.h2g_sam_summary.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.h2g_sam_summary.restype = C.c_size_t
    L.h2g_sam_set_score_min.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_double]
    return L


def flat(reads):
    codes = np.concatenate([np.asarray(r, dtype=np.uint8) for r in reads])
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint32)
    return codes, offs


def flat_names(names):
    nb = "".join(names).encode()
    noffs = np.concatenate([[0], np.cumsum([len(q) for q in names])]).astype(np.uint32)
    return nb, noffs


def _score_min(L, h, options):
    if options and any(o in options for o in ("--score-min", "--sensitive", "--very-sensitive")):   # the presets carry a --score-min
        p = api.AlignParams()
        p.apply_options(list(options))
        L.h2g_sam_set_score_min(h, p.score_min_type, p.score_min_const, p.score_min_coeff)
    if options and "--secondary" in options:
        L.h2g_sam_set_secondary.argtypes = [C.c_void_p, C.c_int]
        L.h2g_sam_set_secondary(h, 1)
    if options and "--rna-strandness" in options:
        code = {"F": 1, "R": 2, "FR": 3, "RF": 4}[options[list(options).index("--rna-strandness") + 1]]
        L.h2g_sam_set_rna_strandness.argtypes = [C.c_void_p, C.c_int]
        L.h2g_sam_set_rna_strandness(h, code)
    if options and "--known-splicesite-infile" in options:
        fn = options[list(options).index("--known-splicesite-infile") + 1].encode()
        L.h2g_sam_read_splice_site_file.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t]
        L.h2g_sam_read_splice_site_file.restype = C.c_size_t
        n = L.h2g_sam_read_splice_site_file(h, fn, 1, None, 0)
        a = (api.SpliceSite * max(1, n))()
        L.h2g_sam_read_splice_site_file(h, fn, 1, a, n)
        L.h2g_sam_set_splice_sites.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]
        L.h2g_sam_set_splice_sites(h, a, n, 0)
    if options and "--novel-splicesite-outfile" in options:
        L.h2g_sam_collect_novel_sites.argtypes = [C.c_void_p, C.c_int]
        L.h2g_sam_collect_novel_sites(h, 1)
    if options and ("--rg-id" in options or "--rg" in options):
        L.h2g_sam_add_read_group.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        ol = list(options)
        for k, o in enumerate(ol):
            if o == "--rg-id":
                L.h2g_sam_add_read_group(h, ol[k + 1].encode(), None)
            elif o == "--rg":
                L.h2g_sam_add_read_group(h, None, ol[k + 1].encode())
    if options and ("--no-sq" in options or "--omit-sec-seq" in options):
        L.h2g_sam_set_header_options.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.h2g_sam_set_header_options(h, 1 if "--no-sq" in options else 0, 1 if "--omit-sec-seq" in options else 0)
    if options and ("--add-chrname" in options or "--remove-chrname" in options):
        L.h2g_sam_set_chrname_mode.argtypes = [C.c_void_p, C.c_int]
        L.h2g_sam_set_chrname_mode(h, 1 if "--remove-chrname" in options else 2)
    if options and "--new-summary" in options:
        L.h2g_sam_set_new_summary.argtypes = [C.c_void_p, C.c_int]
        L.h2g_sam_set_new_summary(h, 1)
    if options and ("--no-mixed" in options or "--no-discordant" in options):
        L.h2g_sam_set_report_policy.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.h2g_sam_set_report_policy(h, 0 if "--no-discordant" in options else 1, 0 if "--no-mixed" in options else 1)
    if options and "--no-templatelen-adjustment" in options:
        L.h2g_sam_set_templatelen_adjustment.argtypes = [C.c_void_p, C.c_int]
        L.h2g_sam_set_templatelen_adjustment(h, 0)


def _novel_out(L, h, options):
    """--novel-splicesite-outfile <path>: written from the handle's site statistics, as the command line does at the end of a run"""
    if options and "--novel-splicesite-outfile" in options:
        L.h2g_sam_novel_splice_sites_text.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.h2g_sam_novel_splice_sites_text.restype = C.c_size_t
        need = L.h2g_sam_novel_splice_sites_text(h, None, 0)
        buf = C.create_string_buffer(need + 1)
        L.h2g_sam_novel_splice_sites_text(h, buf, need)
        open(options[list(options).index("--novel-splicesite-outfile") + 1], "wb").write(buf.raw[:need])


def to_compact(aln, starts, counts):
    """records aln[starts[i] .. starts[i] + counts[i]) of every read -> (bytes, uint64 byte offsets [n + 1]) in the layout of h2g_align_*_fetch_compact:
    40 bytes of fields + 12 per edit held (a record beyond MAX_EDITS: its marker entry), rounded up to 8, the pad zeroed"""
    rs = C.sizeof(api.AlnRes)
    raw = bytes(aln) if not isinstance(aln, np.ndarray) else aln.tobytes()
    out = bytearray()
    offs = np.zeros(len(starts) + 1, dtype=np.uint64)
    for i, (s0, c) in enumerate(zip(starts, counts)):
        for k in range(int(c)):
            at = (int(s0) + k) * rs
            ne = int.from_bytes(raw[at + 24:at + 28], "little")
            e = 1 if ne > api.MAX_EDITS else ne
            out += raw[at:at + 40 + 12 * e]
            if e & 1:
                out += b"\0\0\0\0"
        offs[i + 1] = len(out)
    return bytes(out) + b"\0" * 8, offs


def _check_compact_unpaired(L, h, codes, offs, qp, nb, noffs, n, res, rp, aln, want_text):
    """the compact formatter over a host-compacted copy of the same records must write the same text (every SAM-line test is also its test)"""
    nsel = np.array([min(int(res[i].nselect) if not isinstance(res, np.ndarray) else int(res["nselect"][i]), api.ALN_CAP) for i in range(n)])
    rec, boffs = to_compact(aln, np.arange(n) * api.ALN_CAP, nsel)
    L.h2g_sam_format_unpaired_compact.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    cap = len(want_text) + 64
    buf = C.create_string_buffer(cap)
    used = C.c_size_t(0)
    rc = L.h2g_sam_format_unpaired_compact(h, codes.ctypes.data, offs.ctypes.data, qp, nb, noffs.ctypes.data, n, rp, rec, boffs.ctypes.data, buf, cap, C.byref(used))
    assert rc == 0 and buf.raw[:used.value] == want_text, "compact formatter differs from the row formatter"


def format_unpaired(L, base, reads, names, res, aln, quals=None, options=(), long_edits=None):
    """res: array of api.ReadResult (or same-layout numpy), aln: api.AlnRes * (n*ALN_CAP) -> list of SAM lines.
    long_edits = (api.Edit array, n): the batch's long-edit area (records with nedits > MAX_EDITS, h2g_align_fetch_long_edits)"""
    h = C.c_void_p()
    assert L.h2g_sam_open(base.encode(), C.byref(h)) == 0
    _score_min(L, h, options)
    if long_edits is not None and long_edits[1]:
        L.h2g_sam_set_long_edits.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.h2g_sam_set_long_edits(h, long_edits[0], long_edits[1])
    codes, offs = flat(reads)
    nb, noffs = flat_names(names)
    n = len(reads)
    cap = 600 * n * 3 + 4096 + 6 * int(offs[-1])
    buf = C.create_string_buffer(cap)
    used = C.c_size_t(0)
    rp = res.ctypes.data if isinstance(res, np.ndarray) else C.addressof(res)
    ap = aln.ctypes.data if isinstance(aln, np.ndarray) else C.addressof(aln)
    qp = None
    if quals is not None:
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        qp = quals.ctypes.data
    rc = L.h2g_sam_format_unpaired(h, codes.ctypes.data, offs.ctypes.data, qp, nb, noffs.ctypes.data, n, rp, ap, buf, cap, C.byref(used))
    if rc == 0 and not (options and ("--novel-splicesite-outfile" in options or "--new-summary" in options)) and n <= 20000:
        h2 = C.c_void_p()                                  # (a handle of its own: the summary / site statistics of `h` count every line once)
        assert L.h2g_sam_open(base.encode(), C.byref(h2)) == 0
        _score_min(L, h2, options)
        if long_edits is not None and long_edits[1]:
            L.h2g_sam_set_long_edits(h2, long_edits[0], long_edits[1])
        _check_compact_unpaired(L, h2, codes, offs, qp, nb, noffs, n, res, rp, aln, buf.raw[:used.value])
        L.h2g_sam_close(h2)
    global LAST_SUMMARY
    sb = C.create_string_buffer(4096)
    nsum = L.h2g_sam_summary(h, sb, 4096)
    LAST_SUMMARY = sb.raw[:nsum].decode()
    global LAST_HEADER
    L.h2g_sam_header.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    L.h2g_sam_header.restype = C.c_size_t
    hb = C.create_string_buffer(1 << 16)
    nhead = L.h2g_sam_header(h, b"", hb, 1 << 16)
    LAST_HEADER = hb.raw[:nhead].decode()
    _novel_out(L, h, options)
    L.h2g_sam_close(h)
    assert rc == 0, (rc, used.value, cap)
    return buf.raw[:used.value].decode().splitlines()


def format_paired(L, base, m1, m2, n1, n2, res, a1, a2, khits, options=(), dense=None):
    """dense = (offs1, offs2): a1 / a2 hold every pair's records back to back (h2g_align_pairs_fetch_dense's layout), uint64 offsets [n + 1]"""
    h = C.c_void_p()
    assert L.h2g_sam_open(base.encode(), C.byref(h)) == 0
    _score_min(L, h, options)
    c1, o1 = flat(m1)
    c2, o2 = flat(m2)
    nb1, no1 = flat_names(n1)
    nb2, no2 = flat_names(n2)
    n = len(m1)
    cap = 700 * n * 6 + 4096 + 12 * int(o1[-1] + o2[-1])
    buf = C.create_string_buffer(cap)
    used = C.c_size_t(0)
    ptr = lambda x: x.ctypes.data if isinstance(x, np.ndarray) else C.addressof(x)
    if dense is not None:
        rc = L.h2g_sam_format_paired_dense(h, c1.ctypes.data, o1.ctypes.data, None, nb1, no1.ctypes.data, c2.ctypes.data, o2.ctypes.data, None, nb2,
                                           no2.ctypes.data, n, ptr(res), ptr(a1), dense[0].ctypes.data, ptr(a2), dense[1].ctypes.data, khits, buf, cap, C.byref(used))
    else:
        rc = L.h2g_sam_format_paired(h, c1.ctypes.data, o1.ctypes.data, None, nb1, no1.ctypes.data, c2.ctypes.data, o2.ctypes.data, None, nb2,
                                     no2.ctypes.data, n, ptr(res), ptr(a1), ptr(a2), khits, buf, cap, C.byref(used))
    if rc == 0 and not (options and ("--novel-splicesite-outfile" in options or "--new-summary" in options)) and n <= 20000:
        # the compact formatter over a host-compacted copy of the same records must write the same text
        cnt = lambda r, m: int(r.nres[m])
        if dense is not None:
            s1, c1_ = dense[0][:-1], np.diff(dense[0].astype(np.int64)); s2, c2_ = dense[1][:-1], np.diff(dense[1].astype(np.int64))
        else:
            s1 = s2 = np.arange(n) * api.PAIR_RES_CAP
            c1_ = np.array([min(cnt(res[i], 0), api.PAIR_RES_CAP) for i in range(n)]); c2_ = np.array([min(cnt(res[i], 1), api.PAIR_RES_CAP) for i in range(n)])
        rec1, bo1 = to_compact(a1, s1, c1_); rec2, bo2 = to_compact(a2, s2, c2_)
        h2 = C.c_void_p()
        assert L.h2g_sam_open(base.encode(), C.byref(h2)) == 0
        _score_min(L, h2, options)
        L.h2g_sam_format_paired_compact.argtypes = [C.c_void_p] + [C.c_void_p] * 10 + [C.c_size_t, C.c_void_p, C.c_char_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t,
                                                     C.POINTER(C.c_size_t)]
        buf2 = C.create_string_buffer(used.value + 64)
        u2 = C.c_size_t(0)
        rc2 = L.h2g_sam_format_paired_compact(h2, c1.ctypes.data, o1.ctypes.data, None, nb1, no1.ctypes.data, c2.ctypes.data, o2.ctypes.data, None, nb2, no2.ctypes.data, n, ptr(res),
                                              rec1, bo1.ctypes.data, rec2, bo2.ctypes.data, khits, buf2, used.value + 64, C.byref(u2))
        L.h2g_sam_close(h2)
        assert rc2 == 0 and buf2.raw[:u2.value] == buf.raw[:used.value], "compact formatter differs from the row formatter"
    global LAST_SUMMARY
    sb = C.create_string_buffer(4096)
    nsum = L.h2g_sam_summary(h, sb, 4096)
    LAST_SUMMARY = sb.raw[:nsum].decode()
    global LAST_HEADER
    L.h2g_sam_header.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    L.h2g_sam_header.restype = C.c_size_t
    hb = C.create_string_buffer(1 << 16)
    nhead = L.h2g_sam_header(h, b"", hb, 1 << 16)
    LAST_HEADER = hb.raw[:nhead].decode()
    _novel_out(L, h, options)
    L.h2g_sam_close(h)
    assert rc == 0, (rc, used.value, cap)
    return buf.raw[:used.value].decode().splitlines()


def body_lines(path):
    return [l.rstrip("\n") for l in open(path) if not l.startswith("@")]


def emu_to_abi(outs, recs, cap=api.ALN_CAP):
    """emulator output (ReadOut + 32 AlnRec per read) -> (ReadResult array, AlnRes array) as h2g_align_fetch lays them out"""
    import sam_util as SU
    n = len(outs)
    res = (api.ReadResult * n)()
    aln = (api.AlnRes * (n * cap))()
    for i in range(n):
        o = outs[i]
        r = res[i]
        r.nres, r.nselect, r.overflow, r.nrank, r.nsteps, r.depth = o.nres, o.nselect, o.overflow, o.nrank, o.nsteps, o.depth
        r.best, r.secbest, r.best_h2, r.secbest_h2 = o.best, o.secbest, o.best_h2, o.secbest_h2
        for k in range(min(o.nselect, cap)):
            C.memmove(C.byref(aln[i * cap + k]), C.byref(recs[i * SU.AL_MAX_RESULTS + o.select[k]]), C.sizeof(api.AlnRes))
    return res, aln
