"""CPU tests: the C oracle (oracle/h2o.c) against vectors emitted by the REAL reference
classes (oracle/ref_probe.cpp run on the reference-built index of genome g1)."""
import ctypes as C
import os

import numpy as np
import pytest

import h2o_py as H


def test_params(oracle_lib, g1_index, golden_dir):
    ix = H.load_index(oracle_lib, g1_index)
    kv = H.glines(golden_dir, "probe_params.txt.gz")[0].split()
    d = dict(zip(kv[0::2], map(int, kv[1::2])))
    p = ix.contents.g.p
    for k in ("len", "gbwtLen", "numNodes", "lineRate", "offRate", "ftabChars", "eftabLen", "linear", "sideSz",
              "sideGbwtSz", "sideGbwtLen", "numSides", "offsLen"):
        assert getattr(p, k) == d[k], k
    assert ix.contents.g.nPat == d["nPat"] and ix.contents.g.nFrag == d["nFrag"]


def test_rank_rowL(oracle_lib, g1_index, golden_dir):
    ix = H.load_index(oracle_lib, g1_index)
    g = C.byref(ix.contents.g)
    for l in H.glines(golden_dir, "probe_rank.txt.gz"):
        row, c, r, rl = map(int, l.split())
        assert oracle_lib.h2o_rank(g, row, c) == r
        assert oracle_lib.h2o_rowL(g, row) == rl


def test_ftab(oracle_lib, g1_index, golden_dir):
    ix = H.load_index(oracle_lib, g1_index)
    g = C.byref(ix.contents.g)
    nhit = 0
    for l in H.glines(golden_dir, "probe_ftab.txt.gz"):
        s, ok, top, bot = l.split()
        seq = H.encode(s.encode())
        t, b = C.c_uint32(0), C.c_uint32(0)
        got = oracle_lib.h2o_ftab_lohi(g, seq.ctypes.data, 0, C.byref(t), C.byref(b))
        assert got == int(ok)
        if got:
            assert (t.value, b.value) == (int(top), int(bot))
            nhit += int(bot) > int(top)
    assert nhit > 50


def test_offset_and_joined_to_text(oracle_lib, g1_index, golden_dir):
    ix = H.load_index(oracle_lib, g1_index)
    g = C.byref(ix.contents.g)
    for l in H.glines(golden_dir, "probe_offset.txt.gz"):
        row, off, qlen, ok, tidx, toff, tlen, strad = map(int, l.split())
        st = C.c_uint32(0)
        assert oracle_lib.h2o_get_offset(g, row, C.byref(st)) == off
        if off < ix.contents.g.p.len:
            a, b, c_, s = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_int(0)
            got = oracle_lib.h2o_joined_to_text(g, qlen, off, C.byref(a), C.byref(b), C.byref(c_), 0, C.byref(s))
            assert (got, a.value, b.value, c_.value, s.value) == (ok, tidx, toff, tlen, strad)


def test_stretch(oracle_lib, g1_index, golden_dir):
    ix = H.load_index(oracle_lib, g1_index)
    r = C.byref(ix.contents.r)
    for l in H.glines(golden_dir, "probe_stretch.txt.gz"):
        tidx, toff, cnt, s = l.split()
        buf = np.zeros(int(cnt), dtype=np.uint8)
        oracle_lib.h2o_get_stretch(r, int(tidx), int(toff), int(cnt), buf.ctypes.data)
        assert bytes(b"ACGTN"[x] for x in buf) == s.encode()


def _reads(golden_dir):
    import os
    return H.read_fasta_reads(os.path.join(golden_dir, "reads_se.fa.gz"))


def _psearch(oracle_lib, golden_dir, g1_index, fn, pseudo):
    ix = H.load_index(oracle_lib, g1_index)
    _, seqs = _reads(golden_dir)
    n = 0
    for l in H.glines(golden_dir, fn):
        v = list(map(int, l.split()))
        rid, fw = v[0], v[1]
        seq = seqs[rid] if fw else H.revcomp(seqs[rid])
        seq = np.ascontiguousarray(seq)
        o = H.BwtHit()
        oracle_lib.h2o_partial_search(ix, seq.ctypes.data, len(seq), 0, pseudo, 1, 5, C.byref(o))
        got = [o.top, o.bot, o.node_top, o.node_bot, o.bwoff, o.len, o.hit_type, o.cur, o.done,
               o.numPartialSearch, o.numUniqueSearch, o.pseudogeneStop, o.anchorStop]
        assert got == v[2:], (rid, fw, got, v[2:])
        n += 1
    assert n == 800


def test_partial_search(oracle_lib, g1_index, golden_dir):
    _psearch(oracle_lib, golden_dir, g1_index, "probe_psearch.txt.gz", 0)


def test_partial_search_pseudogene_stop(oracle_lib, g1_index, golden_dir):
    _psearch(oracle_lib, golden_dir, g1_index, "probe_psearch_spliced.txt.gz", 1)


def test_genome_coords(oracle_lib, g1_index, golden_dir):
    ix = H.load_index(oracle_lib, g1_index)
    for l in H.glines(golden_dir, "probe_coords.txt.gz"):
        f = l.split()
        top, bot, rdoff, hlen, strad, n = map(int, f[2:8])
        co = (H.Coord * 32)()
        nco, st, steps = C.c_uint32(0), C.c_int(0), C.c_uint32(0)
        oracle_lib.h2o_genome_coords(ix, top, bot, bot - top, hlen, 0, co, C.byref(nco), C.byref(st), C.byref(steps))
        assert nco.value == n and st.value == strad
        for k in range(n):
            t, o, j = map(int, f[8 + k].split(":"))
            t &= 0xFFFFFFFF
            assert (co[k].tidx, co[k].toff, co[k].joinedOff) == (t, o, j)


def test_extend(oracle_lib, g1_index, golden_dir):
    ix = H.load_index(oracle_lib, g1_index)
    _, seqs = _reads(golden_dir)
    sc = H.Scoring()
    oracle_lib.h2o_scoring_default(C.byref(sc))
    n = 0
    for l in H.glines(golden_dir, "probe_extend.txt.gz"):
        lhs, rhs = l.split(" -> ")
        rid, fw, rdoff, hlen, tidx, toff, joff, mm = map(int, lhs.split())
        r = rhs.split()
        seq = np.ascontiguousarray(seqs[rid] if fw else H.revcomp(seqs[rid]))
        h = H.GHit()
        h.fw, h.rdoff, h.len, h.tidx, h.toff, h.joinedOff = fw, rdoff, hlen, tidx, toff, joff
        le, re = C.c_uint32(H.MAX), C.c_uint32(H.MAX)
        qual = b"I" * len(seq)
        ext = oracle_lib.h2o_extend(ix, C.byref(sc), seq.ctypes.data, qual, len(seq), C.byref(h), C.byref(le),
                                    C.byref(re), mm)
        got = [ext, h.rdoff, h.len, h.toff, h.joinedOff, le.value, re.value, h.score, h.nedits]
        assert got == list(map(int, r[:9])), (l, got)
        eds = [f"{h.edits[k].pos}:{chr(h.edits[k].chr)}>{chr(h.edits[k].qchr)}" for k in range(h.nedits)]
        assert eds == r[9:], (l, eds)
        n += 1
    assert n > 1000


# ---------------------------------------------------------------- graph index (g1s): a2 / a7 / a9
def parse_glf_line(l):
    f = l.split()
    top, bot, c = int(f[0]), int(f[1]), int(f[2])
    exp = tuple(map(int, f[3:7]))
    ie = [tuple(map(int, x.split(":"))) for x in f[8:]]
    assert len(ie) == int(f[7])
    return top, bot, c, exp, ie


def test_graph_params_and_rank(oracle_lib, g1s_index, golden_dir):
    ix = H.load_index(oracle_lib, g1s_index)
    p = ix.contents.g.p
    kv = H.glines(golden_dir, "probe_g1s_params.txt.gz")[0].split()
    d = dict(zip(kv[0::2], map(int, kv[1::2])))
    assert not p.linear and d["linear"] == 0
    for k in ("len", "gbwtLen", "numNodes", "sideSz", "sideGbwtSz", "sideGbwtLen", "numSides", "offsLen"):
        assert getattr(p, k) == d[k], k
    g = C.byref(ix.contents.g)
    for l in H.glines(golden_dir, "probe_g1s_rank.txt.gz"):
        r, c, v, rl = map(int, l.split())
        assert oracle_lib.h2o_rank(g, r, c) == v and oracle_lib.h2o_rowL(g, r) == rl


def test_graph_map_glf(oracle_lib, g1s_index, golden_dir):
    """mapGLF on ranges (rank_M, select_F, getInEdgeCount) against the reference GFM"""
    ix = H.load_index(oracle_lib, g1s_index)
    g = C.byref(ix.contents.g)
    u32 = C.c_uint32
    nie = 0
    for l in H.glines(golden_dir, "probe_g1s_glf.txt.gz"):
        top, bot, c, exp, ie = parse_glf_line(l)
        a, b, na, nb, n = u32(), u32(), u32(), u32(), u32()
        buf = (u32 * 128)()
        ok = oracle_lib.h2o_map_glf(g, top, bot, c, 10, a, b, na, nb, buf, 64, n)
        if exp[0] == 0 and exp[1] == 0:
            assert not ok and a.value == 0 and b.value == 0, l
            continue
        assert ok and (a.value, b.value, na.value, nb.value) == exp, l
        assert [(buf[2 * i], buf[2 * i + 1]) for i in range(n.value)] == ie, l
        nie += bool(ie)
    assert nie >= 20


def test_graph_map_glf1(oracle_lib, g1s_index, golden_dir):
    ix = H.load_index(oracle_lib, g1s_index)
    g = C.byref(ix.contents.g)
    u32 = C.c_uint32
    wide = 0
    for l in H.glines(golden_dir, "probe_g1s_glf1.txt.gz"):
        row, c, t, b, nt, nb = map(int, l.split())
        a, bb, na, nbb = u32(), u32(), u32(), u32()
        ok = oracle_lib.h2o_map_glf1(g, row, c, a, bb, na, nbb)
        if t == 0 and b == 0:
            assert not ok, l
        else:
            assert ok and (a.value, bb.value, na.value, nbb.value) == (t, b, nt, nb), l
            wide += (b - t) > 1
    assert wide >= 5


def _psearch_graph(oracle_lib, golden_dir, g1s_index, fn, pseudo):
    ix = H.load_index(oracle_lib, g1s_index)
    _, seqs = H.read_fasta_reads(os.path.join(golden_dir, "reads_snp.fa.gz"))
    u32 = C.c_uint32
    n = 0
    for l in H.glines(golden_dir, fn):
        f = l.split()
        v = list(map(int, f[:15]))
        ie = [tuple(map(int, x.split(":"))) for x in f[16:]]
        rid, fw = v[0], v[1]
        seq = np.ascontiguousarray(seqs[rid] if fw else H.revcomp(seqs[rid]))
        o = H.BwtHit()
        buf = (u32 * 128)()
        nie = u32()
        oracle_lib.h2o_partial_search_graph(ix, seq.ctypes.data, len(seq), 0, pseudo, 1, 10, 20, C.byref(o), buf, 64, nie)
        got = [o.top, o.bot, o.node_top, o.node_bot, o.bwoff, o.len, o.hit_type, o.cur, o.done,
               o.numPartialSearch, o.numUniqueSearch, o.pseudogeneStop, o.anchorStop]
        assert got == v[2:], (rid, fw, got, v[2:])
        assert [(buf[2 * i], buf[2 * i + 1]) for i in range(nie.value)] == ie
        n += 1
    assert n == 600


def test_graph_partial_search(oracle_lib, g1s_index, golden_dir):
    _psearch_graph(oracle_lib, golden_dir, g1s_index, "probe_g1s_psearch.txt.gz", 0)


def test_graph_partial_search_spliced_mode(oracle_lib, g1s_index, golden_dir):
    # pseudogeneStop is only ever armed on linear indexes (hi_aligner.h:4669), so spliced mode == no-spliced here
    _psearch_graph(oracle_lib, golden_dir, g1s_index, "probe_g1s_psearch_spliced.txt.gz", 0)


# ---------------------------------------------------------------- Smith-Waterman (a23-a25)
@pytest.mark.parametrize("reads_fn,probe_fn", [("reads_sw.fa.gz", "probe_sw.txt.gz"), ("reads_sw16.fa.gz", "probe_sw16.txt.gz")])
def test_sw_align_matches_reference_swaligner(oracle_lib, g1_index, golden_dir, reads_fn, probe_fn):
    """frame + end-to-end fill + gather + first nextAlignment (incl. its PRNG reseeding) == SwAligner: the u8 path, and (probe_sw16: minsc
    below -254) the i16 path"""
    import parity_cases as PC
    ix = H.load_index(oracle_lib, g1_index)
    sc = H.Scoring()
    oracle_lib.h2o_scoring_default(C.byref(sc))
    _, seqs = H.read_fasta_reads(os.path.join(golden_dir, reads_fn))
    cases = PC.parse_sw_probe(golden_dir, probe_fn)
    nfound = ngap = nbig = 0
    for d in cases:
        seq = np.ascontiguousarray(seqs[d["rid"]] if d["fw"] else H.revcomp(seqs[d["rid"]]))
        rnd = C.c_uint32((d["rid"] * 7 + d["k"] + 1) & 0xFFFFFFFF)
        o = H.SwResult()
        oracle_lib.h2o_sw_align(ix, C.byref(sc), seq.ctypes.data, None, len(seq), d["tidx"], d["refoff"], d["minsc"], int(0.15 * len(seq)), 4,
                                C.byref(rnd), C.byref(o))
        assert [o.refl, o.refr, o.refl_pretrim, o.refr_pretrim, o.corel, o.corer] == d["rect"], d
        assert (o.found_align, o.best, o.found) == (d["found_align"], d["best"], d["found"]), d
        assert H.lcg_next(rnd.value)[0] == d["rnd_next"], d
        if d["found"]:
            assert (o.score, o.off) == (d["score"], d["off"]), d
            big = len(d["edits"]) > len(o.edits)                 # more edits than the oracle's record holds: flagged, score / offset still exact
            assert bool(o.overflow) == big, d
            if not big:
                assert PC.sw_edit_strings(o.edits, o.nedits, d["fw"], len(seq)) == d["edits"], d
            nfound += 1
            nbig += big
            ngap += any(e.split(":")[2] in ("1", "2") for e in d["edits"])
    assert len(cases) > 250 and nfound > 100 and ngap > 50
    if probe_fn == "probe_sw16.txt.gz":
        assert min(d["score"] for d in cases if d["found"]) < -300 and nbig > 50


@pytest.mark.parametrize("fn", ["probe_g1s_coords.txt.gz", "probe_g1s_coords_short.txt.gz"])
def test_graph_genome_coords(oracle_lib, g1s_index, golden_dir, fn):
    """the node-based group walk (GWState on a graph index) incl. ref/alt duplicate nodes and in-edge lists"""
    import parity_cases as PC
    ix = H.load_index(oracle_lib, g1s_index)
    u32 = C.c_uint32
    nmulti = 0
    for top, bot, nt, nb, ie, hlen, strad, want in PC.parse_graph_coords(golden_dir, fn):
        buf = (u32 * (2 * max(1, len(ie))))()
        for k, (a, b) in enumerate(ie):
            buf[2 * k], buf[2 * k + 1] = a, b
        co = (H.Coord * 64)()
        nc, st, steps = u32(0), C.c_int(0), u32(0)
        rc = oracle_lib.h2o_genome_coords_graph(ix, top, bot, nt, nb, buf, len(ie), bot - top, hlen, 0, co, C.byref(nc), C.byref(st),
                                                C.byref(steps))
        assert rc == 1 and st.value == strad
        assert [(co[k].tidx, co[k].toff, co[k].joinedOff) for k in range(nc.value)] == want, (top, bot)
        nmulti += len(want) > 1
    assert nmulti >= (100 if "short" in fn else 4)


def test_graph_extend_with_alts(oracle_lib, g1s_index, golden_dir):
    """GenomeHit::extend on a graph index: alignWithALTs_recur through known SNPs / insertions / deletions (edits carry snpID
    and cost nothing), mm = 0..3"""
    ix = H.load_index(oracle_lib, g1s_index)
    assert ix.contents.nalts > 500     # 508 ALTs + one reversed copy per deletion (gfm.h:879-885)
    _, seqs = H.read_fasta_reads(os.path.join(golden_dir, "reads_snp.fa.gz"))
    sc = H.Scoring()
    oracle_lib.h2o_scoring_default(C.byref(sc))
    n = nsnp = ngap = 0
    for l in H.glines(golden_dir, "probe_g1s_extend.txt.gz"):
        lhs, rhs = l.split(" -> ")
        rid, fw, rdoff, hlen, tidx, toff, joff, mm = map(int, lhs.split())
        r = rhs.split()
        seq = np.ascontiguousarray(seqs[rid] if fw else H.revcomp(seqs[rid]))
        h = H.GHit()
        h.fw, h.rdoff, h.len, h.tidx, h.toff, h.joinedOff = fw, rdoff, hlen, tidx, toff, joff
        le, re = C.c_uint32(H.MAX), C.c_uint32(H.MAX)
        ext = oracle_lib.h2o_extend(ix, C.byref(sc), seq.ctypes.data, b"I" * len(seq), len(seq), C.byref(h), C.byref(le), C.byref(re), mm)
        got = [ext, h.rdoff, h.len, h.toff, h.joinedOff, le.value, re.value, h.score, h.nedits]
        assert got == list(map(int, r[:9])), (l, got)
        eds = [f"{h.edits[k].pos}:{chr(h.edits[k].chr)}>{chr(h.edits[k].qchr)}:{h.edits[k].type}:{-1 if h.edits[k].snp == H.MAX else h.edits[k].snp}"
               for k in range(h.nedits)]
        assert eds == r[9:], (l, eds)
        n += 1
        nsnp += any(not e.endswith(":-1") for e in eds)
        ngap += any(e.split(":")[2] in ("1", "2") for e in eds)
    assert n > 1000 and nsnp > 80 and ngap > 10


@pytest.mark.parametrize("fn,reads", [("probe_g1s_adjust.txt.gz", "reads_snp.fa.gz"), ("probe_g1s_adjust_short.txt.gz", "reads_snp_short.fa.gz")])
def test_graph_adjust_with_alt(oracle_lib, g1s_index, golden_dir, fn, reads):
    """static GenomeHit::adjustWithALT: anchor coordinates corrected for indel ALTs (findOffDiffs), known variants written as edits"""
    import parity_cases as PC
    ix = H.load_index(oracle_lib, g1s_index)
    _, seqs = H.read_fasta_reads(os.path.join(golden_dir, reads))
    u32 = C.c_uint32
    nshift = nsnp = 0
    for rid, fw, rdoff, ln, tidx, toff, joff, found, want in PC.parse_adjust(golden_dir, fn):
        seq = np.ascontiguousarray(seqs[rid] if fw else H.revcomp(seqs[rid]))
        hits = (H.GHit * 8)()
        nh = u32(0)
        r = oracle_lib.h2o_adjust_with_alt(ix, seq.ctypes.data, fw, rdoff, ln, tidx, toff, joff, hits, C.byref(nh), 8)
        got = [([hits[k].rdoff, hits[k].len, hits[k].toff, hits[k].joinedOff, hits[k].nedits], PC.edit_strings_snp(hits[k].edits, hits[k].nedits))
               for k in range(nh.value)]
        assert r == found and got == want, (rid, fw, got, want)
        nshift += any(w[0][2] != toff for w in want)
        nsnp += any(w[0][4] > 0 for w in want)
    assert nsnp > 10 and (nshift > 3 or "short" not in fn)


def test_local_graph_lf(oracle_lib, g1s_index, golden_dir):
    """mapGLF / mapGLF1 on the LOCAL graph indexes (u16 words, 232 symbols per 128 B side)"""
    ix = H.load_index(oracle_lib, g1s_index)
    x = ix.contents
    u32 = C.c_uint32
    n = nie = 0
    for l in H.glines(golden_dir, "probe_g1s_lglf.txt.gz"):
        f = l.split()
        single, tidx, toff, top, bot, c = map(int, f[:6])
        exp = tuple(map(int, f[6:10]))
        ie = [tuple(map(int, e.split(":"))) for e in f[11:]]
        g = C.byref(x.local[x.local_first[tidx] + toff // 56320])
        a, b, na, nb, nn = u32(), u32(), u32(), u32(), u32()
        buf = (u32 * 128)()
        if single:
            r = oracle_lib.h2o_map_glf1(g, top, c, a, b, na, nb)
            assert (not r and exp[:2] == (0, 0)) or (r and (a.value, b.value, na.value, nb.value) == exp), l
        else:
            oracle_lib.h2o_map_glf(g, top, bot, c, 10, a, b, na, nb, buf, 64, nn)
            if exp[:2] == (0, 0):
                assert (a.value, b.value) == (0, 0), l
            else:
                assert (a.value, b.value, na.value, nb.value) == exp and [(buf[2 * i], buf[2 * i + 1]) for i in range(nn.value)] == ie, l
                nie += bool(ie)
        n += 1
    assert n == 6000 and nie >= 3


def test_sw_reference_known_answer_cases(oracle_lib, golden_dir):
    """the reference's own SwAligner end-to-end known-answer cases (aligner_sw.cpp:1470-2727; not runnable there any more, lifted into
    tests/golden/sw_kat.json by tests/gen_sw_kat.py): score, reference offset, extent, gaps and Ns of the first alignment"""
    import sw_kat as K
    oracle_lib.h2o_sw_align_window.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int64, C.c_int64, C.c_int64,
                                               C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    cases = K.load(golden_dir)
    nfound = nwide = 0
    for case in cases:
        s = case["scoring"]
        sc = H.Scoring()
        oracle_lib.h2o_scoring_default(C.byref(sc))
        sc.mmpMax, sc.mmpMin = K.mm_range(s)
        sc.nPen, sc.rdGapConst, sc.rdGapLinear, sc.rfGapConst, sc.rfGapLinear = s["npen"], s["rdGapConst"], s["rdGapLinear"], s["rfGapConst"], s["rfGapLinear"]
        seq, ref = K.codes(case["read"]), K.codes(case["ref"])
        refl, refr, corel, corer = K.window(case)
        rnd, ns, o = C.c_uint32(1), C.c_int(0), H.SwResult()
        oracle_lib.h2o_sw_align_window(C.byref(sc), seq.ctypes.data, case["qual"].encode(), len(seq), ref.ctypes.data, len(ref), refl, refr, corel, corer, case["minsc"],
                                       10 ** 6 if case["nceil"] is None else case["nceil"], s["gapbar"], C.byref(rnd), C.byref(o), C.byref(ns))
        K.check(case, dict(found=o.found, score=o.score, off=o.off, gaps=o.gaps, ns=ns.value, edits=[(o.edits[k].type, o.edits[k].pos) for k in range(o.nedits)]))
        nfound += o.found
        nwide += case["minsc"] < -254
    assert len(cases) == 105 and nfound > 60 and nwide == 1
