"""Shared parity checks: run a backend exposing the h2g primitive API (the HIP library on a GPU, or the host
instantiation of the same device functions on CPU) against the golden vectors of the real reference and
against the C oracle.  Used by test_emul_golden.py (CPU) and test_gpu_parity.py (GPU)."""
import ctypes as C
import gzip
import os

import numpy as np

import h2o_py as H
from hisat2_amd import api


def load_reads(golden_dir):
    names, seqs = H.read_fasta_reads(os.path.join(golden_dir, "reads_se.fa.gz"))
    L = len(seqs[0])
    arr = np.stack(seqs)
    offs = (np.arange(len(seqs) + 1, dtype=np.uint64) * L).astype(np.uint32)
    return arr, offs


def load_contigs(golden_dir, name="g1.fa.gz"):
    contigs, cur = [], []
    with gzip.open(os.path.join(golden_dir, name), "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if cur:
                    contigs.append(H.encode(b"".join(cur)))
                cur = []
            else:
                cur.append(line.strip())
    contigs.append(H.encode(b"".join(cur)))
    return contigs


def check_rank(backend_rank, golden_dir):
    rows, cs, want = [], [], []
    for l in H.glines(golden_dir, "probe_rank.txt.gz"):
        row, c, r, _ = map(int, l.split())
        rows.append(row); cs.append(c); want.append(r)
    got = backend_rank(np.array(rows, dtype=np.uint32), np.array(cs, dtype=np.uint8))
    assert np.array_equal(got, np.array(want, dtype=np.uint32))


def check_fm_search(be, golden_dir, fn, pseudo):
    qs, want = [], []
    for l in H.glines(golden_dir, fn):
        v = list(map(int, l.split()))
        qs.append(api.FmQuery(v[0], 0, v[1], 0, pseudo, 1))
        want.append(v[2:])
    out = be.fm_search(qs, khits=5)
    for o, w, q in zip(out, want, qs):
        got = [getattr(o, f) for f in api.FM_HIT_FIELDS[:13]]
        assert got == w, (q.read, q.fw, got, w)
    return len(qs)


def check_coords(be, golden_dir):
    qs, want = [], []
    for l in H.glines(golden_dir, "probe_coords.txt.gz"):
        f = l.split()
        top, bot, rdoff, hlen, strad, n = map(int, f[2:8])
        qs.append(api.SaQuery(top, bot, bot - top, hlen, 0))
        want.append((strad, [tuple(int(x) & 0xFFFFFFFF for x in f[8 + k].split(":")) for k in range(n)]))
    cap = 16
    co, res = be.sa_resolve(qs, cap=cap)
    for i, (strad, cs) in enumerate(want):
        assert res[i].ok == 1 and res[i].ncoords == len(cs) and res[i].straddled == strad
        for k, c in enumerate(cs):
            g = co[i * cap + k]
            assert (g.tidx, g.toff, g.joinedOff) == c
    return len(qs)


def check_extend(be, golden_dir):
    hits, args, want = [], [], []
    for l in H.glines(golden_dir, "probe_extend.txt.gz"):
        lhs, rhs = l.split(" -> ")
        rid, fw, rdoff, hlen, tidx, toff, joff, mm = map(int, lhs.split())
        h = api.GHit()
        h.read, h.fw, h.rdoff, h.len, h.tidx, h.toff, h.joinedOff = rid, fw, rdoff, hlen, tidx, toff, joff
        hits.append(h)
        args.append(api.ExtArgs(mm, api.MAX, api.MAX))
        want.append(rhs.split())
    out, res = be.extend(hits, args)
    for h, r, w in zip(out, res, want):
        got = [r.extended, h.rdoff, h.len, h.toff, h.joinedOff, r.leftext, r.rightext, h.score, h.nedits]
        assert got == list(map(int, w[:9])), (got, w)
        eds = [f"{h.edits[k].pos}:{chr(h.edits[k].chr)}>{chr(h.edits[k].qchr)}" for k in range(h.nedits)]
        assert eds == w[9:] and h.overflow == 0
    return len(hits)


def oracle_seed_extend(olib, oix, reads, pseudo, khits=5, cap=api.SEED_CAP):
    """The fused stage (partialSearch both strands -> coords of the first `cap` rows -> 0-mm extend) computed
    by the C oracle, as a SEED_RESULT_DTYPE array."""
    n, L = reads.shape
    out = np.zeros(n * 2, dtype=api.SEED_RESULT_DTYPE)
    sc = H.Scoring()
    olib.h2o_scoring_default(C.byref(sc))
    minK = oix.contents.minK
    qual = b"I" * L
    for r in range(n):
        for fwi in range(2):
            seq = np.ascontiguousarray(reads[r] if fwi == 0 else H.revcomp(reads[r]))
            o = H.BwtHit()
            olib.h2o_partial_search(oix, seq.ctypes.data, L, 0, pseudo, 1, khits, C.byref(o))
            rec = out[2 * r + fwi]
            for f in api.FM_HIT_FIELDS:
                rec["hit"][f] = getattr(o, f)
            if o.top == H.MAX or o.bot <= o.top or o.len <= minK + 2:
                continue
            co = (H.Coord * 64)()
            nco, st, steps = C.c_uint32(0), C.c_int(0), C.c_uint32(0)
            olib.h2o_genome_coords(oix, o.top, o.bot, min(o.bot - o.top, cap), o.len, 0, co, C.byref(nco), C.byref(st),
                                   C.byref(steps))
            rec["ncoords"], rec["straddled"], rec["nsteps"] = nco.value, st.value, steps.value
            for k in range(nco.value):
                e = rec["ext"][k]
                e["tidx"], e["toff"], e["joinedOff"] = co[k].tidx, co[k].toff, co[k].joinedOff
                e["rdoff"], e["len"], e["score"] = L - o.bwoff - o.len, o.len, 0
                if co[k].tidx == H.MAX:
                    continue
                h = H.GHit()
                h.fw, h.rdoff, h.len, h.tidx, h.toff, h.joinedOff = 1 - fwi, L - o.bwoff - o.len, o.len, co[k].tidx, co[k].toff, co[k].joinedOff
                le, re = C.c_uint32(H.MAX), C.c_uint32(H.MAX)
                olib.h2o_extend(oix, C.byref(sc), seq.ctypes.data, qual, L, C.byref(h), C.byref(le), C.byref(re), 0)
                e["toff"], e["joinedOff"], e["rdoff"], e["len"], e["score"] = h.toff, h.joinedOff, h.rdoff, h.len, h.score
    return out


def assert_seed_equal(got, want):
    for f in ("ncoords", "straddled", "nsteps"):
        assert np.array_equal(got[f], want[f]), f
    for f in api.FM_HIT_FIELDS:
        assert np.array_equal(got["hit"][f], want["hit"][f]), f
    for f in ("tidx", "toff", "joinedOff", "rdoff", "len", "score"):
        assert np.array_equal(got["ext"][f], want["ext"][f]), f


# ---------------------------------------------------------------- graph index (g1s)
def load_snp_reads(golden_dir):
    _, seqs = H.read_fasta_reads(os.path.join(golden_dir, "reads_snp.fa.gz"))
    L = len(seqs[0])
    arr = np.stack(seqs)
    offs = (np.arange(len(seqs) + 1, dtype=np.uint64) * L).astype(np.uint32)
    return arr, offs


def check_graph_rank(backend_rank, golden_dir):
    rows, cs, want = [], [], []
    for l in H.glines(golden_dir, "probe_g1s_rank.txt.gz"):
        row, c, r, _ = map(int, l.split())
        rows.append(row); cs.append(c); want.append(r)
    got = backend_rank(np.array(rows, dtype=np.uint32), np.array(cs, dtype=np.uint8))
    assert np.array_equal(got, np.array(want, dtype=np.uint32))
    return len(rows)


def check_graph_lf(be, golden_dir):
    """mapGLF (ranges, k = 10 as in the probe) and mapGLF1 (single rows) against the reference GFM"""
    qs, want = [], []
    for l in H.glines(golden_dir, "probe_g1s_glf.txt.gz"):
        f = l.split()
        qs.append(api.GlfQuery(int(f[0]), int(f[1]), int(f[2]), 0))
        want.append((tuple(map(int, f[3:7])), [tuple(map(int, x.split(":"))) for x in f[8:]]))
    res, ie = be.graph_lf(qs, k=10)
    nie = 0
    for r, e, (w, wie), q in zip(res, ie, want, qs):
        if w[0] == 0 and w[1] == 0:
            assert not r.ok and r.top == 0 and r.bot == 0, (q.top, q.bot, q.c)
            continue
        assert r.ok and (r.top, r.bot, r.node_top, r.node_bot) == w, (q.top, q.bot, q.c)
        assert e.n == len(wie) and e.pairs() == wie, (q.top, q.bot, q.c)
        nie += bool(wie)
    assert nie >= 20
    qs, want = [], []
    for l in H.glines(golden_dir, "probe_g1s_glf1.txt.gz"):
        row, c, t, b, nt, nb = map(int, l.split())
        qs.append(api.GlfQuery(row, row + 1, c, 1))
        want.append((t, b, nt, nb))
    res, _ = be.graph_lf(qs, k=10)
    wide = 0
    for r, w, q in zip(res, want, qs):
        if w[0] == 0 and w[1] == 0:
            assert not r.ok, (q.top, q.c)
        else:
            assert r.ok and (r.top, r.bot, r.node_top, r.node_bot) == w, (q.top, q.c)
            wide += (w[1] - w[0]) > 1
    assert wide >= 5
    return len(qs)


def check_graph_fm_search(be, golden_dir, fn="probe_g1s_psearch.txt.gz"):
    qs, want = [], []
    for l in H.glines(golden_dir, fn):
        f = l.split()
        v = list(map(int, f[:15]))
        qs.append(api.FmQuery(v[0], 0, v[1], 0, 0, 1))
        want.append((v[2:], [tuple(map(int, x.split(":"))) for x in f[16:]]))
    out, ie = be.fm_search_graph(qs, khits=10, kseeds=20)
    for o, e, (w, wie), q in zip(out, ie, want, qs):
        got = [getattr(o, f) for f in api.FM_HIT_FIELDS[:13]]
        assert got == w, (q.read, q.fw, got, w)
        assert e.pairs() == wie
    return len(qs)


# ---------------------------------------------------------------- Smith-Waterman (a23-a25)
def parse_sw_probe(golden_dir, fn="probe_sw.txt.gz"):
    """-> list of dicts: the reference SwAligner outcome for every (read, strand, seed coordinate) of probe_sw.txt.gz (probe_sw16.txt.gz:
    the vectors of its 16-bit path, --score-min below -254)"""
    out = []
    for l in H.glines(golden_dir, fn):
        f = l.split()
        d = dict(zip(("rid", "fw", "k", "tidx", "refoff", "minsc"), map(int, f[:6])))
        d["rect"] = list(map(int, f[7:13]))
        d["found_align"], d["best"], d["found"] = int(f[14]), int(f[15]), int(f[16])
        rest = f[17:]
        d["rnd_next"] = int(rest[-1][1:])
        if d["found"]:
            d["score"], d["off"], ne = int(rest[0]), int(rest[1]), int(rest[2])
            d["edits"] = rest[3:3 + ne]
        out.append(d)
    return out


def sw_edit_strings(edits, nedits, fw, rdlen):
    """the reference prints AlnRes::ned() after setShape, i.e. w.r.t. the 5' end: invert ours for the rc strand"""
    def s(e, pos):
        return f"{pos}:{chr(e.chr)}>{chr(e.qchr)}:{e.type}"
    if fw:
        return [s(edits[i], edits[i].pos) for i in range(nedits)]
    return [s(edits[i], (rdlen - edits[i].pos) if edits[i].type == 1 else (rdlen - edits[i].pos - 1)) for i in reversed(range(nedits))]


def load_sw_reads(golden_dir, fn="reads_sw.fa.gz"):
    _, seqs = H.read_fasta_reads(os.path.join(golden_dir, fn))
    L = len(seqs[0])
    arr = np.stack(seqs)
    offs = (np.arange(len(seqs) + 1, dtype=np.uint64) * L).astype(np.uint32)
    return arr, offs


def check_sw(be, golden_dir, rdlen=101, fn="probe_sw.txt.gz"):
    """backend.sw_align against the reference SwAligner vectors.  An alignment with more edits than a record holds (H2G_MAX_EDITS; only the
    16-bit vectors have such) must be flagged, with score and offset still the reference's."""
    cases = parse_sw_probe(golden_dir, fn)
    qs = [api.SwQuery(d["rid"], d["fw"], d["tidx"], d["refoff"], d["minsc"], (d["rid"] * 7 + d["k"] + 1) & 0xFFFFFFFF) for d in cases]
    out, _ = be.sw_align(qs)
    nfound = nbig = 0
    for d, o in zip(cases, out):
        big = bool(d["found"]) and len(d["edits"]) > api.MAX_EDITS
        assert bool(o.overflow) == big, d
        assert [o.refl, o.refr] == d["rect"][:2], d
        assert (o.found_align, o.best, o.found) == (d["found_align"], d["best"], d["found"]), d
        assert H.lcg_next(o.rnd)[0] == d["rnd_next"], d
        if d["found"]:
            assert (o.score, o.off) == (d["score"], d["off"]), d
            if not big:
                assert sw_edit_strings(o.edits, o.nedits, d["fw"], rdlen) == d["edits"], d
            nfound += 1
            nbig += big
    assert nfound > 100
    return len(cases), nbig


def parse_graph_coords(golden_dir, fn):
    """-> [(top, bot, node_top, node_bot, iedges, hitlen, straddled, [(tidx, toff, joinedOff)])]"""
    out = []
    for l in H.glines(golden_dir, fn):
        lhs, rhs = l.split(" | ")
        f, r = lhs.split(), rhs.split()
        top, bot, rdoff, hlen, strad, nco = map(int, f[2:8])
        want = [tuple(int(x) & 0xFFFFFFFF for x in c.split(":")) for c in f[8:8 + nco]]
        nt, nb, nie = map(int, r[:3])
        ie = [tuple(map(int, x.split(":"))) for x in r[3:]]
        assert len(ie) == nie
        out.append((top, bot, nt, nb, ie, hlen, strad, want))
    return out


def check_graph_coords(be, golden_dir, fn):
    cases = parse_graph_coords(golden_dir, fn)
    qs, ies = [], []
    for top, bot, nt, nb, ie, hlen, strad, want in cases:
        qs.append(api.GsaQuery(top, bot, nt, nb, bot - top, hlen, 0))
        x = api.IEdges()
        x.n = len(ie)
        for k, (a, b) in enumerate(ie):
            x.e[k][0], x.e[k][1] = a, b
        ies.append(x)
    cap = 24
    co, res = be.sa_resolve_graph(qs, ies, cap=cap)
    nmulti = 0
    for i, (top, bot, nt, nb, ie, hlen, strad, want) in enumerate(cases):
        assert res[i].ok == 1 and res[i].ncoords == len(want) and res[i].straddled == strad, (top, bot, res[i].ok, res[i].nsteps)
        assert [(co[i * cap + k].tidx, co[i * cap + k].toff, co[i * cap + k].joinedOff) for k in range(len(want))] == want, (top, bot)
        nmulti += len(want) > 1
    return len(cases), nmulti


def check_graph_extend(be, golden_dir):
    """GenomeHit::extend on the graph index: ALT-aware (edit strings carry type and snpID)"""
    hits, args, want = [], [], []
    for l in H.glines(golden_dir, "probe_g1s_extend.txt.gz"):
        lhs, rhs = l.split(" -> ")
        rid, fw, rdoff, hlen, tidx, toff, joff, mm = map(int, lhs.split())
        h = api.GHit()
        h.read, h.fw, h.rdoff, h.len, h.tidx, h.toff, h.joinedOff = rid, fw, rdoff, hlen, tidx, toff, joff
        hits.append(h)
        args.append(api.ExtArgs(mm, api.MAX, api.MAX))
        want.append(rhs.split())
    out, res = be.extend(hits, args)
    nsnp = 0
    for h, r, w in zip(out, res, want):
        got = [r.extended, h.rdoff, h.len, h.toff, h.joinedOff, r.leftext, r.rightext, h.score, h.nedits]
        assert got == list(map(int, w[:9])), (got, w)
        eds = [f"{h.edits[k].pos}:{chr(h.edits[k].chr)}>{chr(h.edits[k].qchr)}:{h.edits[k].type}:{-1 if h.edits[k].snp == api.MAX else h.edits[k].snp}"
               for k in range(h.nedits)]
        assert eds == w[9:] and h.overflow == 0, (eds, w)
        nsnp += any(not e.endswith(":-1") for e in eds)
    assert nsnp > 80
    return len(hits)


def parse_adjust(golden_dir, fn):
    """-> [(rid, fw, rdoff, len, tidx, toff, joff, found, [([rdoff, len, toff, joff, nedits], [edit strings])])]"""
    out = []
    for l in H.glines(golden_dir, fn):
        lhs, rhs = l.split(" -> ")
        parts = rhs.split(" | ")
        found, nh = map(int, parts[0].split())
        want = []
        for p in parts[1:]:
            f = p.split()
            want.append((list(map(int, f[:5])), f[5:]))
        assert len(want) == nh
        out.append(tuple(map(int, lhs.split())) + (found, want))
    return out


def edit_strings_snp(edits, n, MAXV=0xFFFFFFFF):
    return [f"{edits[e].pos}:{chr(edits[e].chr)}>{chr(edits[e].qchr)}:{edits[e].type}:{-1 if edits[e].snp == MAXV else edits[e].snp}" for e in range(n)]


def check_graph_adjust(be, golden_dir, fn):
    cases = parse_adjust(golden_dir, fn)
    qs = [api.AdjustQuery(c[0], c[1], c[2], c[3], c[4], c[5], c[6]) for c in cases]
    cap = 8
    hits, nh = be.adjust_with_alt(qs, cap=cap)
    for i, c in enumerate(cases):
        found, want = c[7], c[8]
        assert nh[i] == len(want) and (nh[i] > 0) == bool(found), (c[:7], nh[i])
        got = [([hits[i * cap + k].rdoff, hits[i * cap + k].len, hits[i * cap + k].toff, hits[i * cap + k].joinedOff, hits[i * cap + k].nedits],
                edit_strings_snp(hits[i * cap + k].edits, hits[i * cap + k].nedits)) for k in range(nh[i])]
        assert got == want, (c[:7], got, want)
    return len(cases)


def check_ext_search(search, local_index_of, golden_dir, fn):
    """globalGFMSearch / localGFMSearch against the vectors of the real classes (tests/gen_golden.py extsearch).
    search(queries) -> hits; local_index_of(tidx, toff) -> lidx"""
    qs, want = [], []
    for l in H.glines(golden_dir, fn):
        v = list(map(int, l.split()))
        q = api.ExtSearchQuery()
        q.read, q.fw, q.rdoff = v[0], v[1], v[2]
        q.lidx = 0xffffffff if v[3] == 0 else local_index_of(v[4], v[5])
        q.maxHitLen, q.uniqueStop = (0xffffffff if v[3] == 0 else v[6]), v[7]
        if v[3] == 1 and q.lidx == 0xffffffff:
            assert v[8] == 0            # no local index there: the reference probe reports nothing
            continue
        qs.append(q); want.append(v[8:])
    got = search(qs)
    nel = 0
    for q, g, w in zip(qs, got, want):
        assert g.nelt == w[0] and bool(g.uniqueStop) == bool(w[4]), (q.read, q.fw, q.rdoff, q.lidx, g.nelt, g.uniqueStop, w)
        if w[0] > 0:
            nel += 1
            assert (g.hitlen, g.top, g.bot) == (w[1], w[2], w[3]), (q.read, q.fw, q.rdoff, q.lidx, g.hitlen, g.top, g.bot, w)
    return len(qs), nel
