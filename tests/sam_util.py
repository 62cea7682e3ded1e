"""SAM helpers for the go()-level parity tests: parse the reference's SAM and render the records returned by the
hot path (AlnRec = what reportHit hands to AlnRes::init) into FLAG / RNAME / POS / CIGAR / AS:i."""
import ctypes as C
import gzip

import numpy as np

from hisat2_amd import api

AL_MAX_RESULTS = 32


class AlnRec(C.Structure):
    _fields_ = [("fw", C.c_uint32), ("tidx", C.c_uint32), ("toff", C.c_uint32), ("len", C.c_uint32), ("trim5", C.c_uint32),
                ("trim3", C.c_uint32), ("nedits", C.c_uint32), ("pad", C.c_uint32), ("score", C.c_int64),
                ("edits", api.Edit * api.MAX_EDITS)]


class ReadOut(C.Structure):
    _fields_ = [("nres", C.c_uint32), ("nselect", C.c_uint32), ("overflow", C.c_uint32), ("nrank", C.c_uint32),
                ("nsteps", C.c_uint32), ("depth", C.c_uint32), ("nside", C.c_uint32), ("best", C.c_int32), ("secbest", C.c_int32),
                ("best_h2", C.c_uint32), ("secbest_h2", C.c_uint32), ("select", C.c_uint8 * AL_MAX_RESULTS)]


def parse_sam(path):
    """-> (refnames, {qname: [(flag, rname, pos, cigar, AS), ...] in file order})"""
    op = gzip.open if path.endswith(".gz") else open
    names, recs = [], {}
    with op(path, "rt") as f:
        for line in f:
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


def cigar_of(rec: AlnRec, rd):
    """CIGAR in reference-forward orientation from the stored edit list (5'-relative, inverted when !fw).  `rd` = the read
    as base codes (forward strand): the SAM printer's own gap left-alignment is then reproduced (StackedAln::leftAlign,
    aligner_result.cpp:746, called from aln_sink.h:3048) — it slides a non-SNP gap left over matching bases until it meets
    another gap, which the aligner's GenomeHit::leftAlign does not do next to an ALT gap.  An int (read length) skips it."""
    eds = [(rec.edits[k].pos, rec.edits[k].type, chr(rec.edits[k].chr), rec.edits[k].snp != api.MAX and rec.edits[k].type != 5) for k in range(rec.nedits)]
    # splice edits (type 5, CIGAR N): splLen = chr | qchr << 8 | (pad & 15) << 16 (include/h2g.h)
    skips = [rec.edits[k].chr | (rec.edits[k].qchr << 8) | ((rec.edits[k].pad & 15) << 16) for k in range(rec.nedits) if rec.edits[k].type == 5]
    if not rec.fw:   # stored 5'->3' along the original read, relative to the first aligned base: mirror within len
        eds = [((rec.len - p) if t in (1, 5) else (rec.len - p - 1), t, c, sn) for p, t, c, sn in reversed(eds)]
        skips = skips[::-1]
    rel, snp, refc = [], [], []     # StackedAln::init (aligner_result.cpp:660-728)
    seq = None
    if not isinstance(rd, (int, np.integer)):
        seq = np.asarray(rd)
        if not rec.fw:
            seq = np.where(seq[::-1] < 4, 3 - seq[::-1], 4)
    ei = 0
    for i in range(rec.len):
        consumed = False
        while ei < len(eds) and eds[ei][0] == i and not consumed:
            p, t, c, sn = eds[ei]
            if t == 5:
                rel.append("N"); snp.append(False); refc.append("N")
            elif t == 1:
                rel.append("D"); snp.append(sn); refc.append(c)
            elif t == 2:
                rel.append("I"); snp.append(sn); refc.append("-"); consumed = True
            else:
                rel.append("X"); snp.append(sn); refc.append(c); consumed = True
            ei += 1
        if not consumed:
            rel.append("="); snp.append(False)
            refc.append("ACGTN"[int(seq[rec.trim5 + i])] if seq is not None else "?")
    if seq is not None:
        readc = []
        k = 0
        for r in rel:
            if r == "D":
                readc.append("-")
            elif r == "N":
                readc.append("N")
            else:
                readc.append("ACGTN"[int(seq[rec.trim5 + k])]); k += 1
        ln = len(rel)
        i = 0
        while i < ln:                # StackedAln::leftAlign(false)
            r = rel[i]
            if r in "ID":
                if snp[i]:
                    i += 1
                    continue
                glen = 1
                for j in range(i + 1, ln):
                    if rel[j] != r:
                        break
                    glen += 1
                gp, ngp = (refc, readc) if r == "I" else (readc, refc)
                l = i - 1
                rr = l + glen
                while l > 0 and ngp[l] == ngp[rr]:
                    if rel[l] in "IDXN":
                        break
                    gp[l], gp[rr] = gp[rr], gp[l]
                    rel[l], rel[rr] = rel[rr], rel[l]
                    l -= 1; rr -= 1
                i += glen - 1
            i += 1
    ops = []

    def add(op, n=1):
        if n <= 0:
            return
        if ops and ops[-1][0] == op:
            ops[-1][1] += n
        else:
            ops.append([op, n])
    add("S", rec.trim5)
    nsk = 0
    for r in rel:
        if r == "N":
            ops.append(["N", skips[nsk]]); nsk += 1
        else:
            add("M" if r in "=X" else r)
    add("S", rec.trim3)
    return "".join(f"{n}{op}" for op, n in ops)


def render(outs, recs, names, rdlens, qnames):
    """-> {qname: [(flag, rname, pos, cigar, AS)]} in print order (primary first)"""
    res = {}
    for i, q in enumerate(qnames):
        o = outs[i]
        lst = []
        if o.nselect == 0:
            lst.append((4, "*", 0, "*", None))
        for k in range(o.nselect):
            r = recs[i * AL_MAX_RESULTS + o.select[k]]
            flag = (0 if r.fw else 16) | (256 if k > 0 else 0)
            lst.append((flag, names[r.tidx], r.toff + 1, cigar_of(r, rdlens[i]), int(r.score)))
        res[q] = lst
    return res


def render_selected(res, aln, names, rdlens, qnames, cap=api.ALN_CAP):
    """Same as render() for the C-ABI output (h2g_align_fetch): `aln` already holds the selected alignments in
    print order, `cap` slots per read."""
    out = {}
    for i, q in enumerate(qnames):
        lst = []
        nsel = int(res[i]["nselect"])
        if nsel == 0:
            lst.append((4, "*", 0, "*", None))
        for k in range(min(nsel, cap)):
            r = aln[i * cap + k]
            flag = (0 if r.fw else 16) | (256 if k > 0 else 0)
            lst.append((flag, names[r.tidx], r.toff + 1, cigar_of(r, rdlens[i]), int(r.score)))
        out[q] = lst
    return out
