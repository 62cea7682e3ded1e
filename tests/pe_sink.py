"""Host-side half of AlnSinkWrap::finishRead for read pairs (aln_sink.h:1939-2560), used by the
tests to turn the report events returned by the paired go() into SAM lines: ReportingState bookkeeping
(aln_sink.cpp:30-345), selectByScore (aln_sink.h:2680) continuing the per-pair PRNG, FLAG assembly (appendMate
aln_sink.h:3055-3085).  -k mode, mixed + discordant reporting enabled (hisat2 defaults)."""
import ctypes as C

import sam_util as SU

AL_MAX_PAIRS = 32


class PairOut(C.Structure):
    _fields_ = [("nres", C.c_uint32 * 2), ("npairs", C.c_uint32), ("overflow", C.c_uint32), ("nrank", C.c_uint32),
                ("nsteps", C.c_uint32), ("depth", C.c_uint32), ("nside", C.c_uint32), ("rnd_state", C.c_uint32),
                ("pad", C.c_uint32), ("pair_i", C.c_uint8 * AL_MAX_PAIRS), ("pair_j", C.c_uint8 * AL_MAX_PAIRS)]


class Rng:   # RandomSource random_source.h:33-60
    def __init__(self, last):
        self.last = last & 0xFFFFFFFF

    def next_u32(self):
        self.last = (1664525 * self.last + 1013904223) & 0xFFFFFFFF
        ret = self.last >> 16
        self.last = (1664525 * self.last + 1013904223) & 0xFFFFFFFF
        return ret ^ self.last


def hisat2_score(r):   # AlnScore::calculate_hisat2_score aligner_result.h:322 (no repeat / transcript / splice terms)
    score = max(min(int(r.score), 2 ** 31 - 1), -2 ** 31)
    trim = r.trim5 + r.trim3
    trim = 0 if trim > 0xFFFF else 0xFFFF - trim
    return (score << 32) | (255 << 16) | trim


def select_by_score(keys, num, rnd, secondary=False):
    """selectByScore aln_sink.h:2680-2760: returns indexes to print, best first."""
    sz = len(keys)
    if sz < 1:
        return []
    num = min(num, sz)
    buf = sorted(((k, i) for i, k in enumerate(keys)), reverse=True)
    streak = 0
    i = 1
    n = len(buf)

    def shuffle(begin, cnt):
        left = cnt
        for q in range(begin, begin + cnt - 1):
            r = rnd.next_u32() % left
            if r > 0:
                buf[q], buf[q + r] = buf[q + r], buf[q]
            left -= 1
    while i < n:
        if buf[i][0] == buf[i - 1][0]:
            if streak == 0:
                streak = 1
            streak += 1
        else:
            if streak > 1:
                shuffle(i - streak, streak)
            streak = 0
        i += 1
    if streak > 1:
        shuffle(n - streak, streak)
    sel = [buf[i][1] for i in range(n) if i < num]
    if not secondary:
        for i in range(len(sel) - 1):
            if buf[i][0] != buf[i + 1][0]:
                sel = sel[:i + 1]
                break
    return sel


def finish_pair(out, recs1, recs2, base, names, rdlens, khits=5, secondary=False):
    """-> list of (flag, rname, pos, cigar, AS) in the order the reference prints the lines of this pair."""
    r1 = [recs1[base + k] for k in range(out.nres[0])]
    r2 = [recs2[base + k] for k in range(out.nres[1])]
    pairs = [(out.pair_i[k], out.pair_j[k]) for k in range(out.npairs)]
    rnd = Rng(out.rnd_state)
    # ReportingState::foundConcordant aln_sink.cpp:74-112
    best, nconc = None, 0
    for i, j in pairs:
        sc = int(r1[i].score) + int(r2[j].score)
        if best is None or sc > best:
            best, nconc = sc, 0
        nconc += 1
    lines = []

    def line(rec, mate, rdlen, orec, concordant, primary, mate_aligned):
        fl = 1 | (0x40 if mate == 0 else 0x80)
        if concordant:
            fl |= 2
        if not mate_aligned:
            fl |= 8
        if mate_aligned and orec is not None and not orec.fw:
            fl |= 0x20
        if not primary:
            fl |= 0x100
        if not rec.fw:
            fl |= 0x10
        return (fl, names[rec.tidx], rec.toff + 1, SU.cigar_of(rec, rdlen), int(rec.score))
    if nconc > 0:
        keys = [hisat2_score(r1[i]) + hisat2_score(r2[j]) for i, j in pairs]
        sel = select_by_score(keys, min(khits, nconc), rnd, secondary)
        for n, k in enumerate(sel):
            i, j = pairs[k]
            lines.append(line(r1[i], 0, rdlens[0], r2[j], True, n == 0, True))
            lines.append(line(r2[j], 1, rdlens[1], r1[i], True, n == 0, True))
        return lines
    n1, n2 = len(r1), len(r2)
    if n1 == 1 and n2 == 1:   # finish(): convertUnpairedToDiscordant; prepareDiscordants aln_sink.h:2660
        select_by_score([hisat2_score(r1[0]) + hisat2_score(r2[0])], 1, rnd, secondary)
        lines.append(line(r1[0], 0, rdlens[0], r2[0], False, True, True))
        lines.append(line(r2[0], 1, rdlens[1], r1[0], False, True, True))
        return lines
    sel1 = select_by_score([hisat2_score(x) for x in r1], min(khits, n1), rnd, secondary) if n1 else []
    sel2 = select_by_score([hisat2_score(x) for x in r2], min(khits, n2), rnd, secondary) if n2 else []
    p1 = r1[sel1[0]] if sel1 else None
    p2 = r2[sel2[0]] if sel2 else None
    if p1 is not None and p2 is not None:
        lines.append(line(p1, 0, rdlens[0], p2, False, True, True))
        lines.append(line(p2, 1, rdlens[1], p1, False, True, True))
        for k in sel1[1:]:
            lines.append(line(r1[k], 0, rdlens[0], p2, False, False, True))
        for k in sel2[1:]:
            lines.append(line(r2[k], 1, rdlens[1], p1, False, False, True))
    elif p1 is not None:
        for n, k in enumerate(sel1):
            lines.append(line(r1[k], 0, rdlens[0], None, False, n == 0, False))
        lines.append((1 | 4 | 0x80, names[p1.tidx], p1.toff + 1, "*", None))
    elif p2 is not None:
        for n, k in enumerate(sel2):
            lines.append(line(r2[k], 1, rdlens[1], None, False, n == 0, False))
        lines.append((1 | 4 | 0x40, names[p2.tidx], p2.toff + 1, "*", None))
    else:
        lines.append((1 | 4 | 8 | 0x40, "*", 0, "*", None))
        lines.append((1 | 4 | 8 | 0x80, "*", 0, "*", None))
    return lines
