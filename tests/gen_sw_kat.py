#!/usr/bin/env python3
"""Lifts the reference's own known-answer cases for the end-to-end SwAligner into tests/golden/sw_kat.json.

The cases live in a test `main` the reference no longer compiles (`#ifdef MAIN_ALIGNER_SW`, aligner_sw.cpp:1119-3300: `doTestCase` calls an
`initRead / initRef` signature that does not exist any more), so they cannot be *run*; what can be lifted is their content — inputs and asserted
outcomes of `doTests()` (aligner_sw.cpp:1470-2727) — written down here case by case with the line of each `doTestCase*` call.  The driver's setting,
restated by tests/sw_kat.py:
  * the DP window: `maxGaps = max(sc.maxReadGaps(minsc, len), sc.maxRefGaps(minsc, len))` (scoring.cpp:42-104), columns
    `[off - 2 maxGaps, off + len + maxGaps)` of the reference string, N beyond its ends (aligner_sw.cpp:1213-1245);
  * scoring: mismatch a constant 30 (`sc`) or the Phred quality itself (`sc2`: COST_MODEL_QUAL of that era; in this tree's terms `--mp 40,0`,
    scoring.h:117-124, which gives the same numbers for Q <= 40), N = 1, gap open = const + linear, extension = linear; `gapbar` rows at either
    end closed to gaps; `minsc` = the case's constant (its linear coefficient is 0 except in the last case);
  * the N ceiling applies only where the driver filters (`filterns`), as its old `al.filter(nceil)` did; elsewhere no limit;
  * asserted: whether the first nextAlignment finds an alignment, and its reference offset, reference extent, gaps, score and Ns.
Left out (and why): the three `EList<bool> en` cases of "N ceiling 2 with st_ override" (aligner_sw.cpp:2618-2655: an end-column mask the current
SwAligner no longer takes); the local-alignment cases (`doLocalTests`, :2729: HISAT2 never aligns in local mode); the assertions on
a *second* nextAlignment (the path here takes the first alignment only, spliced_aligner.h:262).

usage: gen_sw_kat.py   (no reference needed at run time: the table below IS the lifted content; /root/reference only to re-check the line numbers)"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF16 = "ACGTACGTACGTACGT"


def sc(mm="30", rd=(40, 15), rf=(40, 15), gapbar=1, npen=1):
    """mm: "30" constant / "qual"; rd / rf = (const, linear)"""
    return {"mm": mm, "npen": npen, "rdGapConst": rd[0], "rdGapLinear": rd[1], "rfGapConst": rf[0], "rfGapLinear": rf[1], "gapbar": gapbar}


def exp(found, score=None, gaps=None, ns=None, extent=None, refoff="off"):
    return {"found": found, "score": score, "gaps": gaps, "ns": ns, "extent": extent, "refoff": refoff}


# (line of the doTestCase call in aligner_sw.cpp, title, read, qual, ref, offsets, scoring, minsc, nceil or None (= not filtered), expected)
LOOP = [0, 4, 8]            # `for(int i = 0; i < 3; i++) … off = i*4` (aligner_sw.cpp:1547)
CASES = [
    (1566, "exact", "ACGTACGT", "IIIIIIII", REF16, LOOP, sc(), -30, None, exp(True, 0, 0, 0, 8)),
    (1599, "1mm allowed by minsc", "ACGTTCGT", "IIIIIIII", REF16, LOOP, sc(), -30, None, exp(True, -30, 0, 0, 8)),
    (1627, "1mm allowed by minsc, check qual 1", "ACGTTCGT", "ABCDEFGH", REF16, LOOP, sc("qual", (25, 15), (25, 15)), -40, None, exp(True, -36, 0, 0, 8)),
    (1665, "1mm allowed by minsc, check qual 2", "ACGAACGT", "ABCDEFGH", REF16, LOOP, sc("qual", (25, 15), (25, 15)), -40, None, exp(True, -35, 0, 0, 8)),
    (1695, "1mm allowed by minsc, check qual", "TCGTACGT", "ABCDEFGH", REF16, LOOP, sc("qual", (25, 15), (25, 15)), -40, None, exp(True, -32, 0, 0, 8)),
    (1724, "1mm at the beginning, allowed by minsc", "CCGTACGT", "IIIIIIII", REF16, LOOP, sc(), -30, None, exp(True, -30, 0, 0, 8)),
    (1754, "1 n in read, allowed", "ACGTNCGT", "IIIIIIII", REF16, LOOP, sc(), -30, None, exp(True, -1, 0, 1, 8)),
    (1784, "2 n in read, allowed", "ACGNNCGT", "IIIIIIII", REF16, LOOP, sc(), -30, None, exp(True, -2, 0, 2, 8)),
    (1814, "2 n in read, 1 at beginning, allowed", "NCGTNCGT", "IIIIIIII", REF16, LOOP, sc(), -30, None, exp(True, -2, 0, 2, 8)),
    (1842, "1 n in ref, allowed", "ACGTACGT", "IIIIIIII", "ACGTNCGTACGTANGT", LOOP, sc(), -30, None, exp(True, -1, 0, 1, 8)),
    (1870, "1mm disallowed by minsc", "ACGTTCGT", "IIIIIIII", REF16, LOOP, sc(), -10, None, exp(False)),
    (1895, "read gap allowed by minsc", "ACGTCGT", "IIIIIII", REF16, LOOP, sc(rd=(25, 15), rf=(25, 15)), -40, None, exp(True, -40, 1, 0, 8)),
    (1927, "read gap disallowed by minsc", "ACGTCGT", "IIIIIII", REF16, LOOP, sc(rd=(25, 15), rf=(25, 15)), -30, None, exp(False)),
    (1948, "ref gap allowed by minsc", "ACGTAACGT", "IIIIIIIII", REF16, LOOP, sc(rd=(25, 15), rf=(25, 15)), -40, None, exp(True, -40, 1, 0, 8)),
    (1981, "read gap disallowed by gap barrier", "ACGTCGT", "IIIIIII", REF16, LOOP, sc(rd=(25, 15), rf=(25, 15), gapbar=4), -40, None, exp(False)),
    (2004, "ref gap allowed by minsc, gapbar=3", "ACGTAACGT", "IIIIIIIII", REF16, LOOP, sc(rd=(25, 15), rf=(25, 15), gapbar=3), -40, None, exp(True, -40, 1, 0, 8)),
    (2035, "ref gap allowed by minsc, gapbar=4", "ACGTAACGT", "IIIIIIIII", REF16, LOOP, sc(rd=(25, 15), rf=(25, 15), gapbar=4), -40, None, exp(True, -40, 1, 0, 8)),
    (2064, "ref gap disallowed by minsc", "ACGTAACGT", "IIIIIIIII", REF16, LOOP, sc(rd=(25, 15), rf=(25, 15)), -30, None, exp(False)),
    (2085, "ref gap disallowed by gap barrier", "ACGTAACGT", "IIIIIIIII", REF16, LOOP, sc(rd=(25, 15), rf=(25, 15), gapbar=5), -40, None, exp(False)),
    (2111, "1 read gap, ref gaps disallowed by minsc", "ACGTCGT", "IIIIIII", REF16, LOOP, sc(rd=(25, 10), rf=(35, 20)), -40, None, exp(True, -35, 1, 0, 8)),
    (2143, "gaps disallowed by minsc (read gap 35)", "ACGTCGT", "IIIIIII", REF16, LOOP, sc(rd=(25, 10), rf=(25, 10)), -30, None, exp(False)),
    (2169, "1 ref gap, read gaps disallowed by minsc", "ACGTAACGT", "IIIIIIIII", REF16, LOOP, sc(rd=(35, 22), rf=(25, 12)), -40, None, exp(True, -37, 1, 0, 8)),
    (2197, "gaps disallowed by minsc (ref gap 37)", "ACGTAACGT", "IIIIIIIII", REF16, LOOP, sc(rd=(35, 22), rf=(25, 12)), -30, None, exp(False)),
    (2221, "1 read gap, 2 ref gaps allowed by minsc", "ACGTCGT", "IIIIIII", REF16, LOOP, sc(rd=(25, 15), rf=(20, 10)), -40, None, exp(True, -40, 1, 0, 8)),
    (2249, "gaps disallowed by minsc (read gap 40)", "ACGTCGT", "IIIIIII", REF16, LOOP, sc(rd=(25, 15), rf=(20, 10)), -30, None, exp(False)),
    (2273, "1 ref gap, 2 read gaps allowed by minsc", "ACGTAACGT", "IIIIIIIII", REF16, LOOP, sc(rd=(11, 10), rf=(25, 15)), -40, None, exp(True, -40, 1, 0, 8)),
    (2300, "gaps disallowed by minsc (ref gap 40)", "ACGTAACGT", "IIIIIIIII", REF16, LOOP, sc(rd=(11, 10), rf=(25, 15)), -30, None, exp(False)),
    (2325, "2 ref gaps, 2 read gaps allowed by minsc", "ACGTCGT", "IIIIIII", REF16, LOOP, sc(rd=(15, 10), rf=(15, 10)), -40, 1, exp(True, -25, 1, 0, 8)),
    (2371, "1 ref gap, 1 read gap allowed by minsc", "ACGTCGT", "IIIIIII", REF16, LOOP, sc(rd=(10, 10), rf=(10, 10)), -30, None, exp(True, -20, 1, 0, 8)),
    (2402, "2 ref gaps, 2 read gaps allowed by minsc (ref gap)", "ACGTAACGT", "IIIIIIIII", REF16, LOOP, sc(rd=(15, 5), rf=(15, 5)), -35, 1, exp(True, -20, 1, 0, 8, refoff="any of 0 4 8")),
    (2447, "1 ref gap, 1 read gap allowed by minsc (ref gap 29)", "ACGTAACGT", "IIIIIIIII", REF16, LOOP, sc(rd=(25, 4), rf=(25, 4)), -30, None, exp(True, -29, 1, 0, 8)),
    (2471, "short read", "A", "I", "AAAAAAAAAAAA", LOOP, sc(rd=(25, 4), rf=(25, 4)), -30, None, exp(True, 0, 0, 0, None, refoff=None)),
    (2494, "short read & ref", "A", "I", "A", [0], sc(rd=(25, 4), rf=(25, 4)), -30, None, exp(True, 0, 0, 0, None, refoff=None)),
    (2517, "short read, many allowed gaps", "A", "I", "AAAAAAAAAAAA", LOOP, sc(rd=(25, 4), rf=(25, 4)), -150, None, exp(True, 0, 0, 0, None, refoff=None)),
    (2541, "short read & ref, many allowed gaps", "A", "I", "A", [0], sc(rd=(25, 4), rf=(25, 4)), -150, None, exp(True, 0, 0, 0, None, refoff=None)),
    # after the loop: mismatch 10, N 2, ref gaps 10/10, read gaps 25/10 (rdGapConst keeps the loop's last value, :2572-2580)
    (2579, "N ceiling 1", "ACGTACGT", "IIIIIIII", "NCGTACGT", [0], sc("10", (25, 10), (10, 10), npen=2), -25, 0, exp(False)),
    (2599, "N ceiling 2 (filter off)", "ACGTACGT", "IIIIIIII", "NCGTACGT", [0], sc("10", (25, 10), (10, 10), npen=2), -25, None, exp(True, -2, 0, 1, None, refoff=None)),
    (2664, "N ceiling 3 (filter on, one N allowed)", "ACGTACGT", "IIIIIIII", "NCGTACGT", [0], sc("10", (25, 10), (10, 10), npen=2), -25, 1, exp(True, -2, 0, 1, None, refoff=None)),
    # minsc = -25 - 5 x 47 = -260: below -254, i.e. the 16-bit cells of the current SwAligner
    (2693, "redundant alignment elimination 1", "AGGCTATGCCTCTGACGCGATATCGGCGCCCACTTCAGAGCTAACCG", "I" * 47,
     "TTTTTTTTAGGCTATGCCTCTGACGCGATATCGGCGCCCACTTCAGAGCTAACCGTTTTTTT", [8], sc("10", (25, 15), (25, 15), npen=2), -260, 1, exp(True, 0, 0, 0, 47)),
]


def main():
    out = []
    for line, title, read, qual, ref, offs, scoring, minsc, nceil, e in CASES:
        assert len(read) == len(qual)
        for off in offs:
            out.append({"line": line, "title": title, "read": read, "qual": qual, "ref": ref, "off": off, "scoring": scoring, "minsc": minsc,
                        "nceil": nceil, "expect": e})
    path = os.path.join(HERE, "golden", "sw_kat.json")
    json.dump({"source": "aligner_sw.cpp:1470-2727 doTests() of the reference (MAIN_ALIGNER_SW), lifted by tests/gen_sw_kat.py", "cases": out}, open(path, "w"), indent=0)
    print(len(out), "cases ->", path)


if __name__ == "__main__":
    main()
