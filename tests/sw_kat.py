"""Runner of tests/golden/sw_kat.json — the reference's own SwAligner known-answer cases (aligner_sw.cpp:1470-2727, lifted by gen_sw_kat.py) —
over any implementation of `align(case, window) -> dict(found, score, off, gaps, ns, edits=[(type, pos)...])`."""
import json
import os

import numpy as np

CODE = {"A": 0, "C": 1, "G": 2, "T": 3, "N": 4}


def load(golden_dir):
    return json.load(open(os.path.join(golden_dir, "sw_kat.json")))["cases"]


def max_gaps(minsc, open_pen, ext_pen):
    """Scoring::maxReadGaps / maxRefGaps (scoring.cpp:42-104) with no match bonus: matches are traded for gaps until the score falls below minsc"""
    sc, first, num = 0, True, 0
    while sc >= minsc:
        sc -= open_pen if first else ext_pen
        first = False
        num += 1
    return num - 1


def window(case):
    """(refl, refr, corel, corer) of the test driver's DP rectangle (aligner_sw.cpp:1213-1250, :1268-1276): 2 maxGaps columns before `off`, maxGaps after
    the read's end; a solution may end on the `width - maxGaps` rightmost diagonals, i.e. within maxGaps of `off` — what DynProgFramer calls the
    core diagonals (dp_framer.cpp:81-130 builds HISAT2's rectangle the same way: refl = refoff - 2 maxgap, core = maxgap .. 3 maxgap)"""
    s = case["scoring"]
    g = max(max_gaps(case["minsc"], s["rdGapConst"] + s["rdGapLinear"], s["rdGapLinear"]),
            max_gaps(case["minsc"], s["rfGapConst"] + s["rfGapLinear"], s["rfGapLinear"]))
    n = len(case["read"])
    return case["off"] - 2 * g, case["off"] + n + g - 1, g, 3 * g


def codes(s):
    return np.array([CODE[c] for c in s], dtype=np.uint8)


def mm_range(scoring):
    """(mmpMax, mmpMin) of this tree's Scoring giving the case's mismatch model: a constant, or the quality itself (`--mp 40,0`)"""
    return (40, 0) if scoring["mm"] == "qual" else (int(scoring["mm"]), int(scoring["mm"]))


def check(case, r):
    e = case["expect"]
    assert bool(r["found"]) == e["found"], (case["line"], case["title"], case["off"], r)
    if not e["found"]:
        return
    assert r["score"] == e["score"] and r["gaps"] == e["gaps"] and r["ns"] == e["ns"], (case["line"], case["title"], case["off"], r)
    if e["refoff"] == "off":
        assert r["off"] == case["off"], (case["line"], case["title"], case["off"], r)
    elif e["refoff"] is not None:
        assert r["off"] in (0, 4, 8), (case["line"], r)
    if e["extent"] is not None:
        # AlnRes::refExtent: read length - read characters opposite a reference gap + reference characters opposite a read gap
        ext = len(case["read"]) - sum(1 for t, _ in r["edits"] if t == 2) + sum(1 for t, _ in r["edits"] if t == 1)
        assert ext == e["extent"], (case["line"], case["title"], case["off"], r)
