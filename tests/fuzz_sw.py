"""SwAligner against the LIVE reference classes: fresh reads of the golden genome g1 through oracle/_ref/ref_probe `sw` (the hybridSearch call site of
SwAligner run by the reference's own objects, oracle/ref_probe.cpp) at a given --score-min and placement shift, compared problem by problem with a
backend's sw_align (the host instantiation of h2g_sw.h, or the device).  usage: fuzz_sw.py seed nreads rdlen sub minsc shift"""
import gzip
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import parity_cases as PC  # noqa: E402
from hisat2_amd import api, synth  # noqa: E402

PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_probe")


def run_case(make_backend, golden_dir, seed, nreads, rdlen, sub, minsc, shift=0, indel=0.01, nrate=0.002):
    """-> (problems, differing).  make_backend(base, reads[n, rdlen]) -> object with sw_align(queries)"""
    tmp = tempfile.mkdtemp(prefix="h2fuzzsw")
    base = os.path.join(tmp, "g1")
    for k in range(1, 9):
        open(f"{base}.{k}.ht2", "wb").write(gzip.open(os.path.join(golden_dir, f"g1.{k}.ht2.gz")).read())
    contigs = PC.load_contigs(golden_dir)
    reads, _ = synth.make_reads(contigs, nreads, rdlen, seed, sub_rate=sub, indel_rate=indel, n_rate=nrate)
    rfa = os.path.join(tmp, "r.fa")
    synth.write_reads_fasta(rfa, reads)
    out = subprocess.run([PROBE, "sw", base, rfa, "1", str(minsc), str(shift)], check=True, stdout=subprocess.PIPE).stdout
    open(os.path.join(tmp, "probe.txt.gz"), "wb").write(gzip.compress(out))
    cases = PC.parse_sw_probe(tmp, "probe.txt.gz")
    be = make_backend(base, reads)
    qs = [api.SwQuery(d["rid"], d["fw"], d["tidx"], d["refoff"], d["minsc"], (d["rid"] * 7 + d["k"] + 1) & 0xFFFFFFFF) for d in cases]
    res, _ = be.sw_align(qs)
    bad = 0
    for d, o in zip(cases, res):
        big = bool(d["found"]) and len(d["edits"]) > api.MAX_EDITS
        ok = bool(o.overflow) == big and [o.refl, o.refr] == d["rect"][:2] and (o.found_align, o.best, o.found) == (d["found_align"], d["best"], d["found"])
        ok = ok and PC.H.lcg_next(o.rnd)[0] == d["rnd_next"]
        if ok and d["found"]:
            ok = (o.score, o.off) == (d["score"], d["off"]) and (big or PC.sw_edit_strings(o.edits, o.nedits, d["fw"], rdlen) == d["edits"])
        if not ok:
            bad += 1
            if bad <= 3:
                print("DIFF", d, (o.found_align, o.best, o.found, o.score, o.off, o.nedits, o.overflow))
    return len(cases), bad


def emu_backend(base, reads):
    from h2gemu_py import Emu
    e = Emu(base)
    offs = (np.arange(len(reads) + 1, dtype=np.uint64) * reads.shape[1]).astype(np.uint32)
    e.set_reads(reads.reshape(-1), offs)
    return e


if __name__ == "__main__":
    a = sys.argv[1:]
    n, bad = run_case(emu_backend, os.path.join(HERE, "golden"), int(a[0]), int(a[1]), int(a[2]), float(a[3]), int(a[4]), int(a[5]) if len(a) > 5 else 0)
    print("problems", n, "differing", bad)
    sys.exit(1 if bad else 0)
