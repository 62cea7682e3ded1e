"""Campaign behind tests/test_long_edits_cpu.py: random genomes, reads of 60-250 bases with substitutions, short indels AND a long deletion in a third of them,
--score-min L,0,-1 ... L,0,-3, single-end — the host instantiation of the large-workspace configuration (libh2gemu_long.so: 192 edits per working hit, records through the
long-edit area) against oracle/_ref/hisat2-align-s, every SAM line; counts the reads still flagged.   usage: fuzz_long_edits.py [cases] [seed0]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import sam_lines as SL
import sam_util as SU
from h2gemu_py import Emu
from h2gemu_align import set_options
from hisat2_amd import api, synth
from test_long_edits_cpu import deletion_reads

REF = os.path.join(ROOT, "oracle", "_ref")


def run_case(seed, verbose=True):
    rng = np.random.default_rng(seed)
    rdlen = int(rng.choice([60, 76, 101, 125, 150, 200, 250]))
    coeff = float(rng.choice([-1.0, -1.5, -2.0, -2.4, -3.0]))
    opts = ["--score-min", "L,0,%g" % coeff] + (["-k", str(int(rng.integers(1, 8)))] if rng.random() < 0.5 else [])
    n = 500
    tmp = tempfile.mkdtemp(prefix="h2flong")
    contigs = synth.make_genome([int(rng.integers(200000, 500000)), 120000], seed, n_gaps=1, gap_len=200, repeats=int(rng.integers(2, 30)), repeat_len=400)
    fa, base = os.path.join(tmp, "g.fa"), os.path.join(tmp, "g")
    synth.write_fasta(fa, contigs)
    subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads, _ = synth.make_reads(contigs, n, rdlen, seed + 1, sub_rate=float(rng.choice([0.005, 0.02, 0.05])), indel_rate=float(rng.choice([0.0, 0.002, 0.008])), n_rate=0.001)
    reads = np.asarray(reads).copy()
    dmax = max(27, min(90, int((-coeff * rdlen - 5) / 3)))
    d = deletion_reads(contigs, n, rdlen, seed + 2, dmin=26, dmax=dmax, sub=0.01)
    reads[::3] = d[::3]
    rfa, sam = os.path.join(tmp, "r.fa"), os.path.join(tmp, "ref.sam")
    synth.write_reads_fasta(rfa, reads)
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", "1", "--no-spliced-alignment", "-x", base, "-U", rfa, "-S", sam] + opts, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    want = SL.body_lines(sam)
    e = Emu(base, "long")
    set_options(e, 0, opts)
    codes, offs = SL.flat([reads[i] for i in range(n)])
    e.set_reads(codes, offs, None)
    names = [str(i) for i in range(n)]
    nb, noffs = SL.flat_names(names)
    outs = (SU.ReadOut * n)()
    rows = (api.AlnRes * (n * api.ALN_CAP))()
    cap = 1 << 19
    led = (api.Edit * cap)()
    used = C.c_uint32(0)
    vp = C.c_void_p
    e.L.h2gemu_align_abi.argtypes = [vp, C.c_uint32, C.c_char_p, vp, vp, vp, C.c_uint32, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    e.L.h2gemu_align_abi(e.h, 1, nb, noffs.ctypes.data, outs, rows, api.ALN_CAP, led, cap, C.byref(used))
    res = (api.ReadResult * n)()
    flagged = nlong = 0
    for i in range(n):
        o, r = outs[i], res[i]
        flagged += o.overflow != 0
        r.nres, r.nselect, r.overflow, r.nrank, r.nsteps, r.depth = o.nres, o.nselect, o.overflow, o.nrank, o.nsteps, o.depth
        r.best, r.secbest, r.best_h2, r.secbest_h2 = o.best, o.secbest, o.best_h2, o.secbest_h2
        nlong += sum(1 for k in range(min(o.nselect, api.ALN_CAP)) if rows[i * api.ALN_CAP + k].nedits > api.MAX_EDITS)
    got = SL.format_unpaired(SL.load_sam_lib(), base, [reads[i] for i in range(n)], names, res, rows, options=opts, long_edits=(led, used.value))
    bad = sum(1 for x, y in zip(got, want) if x != y) + abs(len(got) - len(want))
    if verbose:
        print("seed %d len %d %s: lines %d differing %d, flagged reads %d, records beyond 32 edits %d" % (seed, rdlen, " ".join(opts), len(want), bad, flagged, nlong), flush=True)
    return bad, flagged, nlong


if __name__ == "__main__":
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    tot = [0, 0, 0]
    for k in range(ncases):
        r = run_case(s0 + k)
        tot = [a + b for a, b in zip(tot, r)]
    print("total: differing lines %d, flagged reads %d, long records %d over %d cases" % (tot[0], tot[1], tot[2], ncases))
