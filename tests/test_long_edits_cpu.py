"""Alignments with more edits than a record's 32 inline entries (the reference's edit lists are unbounded, hi_aligner.h:421, and a
deletion of n bases is n edits, edit.h): the units with the large workspace hold 192 edits per working hit (H2G_GHIT_EDITS, h2g_go_big.h) and such a
record leaves through the long-edit area (MachOut::ledits -> h2g_align_fetch_long_edits -> h2g_sam_set_long_edits).  Host instantiation of exactly that
configuration (tests/emul/libh2gemu_long.so) against the reference binary: reads carrying a 26-70-base deletion at --score-min L,0,-2.4 — every SAM line."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

import sam_lines as SL
import sam_util as SU
from h2gemu_py import Emu
from h2gemu_align import set_options
from hisat2_amd import api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def deletion_reads(contigs, n, rdlen, seed, dmin=26, dmax=70, sub=0.01):
    """reads = a genome window with its middle D bases left out (D in [dmin, dmax]), both flanks >= 30 bases, either strand, a few substitutions"""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, rdlen), dtype=np.uint8)
    k = 0
    while k < n:
        c = contigs[int(rng.integers(0, len(contigs)))]
        D = int(rng.integers(dmin, dmax + 1))
        left = int(rng.integers(min(30, rdlen // 3), max(min(30, rdlen // 3) + 1, rdlen - 30)))
        s = int(rng.integers(0, len(c) - rdlen - D - 1))
        w = np.concatenate([c[s:s + left], c[s + left + D:s + rdlen + D]])
        if (w > 3).any():
            continue
        m = rng.random(rdlen) < sub
        w = np.where(m, (w + rng.integers(1, 4, size=rdlen)) & 3, w).astype(np.uint8)
        if rng.random() < 0.5:
            w = (3 - w[::-1]).astype(np.uint8)
        out[k] = w
        k += 1
    return out


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "hisat2-align-s")), reason="needs oracle/_ref")
@pytest.mark.parametrize("case", [dict(seed=2801, n=600, rdlen=101, opts=("--score-min", "L,0,-2.4")),
                                  dict(seed=2802, n=400, rdlen=150, opts=("--score-min", "L,0,-1.5", "-k", "3"))])
def test_records_beyond_32_edits_equal_the_reference(case):
    tmp = tempfile.mkdtemp(prefix="h2long")
    contigs = synth.make_genome([400000, 150000], case["seed"], n_gaps=1, gap_len=200, repeats=4, repeat_len=400)
    fa, base = os.path.join(tmp, "g.fa"), os.path.join(tmp, "g")
    synth.write_fasta(fa, contigs)
    subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    n, rdlen, opts = case["n"], case["rdlen"], list(case["opts"])
    reads = deletion_reads(contigs, n, rdlen, case["seed"] + 1)
    rfa, sam = os.path.join(tmp, "r.fa"), os.path.join(tmp, "ref.sam")
    synth.write_reads_fasta(rfa, reads)
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", "1", "--no-spliced-alignment", "-x", base, "-U", rfa, "-S", sam] + opts, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    want = SL.body_lines(sam)
    e = Emu(base, "long")
    e.L.h2gemu_ghit_edits.restype = C.c_uint32
    assert e.L.h2gemu_ghit_edits() == 192                          # the *_big units' own configuration
    p = set_options(e, 0, opts)
    codes, offs = SL.flat([reads[i] for i in range(n)])
    e.set_reads(codes, offs, None)
    names = [str(i) for i in range(n)]
    nb, noffs = SL.flat_names(names)
    outs = (SU.ReadOut * n)()
    rows = (api.AlnRes * (n * api.ALN_CAP))()
    cap = 1 << 18
    led = (api.Edit * cap)()
    used = C.c_uint32(0)
    vp = C.c_void_p
    e.L.h2gemu_align_abi.argtypes = [vp, C.c_uint32, C.c_char_p, vp, vp, vp, C.c_uint32, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    e.L.h2gemu_align_abi(e.h, 1, nb, noffs.ctypes.data, outs, rows, api.ALN_CAP, led, cap, C.byref(used))
    res = (api.ReadResult * n)()
    nlong = 0
    for i in range(n):
        o, r = outs[i], res[i]
        assert o.overflow == 0, (i, o.overflow)
        r.nres, r.nselect, r.overflow, r.nrank, r.nsteps, r.depth = o.nres, o.nselect, o.overflow, o.nrank, o.nsteps, o.depth
        r.best, r.secbest, r.best_h2, r.secbest_h2 = o.best, o.secbest, o.best_h2, o.secbest_h2
        for k in range(min(o.nselect, api.ALN_CAP)):
            rec = rows[i * api.ALN_CAP + k]
            if rec.nedits > api.MAX_EDITS:
                nlong += 1
                assert rec.edits[0].snp == 0x4c4f4e47 and rec.edits[0].pos + rec.nedits <= used.value
    assert nlong >= 20, nlong                                      # the long path is what this test is about (most such reads end soft-clipped)
    L = SL.load_sam_lib()
    got = SL.format_unpaired(L, base, [reads[i] for i in range(n)], names, res, rows, options=opts, long_edits=(led, used.value))
    assert len(got) == len(want)
    bad = [i for i, (x, y) in enumerate(zip(got, want)) if x != y]
    assert not bad, (len(bad), got[bad[0]], want[bad[0]])
    # a long record without its area is refused by the dense formatter, never mis-printed
    nsel = np.array([min(outs[i].nselect, api.ALN_CAP) for i in range(n)])
    dense = (api.AlnRes * int(nsel.sum()))()
    offs64 = np.concatenate([[0], np.cumsum(nsel)]).astype(np.uint64)
    for i in range(n):
        for k in range(int(nsel[i])):
            C.memmove(C.byref(dense[int(offs64[i]) + k]), C.byref(rows[i * api.ALN_CAP + k]), C.sizeof(api.AlnRes))
    h = vp()
    L.h2g_sam_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    assert L.h2g_sam_open(base.encode(), C.byref(h)) == 0
    SL._score_min(L, h, opts)
    L.h2g_sam_format_unpaired_dense.argtypes = [vp] + [vp] * 5 + [C.c_size_t, vp, vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    buf = C.create_string_buffer(1 << 22)
    u = C.c_size_t(0)
    rc = L.h2g_sam_format_unpaired_dense(h, codes.ctypes.data, offs.ctypes.data, None, nb, noffs.ctypes.data, n, C.addressof(res), C.addressof(dense), offs64.ctypes.data, buf, 1 << 22, C.byref(u))
    assert rc != 0
    L.h2g_sam_set_long_edits.argtypes = [vp, vp, C.c_size_t]
    L.h2g_sam_set_long_edits(h, led, used.value)
    rc = L.h2g_sam_format_unpaired_dense(h, codes.ctypes.data, offs.ctypes.data, None, nb, noffs.ctypes.data, n, C.addressof(res), C.addressof(dense), offs64.ctypes.data, buf, 1 << 22, C.byref(u))
    assert rc == 0 and buf.raw[:u.value].decode().splitlines() == want
    L.h2g_sam_close(h)
