"""Records beyond 32 edits on the device: reads carrying a 26-70-base deletion at --score-min L,0,-2.4 (a deletion of n bases is n edits,
edit.h; the reference's lists are unbounded, hi_aligner.h:421) through hisat2-align-amd — default units flag them, the large-workspace units (192 edits per
working hit) align them, the records leave through the long-edit area (h2g_align_fetch_long_edits) — against the reference binary, every SAM line + summary."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

import sam_lines as SL
from hisat2_amd import synth
from test_long_edits_cpu import deletion_reads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
CLI = os.path.join(ROOT, "hisat2_amd", "hisat2-align-amd")


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "hisat2-align-s")), reason="needs oracle/_ref")
@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("opts", [("--score-min", "L,0,-2.4"), ("--score-min", "L,0,-3", "-k", "3")])
def test_long_deletions_through_the_command_line(paired, opts):
    tmp = tempfile.mkdtemp(prefix="h2glong")
    contigs = synth.make_genome([600000, 200000], 2901, n_gaps=1, gap_len=200, repeats=4, repeat_len=400)
    fa, base = os.path.join(tmp, "g.fa"), os.path.join(tmp, "g")
    synth.write_fasta(fa, contigs)
    subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    n = 3000
    if paired:   # ordinary pairs, every third mate 1 replaced by a read with a long deletion
        m1, m2 = synth.make_pairs(contigs, n, 101, 2903, frag_mean=300, frag_sd=30, sub_rate=0.01)
        d = deletion_reads(contigs, n, 101, 2904)
        m1 = np.stack(m1).copy()
        m1[::3] = d[::3]
        synth.write_reads_fasta(os.path.join(tmp, "1.fa"), m1); synth.write_reads_fasta(os.path.join(tmp, "2.fa"), np.stack(m2))
        rd = ["-1", os.path.join(tmp, "1.fa"), "-2", os.path.join(tmp, "2.fa")]
    else:
        synth.write_reads_fasta(os.path.join(tmp, "r.fa"), deletion_reads(contigs, n, 101, 2902))
        rd = ["-U", os.path.join(tmp, "r.fa")]
    common = ["-f", "--no-spliced-alignment", "-x", base] + rd + list(opts)
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-p", "1"] + common + ["-S", os.path.join(tmp, "ref.sam")], check=True, stderr=open(os.path.join(tmp, "ref.err"), "w"))
    stats = os.path.join(tmp, "stats.json")
    r = subprocess.run([CLI, "-p", "4", "--batch", "1000", "--h2g-stats", stats] + common + ["-S", os.path.join(tmp, "amd.sam")], stderr=open(os.path.join(tmp, "amd.err"), "w"))
    assert r.returncode == 0, open(os.path.join(tmp, "amd.err")).read()[-1500:]
    st = json.load(open(stats))
    assert st["overflow"] == 0 and st["second_pass"] > 0, st
    want = SL.body_lines(os.path.join(tmp, "ref.sam"))
    got = SL.body_lines(os.path.join(tmp, "amd.sam"))
    long_lines = sum(1 for l in want if any(int(x[:-1]) >= 26 for x in __import__("re").findall(r"\d+D", l.split("\t")[5])))
    assert long_lines >= 50, long_lines                        # the reference did place deletions of 26 bases and more
    assert len(got) == len(want) and not [i for i, (x, y) in enumerate(zip(got, want)) if x != y][:3]
    assert open(os.path.join(tmp, "amd.err")).read() == open(os.path.join(tmp, "ref.err")).read()
