"""Stress of the SHIPPED build: the hard-read case of test_gpu_fast_pass.py (half the reads are handed on to the general machine), 8 runs
queued back to back per configuration — machine passes of two earlier runs in flight on the machine streams next to each fast pass —
over four batch sizes, with the fast pass on and off, pairs and single reads: 128 queued runs, every fetched result compared read by
read with one machine-only run (tests/fast_stress.py).  profiles/r04_NOTES.md has the experiment this test came out of."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from hisat2_amd import synth
import fast_stress as FS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "oracle", "_ref", "hisat2-build-s")


def hard_case(tmp, seed=72, npairs=60000, nreads=60000, rdlen=76):
    contigs = synth.make_genome([1500000, 400000, 100000], seed, n_gaps=3, gap_len=300, repeats=80, repeat_len=600)
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs)
    base = os.path.join(tmp, "g")
    subprocess.run([BUILD, "-q", "-p", "16", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    m1, m2 = synth.make_pairs(contigs, npairs, rdlen, seed + 1, frag_mean=300, frag_sd=40, sub_rate=0.03)
    reads, _ = synth.make_reads(contigs, nreads, rdlen, seed + 2, sub_rate=0.03, indel_rate=0.002, n_rate=0.002)
    npz = os.path.join(tmp, "reads.npz")
    np.savez(npz, m1=np.stack(m1), m2=np.stack(m2), reads=np.asarray(reads))
    return base, npz


@pytest.mark.skipif(not os.path.exists(BUILD), reason="needs oracle/_ref/hisat2-build-s")
def test_queued_runs_equal_the_machine_read_by_read():
    tmp = tempfile.mkdtemp(prefix="h2fs")
    base, npz = hard_case(tmp)
    r = FS.run(base, npz, runs=8, sizes=("all", 0.63, 0.2, 999), log=print)
    assert len(r["cases"]) == 16
    assert r["differing"] == 0, [c for c in r["cases"] if c["differing"]]
    on = [c for c in r["cases"] if c["fast"] == 1 and c["n"] > 1000]
    assert all(c["handed_on"] > 0.2 * c["n"] for c in on), on        # hard reads: the machine passes carry real work
