"""The fast pass of go() on the device (h2g_k_go_fast.hip) against the general machine on the device: the same batch with H2G_GO_FAST=1
and =0 (separate processes: the switch is read once) must give byte-identical results — every PairOut / ReadOut incl. the work counters
and the PRNG state, every record — for pairs and for single reads, with runs queued back to back (machine passes in flight next to the
following fast passes).  The machine is pinned to the reference binary by the other GPU suites."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from hisat2_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "oracle", "_ref", "hisat2-build-s")


@pytest.mark.skipif(not os.path.exists(BUILD), reason="needs oracle/_ref/hisat2-build-s")
@pytest.mark.parametrize("case", [
    dict(seed=71, npairs=150000, nreads=150000, rdlen=101, sub=0.005),
    dict(seed=72, npairs=60000, nreads=60000, rdlen=76, sub=0.03, indel=0.002, nrate=0.002, least=0.25),   # hard reads: half are handed on
    dict(seed=73, npairs=100000, nreads=100000, rdlen=101, sub=0.005, snps=250),                            # SNP-graph index (h2g_k_go_fast_graph.hip), reads from the alternate haplotype
    dict(seed=74, npairs=40000, nreads=40000, rdlen=90, sub=0.02, indel=0.002, snps=120, least=0.25),        # ... dense variants, harder reads
    dict(seed=75, npairs=120000, nreads=60000, rdlen=101, sub=0.005, repeat_genome=True),                    # human-like repeat structure: the cold genome hits / pool entries
    dict(seed=76, npairs=100000, nreads=20000, rdlen=101, sub=0.01, repeat_genome=True, env={"H2G_FAST_AM": "1", "H2G_FAST_TAIL": "16"}),   # k_go_fast_am + tail hand-off
    dict(seed=72, npairs=60000, nreads=20000, rdlen=76, sub=0.03, indel=0.002, nrate=0.002, least=0.25, env={"H2G_FAST_AM": "1"}),         # alignMate in the pass on hard reads
    # the end of the batch through the drain launch (k_go_fast_drain: workgroups that hold <= H2G_FAST_ORPHAN reads list them and leave; a small batch needs the knob)
    dict(seed=71, npairs=150000, nreads=150000, rdlen=101, sub=0.005, env={"H2G_FAST_ORPHAN": "64", "H2G_DRAIN_GRID": "8", "H2G_FAST_MATE_HANDOVER": "0"}, adopted=True),
    dict(seed=72, npairs=60000, nreads=60000, rdlen=76, sub=0.03, indel=0.002, nrate=0.002, least=0.25, env={"H2G_FAST_ORPHAN": "512", "H2G_DRAIN_GRID": "32", "H2G_FAST_MATE_HANDOVER": "0"}, adopted=True),
    dict(seed=73, npairs=100000, nreads=100000, rdlen=101, sub=0.005, snps=250, env={"H2G_FAST_ORPHAN": "100", "H2G_DRAIN_GRID": "16"}, adopted=True),   # k_go_fast_graph_drain
    dict(seed=76, npairs=100000, nreads=20000, rdlen=101, sub=0.01, repeat_genome=True, env={"H2G_FAST_AM": "1", "H2G_FAST_TAIL": "16", "H2G_FAST_ORPHAN": "200", "H2G_DRAIN_GRID": "4"}, adopted=True),
    # ... and with the pairs that need alignMate parked for it (the drain launch is then the alignMate build: k_go_fast_am_drain behind k_go_fast) — the policy of large batches
    dict(seed=76, npairs=100000, nreads=20000, rdlen=101, sub=0.01, repeat_genome=True, env={"H2G_FAST_ORPHAN": "200", "H2G_DRAIN_GRID": "8", "H2G_FAST_MATE_HANDOVER": "1"}, adopted=True),
    dict(seed=72, npairs=60000, nreads=20000, rdlen=76, sub=0.03, indel=0.002, nrate=0.002, least=0.25, env={"H2G_FAST_ORPHAN": "64", "H2G_DRAIN_GRID": "16", "H2G_FAST_MATE_HANDOVER": "1"}, adopted=True),
    dict(seed=77, npairs=150000, nreads=20000, rdlen=101, sub=0.005, env={"H2G_FAST_ORPHAN": "512", "H2G_DRAIN_GRID": "32", "H2G_FAST_MATE_HANDOVER": "1", "H2G_FAST_TAIL": "16"}, adopted=True),
])
def test_fast_pass_equals_the_machine(case):
    tmp = tempfile.mkdtemp(prefix="h2fp")
    if case.get("repeat_genome"):
        contigs = synth.make_repeat_genome([5000000, 2000000, 1000000], case["seed"])
    else:
        contigs = synth.make_genome([1500000, 400000, 100000], case["seed"], n_gaps=3, gap_len=300, repeats=80, repeat_len=600)
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, contigs)
    base = os.path.join(tmp, "g")
    if case.get("snps"):
        var = synth.make_snps(contigs, case["seed"] + 9, every=case["snps"])
        synth.write_snps(os.path.join(tmp, "g.snp"), var)
        subprocess.run([BUILD, "-q", "-p", "16", "--snp", os.path.join(tmp, "g.snp"), fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        contigs = synth.apply_snps(contigs, var)
    else:
        subprocess.run([BUILD, "-q", "-p", "16", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    m1, m2 = synth.make_pairs(contigs, case["npairs"], case["rdlen"], case["seed"] + 1, frag_mean=300, frag_sd=40, sub_rate=case["sub"])
    reads, _ = synth.make_reads(contigs, case["nreads"], case["rdlen"], case["seed"] + 2, sub_rate=case["sub"], indel_rate=case.get("indel", 0.0), n_rate=case.get("nrate", 0.0))
    npz = os.path.join(tmp, "reads.npz")
    np.savez(npz, m1=np.stack(m1), m2=np.stack(m2), reads=np.asarray(reads))
    got = {}
    for fast in ("0", "1"):
        env = dict(os.environ, H2G_GO_FAST=fast, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"), **(case.get("env", {}) if fast == "1" else {}))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fast_digest.py"), base, npz], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[fast] = json.loads(r.stdout.strip().splitlines()[-1])
    print(got)
    for k in ("pairs", "reads"):
        assert got["0"][k]["fast"] == 0
        assert got["1"][k]["fast"] + got["1"][k]["handed_on"] == (case["npairs"] if k == "pairs" else case["nreads"])
        assert got["1"][k]["fast"] > case.get("least", 0.5) * (case["npairs"] if k == "pairs" else case["nreads"]), got["1"][k]      # the fast pass is the path
        assert got["0"][k]["overflow"] == 0 and got["1"][k]["overflow"] == 0
        assert got["0"][k]["aligned"] == got["1"][k]["aligned"]
        assert got["0"][k]["sha"] == got["1"][k]["sha"], k
        if case.get("adopted"): assert got["1"][k]["adopted"] > 0, got["1"][k]
