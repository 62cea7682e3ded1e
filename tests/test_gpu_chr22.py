"""-m gpu: parity on REAL sequence — the reference's own example (BASELINE configs[0]: 1 Mbp of chr22 + 3 689 dbSNP variants +
1000 read pairs, committed as tests/golden/example_*.gz by copying /root/reference/example) — for the linear and the --snp graph
index, single-end and paired, under the option sets that matter on repetitive sequence.  Reads: the example's own, and reads
drawn from the real sequence.  Every SAM line of `hisat2-align-amd` must equal `hisat2-align-s` byte for byte, the alignment
summary too, and no read may leave the device with an overflow flag (the second pass has to absorb them)."""
import gzip
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import sam_lines as SL
from test_sam_lines import diff_lines
from hisat2_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "hisat2_amd", "hisat2-align-amd")
REF = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")
needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "hisat2-align-s")), reason="needs oracle/_ref")


def _gunzip(name, dst):
    with gzip.open(os.path.join(GOLD, name), "rb") as fi, open(dst, "wb") as fo:
        shutil.copyfileobj(fi, fo)


@pytest.fixture(scope="module")
def chr22(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("chr22"))
    fa, snp = os.path.join(d, "22.fa"), os.path.join(d, "22.snp")
    _gunzip("example_22_20-21M.fa.gz", fa)
    _gunzip("example_22_20-21M.snp.gz", snp)
    _gunzip("example_reads_1.fa.gz", os.path.join(d, "ex_1.fa"))
    _gunzip("example_reads_2.fa.gz", os.path.join(d, "ex_2.fa"))
    subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", fa, os.path.join(d, "lin")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", "--snp", snp, fa, os.path.join(d, "snp")], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    # reads drawn from the real sequence (0.5 %/base substitutions, a few indels): 101 bp single-end and --fr pairs
    seq = b"".join(l.strip() for l in open(fa, "rb") if not l.startswith(b">")).upper()
    code = np.full(256, 4, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    contig = code[np.frombuffer(seq, dtype=np.uint8)]
    reads, _ = synth.make_reads([contig], 20000, 101, 2201, sub_rate=0.005, indel_rate=0.0005)
    synth.write_reads_fasta(os.path.join(d, "dr_se.fa"), reads)
    m1, m2 = synth.make_pairs([contig], 10000, 101, 2202, sub_rate=0.005)
    synth.write_reads_fasta(os.path.join(d, "dr_1.fa"), m1)
    synth.write_reads_fasta(os.path.join(d, "dr_2.fa"), m2)
    return d


OPTS = {"default": [], "sensitive": ["--sensitive"], "k10sec": ["-k", "10", "--secondary"], "dp2": ["--bowtie2-dp", "2"],
        "verysensitive": ["--very-sensitive"]}       # -k 30, --max-seeds 60: runs on the large-workspace unit from the start


@needs_ref
@pytest.mark.parametrize("opt", list(OPTS))
@pytest.mark.parametrize("reads", ["example", "drawn"])
@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("index", ["lin", "snp"])
def test_real_sequence_sam_identical(chr22, index, paired, reads, opt):
    d = chr22
    tag = f"{index}_{'pe' if paired else 'se'}_{reads}_{opt}"
    if paired:
        rd = ["-1", os.path.join(d, "ex_1.fa" if reads == "example" else "dr_1.fa"), "-2", os.path.join(d, "ex_2.fa" if reads == "example" else "dr_2.fa")]
    else:
        rd = ["-U", os.path.join(d, "ex_1.fa" if reads == "example" else "dr_se.fa")]
    common = ["-f", "--no-spliced-alignment", "-x", os.path.join(d, index)] + rd + OPTS[opt]
    ref_sam, ref_err = os.path.join(d, tag + ".ref.sam"), os.path.join(d, tag + ".ref.err")
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-p", "1"] + common + ["-S", ref_sam], check=True, stderr=open(ref_err, "w"))
    out, err, stats = os.path.join(d, tag + ".amd.sam"), os.path.join(d, tag + ".amd.err"), os.path.join(d, tag + ".stats")
    subprocess.run([CLI, "-p", "4", "--batch", "6000", "--h2g-stats", stats] + common + ["-S", out], check=True, stderr=open(err, "w"))
    st = json.load(open(stats))
    print(f"{tag}: {st['reads']} reads/pairs, second pass {st['second_pass']} ({100.0 * st['second_pass'] / max(1, st['reads']):.2f} %), still flagged {st['overflow']}")
    assert st["overflow"] == 0
    assert diff_lines(SL.body_lines(out), SL.body_lines(ref_sam)) == 0
    assert open(err).read() == open(ref_err).read()          # alignment summary
