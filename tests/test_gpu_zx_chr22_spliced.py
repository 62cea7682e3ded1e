"""-m gpu: the reference's example reads through its DEFAULT (spliced) mode — split from test_gpu_chr22.py so that the files exercising
the units with the splice-site database (whose code changed after the round's last GPU run) sort behind every file of the unspliced
units: under `-x` a failure here cannot hide those."""
import os
import subprocess

import pytest

import sam_lines as SL
from test_sam_lines import diff_lines
from test_gpu_chr22 import chr22, needs_ref, CLI, REF  # noqa: F401  (chr22 is the module-scoped fixture)

pytestmark = pytest.mark.gpu


@needs_ref
@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("mode", ["default", "notemp"])
def test_example_reads_spliced(chr22, paired, mode):
    """the reference's example through its DEFAULT (spliced) mode on the linear index — these reads carry …200N… CIGARs —
    at -p 2 (--reorder): temporary splice sites on (waves of 2000 reads) and off, every line and the summary identical"""
    d = chr22
    tag = f"spl_{'pe' if paired else 'se'}_{mode}"
    inputs = ["-1", os.path.join(d, "ex_1.fa"), "-2", os.path.join(d, "ex_2.fa")] if paired else ["-U", os.path.join(d, "ex_1.fa")]
    opts = ["--no-temp-splicesite"] if mode == "notemp" else []
    ref_sam, amd_sam = os.path.join(d, tag + ".ref.sam"), os.path.join(d, tag + ".amd.sam")
    subprocess.run([os.path.join(REF, "hisat2-align-s"), "-f", "-p", "2", "--reorder", "-x", os.path.join(d, "lin"), "-S", ref_sam] + inputs + opts,
                   check=True, stdout=subprocess.DEVNULL, stderr=open(os.path.join(d, tag + ".ref.err"), "w"))
    subprocess.run([CLI, "-f", "-p", "2", "-x", os.path.join(d, "lin"), "-S", amd_sam] + inputs + opts, check=True, stderr=open(os.path.join(d, tag + ".amd.err"), "w"))
    want = SL.body_lines(ref_sam)
    assert sum(1 for l in want if "N" in l.split("\t")[5]) > 20
    assert diff_lines(SL.body_lines(amd_sam), want) == 0
    assert open(os.path.join(d, tag + ".amd.err")).read() == open(os.path.join(d, tag + ".ref.err")).read()
