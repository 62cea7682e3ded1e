"""CPU tests of the *device* item functions (hisat2_amd/csrc/h2g_core.h) instantiated on the host by
tests/emul — same golden vectors as the oracle, plus emul == oracle on fresh seeded inputs."""
import pytest

import h2o_py as H
import parity_cases as PC
from h2gemu_py import Emu
from hisat2_amd import synth


@pytest.fixture(scope="module")
def emu(g1_index, golden_dir):
    e = Emu(g1_index)
    reads, offs = PC.load_reads(golden_dir)
    e.set_reads(reads.reshape(-1), offs)
    return e


def test_rank(emu, golden_dir):
    PC.check_rank(emu.rank, golden_dir)


def test_fm_search(emu, golden_dir):
    assert PC.check_fm_search(emu, golden_dir, "probe_psearch.txt.gz", 0) == 800
    assert PC.check_fm_search(emu, golden_dir, "probe_psearch_spliced.txt.gz", 1) == 800


def test_coords(emu, golden_dir):
    assert PC.check_coords(emu, golden_dir) > 300


def test_extend(emu, golden_dir):
    assert PC.check_extend(emu, golden_dir) > 1000


def test_seed_stage_matches_oracle(oracle_lib, g1_index, golden_dir):
    # fresh reads (incl. Ns and indels) drawn from the golden genome
    contigs = PC.load_contigs(golden_dir)
    reads, _ = synth.make_reads(contigs, 600, 101, 77, sub_rate=0.02, indel_rate=0.001, n_rate=0.002)
    codes, offs = synth.flatten_reads(reads)
    e = Emu(g1_index)
    e.set_reads(codes, offs)
    oix = H.load_index(oracle_lib, g1_index)
    for pseudo in (0, 1):
        got = e.seed_extend(pseudogeneStop=pseudo)
        want = PC.oracle_seed_extend(oracle_lib, oix, reads, pseudo)
        PC.assert_seed_equal(got, want)
    assert (got["ncoords"] > 0).sum() > 300
