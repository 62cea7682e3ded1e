"""N>1 path on CPU: two gloo processes each own a contiguous read-id range, run the stage on their shard (host
instantiation of the device functions) and all-reduce the summary counters; the result must equal the unsharded run."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

import parity_cases as PC
from hisat2_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, base, golden, q):
    import torch.distributed as dist
    from h2gemu_py import Emu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    reads, _ = PC.load_reads(golden)
    lo, hi = shard.shard_range(len(reads), rank, world)
    mine = reads[lo:hi]
    e = Emu(base)
    offs = (np.arange(len(mine) + 1, dtype=np.uint64) * mine.shape[1]).astype(np.uint32)
    e.set_reads(mine.reshape(-1), offs)
    res = e.seed_extend(pseudogeneStop=0)
    tot = shard.all_reduce_sum(shard.summarize(res, read_len=mine.shape[1]), dist)
    if rank == 0:
        q.put(tot.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single(g1_index, golden_dir):
    from h2gemu_py import Emu
    reads, offs = PC.load_reads(golden_dir)
    e = Emu(g1_index)
    e.set_reads(reads.reshape(-1), offs)
    want = shard.summarize(e.seed_extend(pseudogeneStop=0), read_len=reads.shape[1])
    assert [shard.shard_range(401, r, 2) for r in range(2)] == [(0, 200), (200, 401)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, g1_index, golden_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == want.tolist()
    assert got[0] == len(reads) and got[1] > 300


def _go_worker(rank, world, port, base, golden, q):
    """one shard of the batch through the whole go() machine (host instantiation), rendered as SAM fields"""
    import torch.distributed as dist
    import h2o_py as H
    import sam_util as SU
    from h2gemu_align import emu_align
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    names, seqs = H.read_fasta_reads(os.path.join(golden, "reads_se.fa.gz"))
    lo, hi = shard.shard_range(len(seqs), rank, world)
    outs, recs = emu_align(base, seqs[lo:hi], names[lo:hi])      # the names travel with the reads: they seed the per-read PRNG
    refnames, _ = SU.parse_sam(os.path.join(golden, "ref_se_nospliced.sam.gz"))
    got = SU.render(outs, recs, refnames, seqs[lo:hi], names[lo:hi])
    naln = np.array([sum(1 for o in outs if o.nselect > 0), hi - lo], dtype=np.int64)
    tot = shard.all_reduce_sum(naln, dist)
    q.put((rank, [(nm, got[nm]) for nm in names[lo:hi]], tot.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_go_concatenation_equals_unsharded(g1_index, golden_dir):
    """N>1 of the product path: each rank runs HI_Aligner::go on its id range; the shards' records concatenated in rank order are
    the reference's SAM of the whole file, and the one collective (alignment-count all-reduce) gives the whole-file count"""
    import h2o_py as H
    import sam_util as SU
    names, seqs = H.read_fasta_reads(os.path.join(golden_dir, "reads_se.fa.gz"))
    _, want = SU.parse_sam(os.path.join(golden_dir, "ref_se_nospliced.sam.gz"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_go_worker, args=(r, 2, port, g1_index, golden_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    parts = sorted([q.get(timeout=300) for _ in range(2)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cat = [x for _, part, _ in parts for x in part]
    assert [nm for nm, _ in cat] == names                       # ordered concatenation = --reorder output
    for nm, rec in cat:
        assert rec == want[nm], nm
    assert parts[0][2] == parts[1][2] == [sum(1 for nm in names if want[nm][0][0] != 4), len(names)]


def _wave_worker(rank, world, port, base, rfa, P, q):
    """temporary splice sites, sharded: this rank runs its shard of every wave of 1000 x P reads and all-gathers the junctions"""
    import torch.distributed as dist
    import h2o_py as H
    import temp_splice as TS
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    names, seqs = H.read_fasta_reads(rfa)
    got, _ = TS.wave_run(base, seqs, names, P, lambda lo, hi, o, r, a, k, W: TS.format_wave(base, seqs, names, lo, hi, o, r, a, k, W),
                         rank=rank, world=world, exchange=lambda rows: shard.all_gather_junctions(rows, dist))
    q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_temporary_splice_site_waves_equal_the_reference(tmp_path):
    """SURVEY §8(e) for the reference's DEFAULT mode: every wave of 1000 x P reads is cut into two shards (one per rank), the ranks
    exchange the wave's junctions (shard.all_gather_junctions: smallest read id per site, whatever the arrival order) before the next
    wave, and the shards' lines concatenated wave by wave, rank by rank, are `hisat2 -p P --reorder` byte for byte"""
    import subprocess
    import pytest
    import fuzz_spliced as F
    import sam_lines as SL
    from hisat2_amd import synth
    from test_sam_lines import diff_lines
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
    if not os.path.exists(os.path.join(ref, "hisat2-align-s")):
        pytest.skip("needs oracle/_ref")
    P, n = 2, 7000
    contigs, reads, _ = F.make_case(4711, n, sub=0.01)
    t = str(tmp_path)
    fa, rfa, base = os.path.join(t, "g.fa"), os.path.join(t, "r.fa"), os.path.join(t, "g")
    synth.write_fasta(fa, contigs, names=F.contig_names(contigs))
    subprocess.run([os.path.join(ref, "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    synth.write_reads_fasta(rfa, reads)
    subprocess.run([os.path.join(ref, "hisat2-align-s"), "-f", "-p", str(P), "--reorder", "-x", base, "-U", rfa, "-S", os.path.join(t, "ref.sam")],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(ref, "hisat2-align-s"), "-f", "-p", str(P), "--reorder", "--no-temp-splicesite", "-x", base, "-U", rfa, "-S", os.path.join(t, "nt.sam")],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wave_worker, args=(r, 2, port, base, rfa, P, q)) for r in range(2)]
    for p in procs:
        p.start()
    parts = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    nwaves = len(parts[0])
    assert nwaves == (n + 1000 * P - 1) // (1000 * P)
    got = [l for w in range(nwaves) for r in range(2) for l in parts[r][w][1]]
    want = SL.body_lines(os.path.join(t, "ref.sam"))
    assert diff_lines(got, want) == 0
    assert sum(1 for a, b in zip(want, SL.body_lines(os.path.join(t, "nt.sam"))) if a != b) > n // 50      # the database matters on this input
