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
