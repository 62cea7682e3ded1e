"""Read sharding across GPUs: contiguous read-id ranges, index replicated, and ONE collective —
the sum of the alignment-summary counters (cf. ReportingMetrics, aln_sink.h:51) over RCCL (backend "nccl" on ROCm;
"gloo" in the CPU tests)."""
import numpy as np


def shard_range(n_total: int, rank: int, world: int):
    """[lo, hi) of read ids owned by `rank`: k*N/G .. (k+1)*N/G, identical to -s/--skip and -u/--upto restarts."""
    lo = (n_total * rank) // world
    hi = (n_total * (rank + 1)) // world
    return lo, hi


def summarize(seed_results, read_len=None):
    """Per-shard summary counters from the fused stage results (SEED_RESULT_DTYPE array, 2 entries per read)."""
    from . import api
    r = seed_results
    valid = np.arange(api.SEED_CAP)[None, :] < r["ncoords"][:, None]
    full = (r["ext"]["score"] == 0) & valid
    if read_len is not None:
        full &= r["ext"]["len"] == read_len
    return np.array([len(r) // 2,
                     int((r["ncoords"] > 0).reshape(-1, 2).any(axis=1).sum()),
                     int(full.any(axis=1).reshape(-1, 2).any(axis=1).sum()),
                     int(r["hit"]["nrank"].sum()), int(r["hit"]["nside"].sum()), int(r["nsteps"].sum()),
                     int(r["ncoords"].sum())], dtype=np.int64)


def all_reduce_sum(arr: np.ndarray, dist=None, device=None):
    """Sum `arr` over all ranks (no-op for a single process)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return arr
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def all_gather_junctions(sites: np.ndarray, dist=None, device=None):
    """Spliced alignment with temporary splice sites, sharded (hisat2.cpp:3687, splice_site.cpp:190-347): a wave of W reads
    is cut into one shard per rank — a read never sees the junctions of its own wave — and between two waves every rank needs every
    shard's new junctions.  `sites`: this rank's [k, 5] int64 rows (text, left, right, dir, read id).  Returns the rows of all ranks
    concatenated in rank order (= read-id order of the shards); the caller merges them into its database keeping the smallest read id
    per (text, left, right, dir), which is what SpliceSiteDB::addSpliceSite does whatever the arrival order.  Two collectives: the row
    counts, then the padded rows (tens of bytes per junction)."""
    sites = np.ascontiguousarray(sites, dtype=np.int64).reshape(-1, 5)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return sites
    import torch
    world = dist.get_world_size()
    cnt = torch.tensor([sites.shape[0]], dtype=torch.int64)
    if device is not None:
        cnt = cnt.to(device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    m = max(counts) if counts else 0
    if m == 0:
        return sites
    pad = np.zeros((m, 5), dtype=np.int64)
    pad[:sites.shape[0]] = sites
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return np.concatenate([p.cpu().numpy()[:c] for p, c in zip(parts, counts)], axis=0)
