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
