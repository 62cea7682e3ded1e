#!/usr/bin/env python3
"""bench.py — throughput of the MI355X seed-and-extend hot path (BASELINE.json metric: reads/sec; Occ-rank HBM GB/s).

One "step" = one pass of the hot path over one resident batch of synthetic reads:
    both strands' FM backward search (partialSearch, hi_aligner.h:6361) -> SA-offset resolution of the anchor
    ranges (getGenomeCoords, hi_aligner.h:5774) -> 0-mismatch extension (GenomeHit::extend, hi_aligner.h:2031)
Workload at N=1 = BASELINE.json configs[1]: E. coli-size linear index, 1 M synthetic 101 bp SE reads.  The E. coli
FASTA cannot be fetched here (no network), so the genome is the seeded uniform-random 4.9 Mbp substitute that
SURVEY.md §8(d) prescribes; that is stated in `config` and `data`.

Launch:  python bench.py [--gpus N --steps K --warmup W]        (N>1 via torch.distributed.run, one rank per GPU)
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming copy)
SEED = 20260925 + 2     # config #2


def build_index(cache, genome_len):
    """Index of the seeded substitute genome.  Index construction is outside the hot path: the
    .ht2 files are produced once by the reference's own builder (oracle/_ref/hisat2-build-s, prebuilt in the
    build container) and cached; the on-disk format is only ever read by this framework."""
    from hisat2_amd import synth
    base = os.path.join(cache, f"rnd{genome_len}_s{SEED}")
    contigs = synth.make_genome([genome_len], SEED)
    if not all(os.path.exists(f"{base}.{k}.ht2") for k in range(1, 7)):
        os.makedirs(cache, exist_ok=True)
        builder = os.path.join(ROOT, "oracle", "_ref", "hisat2-build-s")
        if not os.path.exists(builder):
            raise SystemExit("bench.py: no cached index and oracle/_ref/hisat2-build-s is missing; run __graft_entry__.build() where /root/reference exists")
        fa = base + ".fa"
        synth.write_fasta(fa, contigs, names=["ecoli_substitute"])
        subprocess.run([builder, "-q", fa, base + ".tmp"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for k in range(1, 9):
            os.replace(f"{base}.tmp.{k}.ht2", f"{base}.{k}.ht2")
        os.remove(fa)
    return base, contigs


def cpu_baseline(base, reads, sample):
    """Oracle ("port") leg: the same stage computed by the scalar C restatement on ONE host core, over a bounded
    sample of the same reads.  Reported baseline, not the target."""
    import h2o_py as H
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    olib = H.load()
    oix = H.load_index(olib, base)
    sub = np.ascontiguousarray(reads[:sample])
    offs = (np.arange(sample + 1, dtype=np.uint64) * reads.shape[1]).astype(np.uint32)
    cnt = (C.c_uint64 * 4)()
    t0 = time.perf_counter()
    ck = olib.h2o_seed_extend_batch(oix, sub.ctypes.data, offs.ctypes.data, sample, 0, 5, cnt)
    dt = time.perf_counter() - t0
    return {"value": sample / dt, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": f"first {sample} reads of the bench batch, oracle/h2o.c h2o_seed_extend_batch, {dt:.2f} s",
            "ranks_per_read": cnt[0] / sample, "sa_steps_per_read": cnt[1] / sample, "checksum": int(ck)}


def verify_sample(base, reads, got, nver):
    import h2o_py as H
    import parity_cases as PC
    olib = H.load()
    oix = H.load_index(olib, base)
    want = PC.oracle_seed_extend(olib, oix, reads[:nver], 0)
    PC.assert_seed_equal(got[:2 * nver], want)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--genome-len", type=int, default=4_900_000)
    ap.add_argument("--cpu-sample", type=int, default=200_000)
    ap.add_argument("--rank-queries", type=int, default=1 << 26)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    from hisat2_amd import api, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)   # "nccl" == RCCL on ROCm

    cache = os.path.join(ROOT, ".bench_cache")
    if rank == 0:
        base, contigs = build_index(cache, a.genome_len)
    if dist is not None:
        dist.barrier()
    if rank != 0:
        base, contigs = build_index(cache, a.genome_len)

    # reads are sharded by id range: rank r owns ids [r*n, (r+1)*n) — per-GPU work is fixed (weak scaling)
    reads, _ = synth.make_reads(contigs, a.reads, 101, SEED + 1000 * (rank + 1), sub_rate=0.005)
    codes, offs = synth.flatten_reads(reads)
    ix = api.Index(base, device=local)
    st = api.Stream(ix, max_reads=a.reads, max_bases=codes.size)
    st.set_reads(codes, offs)                 # inputs resident in HBM before the timed region
    params = st.seed_params(no_spliced=True)  # config 2 runs --no-spliced-alignment

    def barrier():
        st.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        st.seed_extend_run(params)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st.seed_extend_run(params)
    barrier()
    dt = time.perf_counter() - t0
    cnt = st.counters()                        # counters + HIP-event kernel times of the LAST step
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # final alignment-count reduction over RCCL/xGMI (the only collective on this path)
    got = st.seed_extend_fetch()
    from hisat2_amd import shard
    sm = shard.summarize(got, read_len=101)    # [reads, anchored, fully extended, n_rank, n_side, n_sa_steps, n_ext]
    sm = shard.all_reduce_sum(sm, dist, device="cuda")
    summ = np.array([sm[1], sm[2], sm[3], sm[4], sm[5], sm[6]], dtype=np.int64)

    if rank == 0:
        total_reads = a.reads * world
        value = total_reads * a.steps / dt
        # dominant kernel = the FM search kernel (k_seed_search); algorithmic bytes = unique sides visited x 64 B
        ms_search, ms_re = float(cnt.ms_search), float(cnt.ms_resolve_extend)
        dom_is_search = ms_search >= ms_re
        alg_bytes = cnt.n_side * 64 if dom_is_search else (cnt.n_sa_steps * 64 + cnt.n_ext * 28 + cnt.n_ext * 4)
        dom_ms = ms_search if dom_is_search else ms_re
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "kernel": "k_seed_search" if dom_is_search else "k_seed_resolve_extend",
                    "kernel_ms": dom_ms, "algorithmic_bytes_per_launch": int(alg_bytes),
                    "other_kernel_ms": ms_re if dom_is_search else ms_search,
                    "note": "index is 1.2 MB of sides (L2-resident at E. coli scale): the HBM-scale Occ-rank number is rank_microbench"}
        # Occ-rank micro-kernel at GRCh38 scale: 15.3 M synthetic 64 B sides (0.98 GB), uniform rows
        rix = api.Index(synth_sides=15_300_000, seed=SEED, device=local)
        rst = api.Stream(rix)
        micro = {}
        for v, name in ((0, "lane_per_side"), (1, "4_lanes_per_side"), (2, "8_lanes_per_side")):
            rst.rank_synth(a.rank_queries, SEED, variant=v, repeats=1)
            ms, ck = rst.rank_synth(a.rank_queries, SEED, variant=v, repeats=3)
            gbs = a.rank_queries * 64 / (ms * 1e-3) / 1e9
            micro[name] = {"ms": ms, "GB/s": gbs, "frac_of_8TBs": gbs / HBM_PEAK_GBS, "checksum": int(ck)}
        rst.close()
        rix.close()
        nver = 2000
        verify_sample(base, reads, got, nver)
        out = {
            "metric": "reads/sec, 101 bp SE, seed-and-extend hot path (FM backward search both strands + SA resolve + 0-mm extend)",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "configs[1]: E. coli-size linear GFM, 1M synthetic 101 bp SE reads per GPU, 1xMI355X",
                       "genome": f"seeded uniform-random {a.genome_len} bp substitute for NC_008253 (no network)",
                       "reads_per_gpu": a.reads, "read_len": 101, "sub_rate": 0.005, "mode": "--no-spliced-alignment",
                       "stage": "a11 partialSearch x2 strands -> a14 getGenomeCoords (<=5 rows) -> a18 extend(mm=0); full HI_Aligner::go() state machine is the next §8 row",
                       "sharding": f"reads by id range across {world} GPU(s), index replicated; RCCL all-reduce of summary counters only"},
            "roofline": roofline,
            "rank_microbench": {"sides": 15_300_000, "bytes": 15_300_000 * 64, "queries": a.rank_queries, **micro},
            "counters": {"reads_with_anchor": int(summ[0]), "reads_fully_extended_0mm": int(summ[1]), "n_rank": int(summ[2]),
                         "n_side": int(summ[3]), "n_sa_steps": int(summ[4]), "n_ext": int(summ[5]),
                         "ranks_per_read": float(summ[2]) / total_reads, "sides_per_read": float(summ[3]) / total_reads},
            "parity": {"checked_reads": nver, "against": "oracle/h2o.c", "bit_exact": True},
        }
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(base, reads, min(a.cpu_sample, a.reads))
        print(json.dumps(out))
    st.close()
    ix.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
