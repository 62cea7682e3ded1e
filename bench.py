#!/usr/bin/env python3
"""bench.py — throughput of the MI355X seed-and-extend hot path (BASELINE.json metric: reads/sec, 101 bp PE on a GRCh38-size
linear index; achieved HBM GB/s on Occ-rank).

One "step" = one pass of the hot path over one resident batch of synthetic read pairs = HI_Aligner::go for every pair
(hi_aligner.h:4048: FM backward search of both mates on both strands, SA-offset resolution, ungapped extension, local-index
search, indel joins, recursion, mate rescue, pairing, sink feedback) — h2g_align_pairs_run: the fast pass (h2g_k_go_fast.hip:
the dominant traces with a compact per-read state), the general machine's pass over the reads the fast pass handed on, and the
large-workspace pass over what overflowed that.  The machine's pass of step k runs on machine stream k mod 8 next to the fast passes of
the following steps (up to eight in flight; the process asks for 16 hardware queues); the K timed steps are bracketed by a full
synchronisation of every stream, so the drain of the last machine passes is inside the timed region.

Parity at the size that is timed: every record of the batch the timed steps ran on goes through the product's sink + SAM text and is
compared with oracle/_ref/hisat2-align-s over the same reads, byte for byte (`parity_whole_batch`; a mismatch prints the line and fails
the run) — the reference binary is the checker, never inside a timed region.

Workload at N=1 = BASELINE.json configs[2]: GRCh38-SIZE linear index (3.1 Gbp, 4.7 GB resident), 101 bp paired-end reads,
--no-spliced-alignment.  The index is the reference builder's (oracle/_ref/hisat2-build-s) over a seeded uniform-random genome in
24 human-profile contigs (no network: GRCh38 itself cannot be fetched), built on the box (~18 min at -p 64) and cached in
.bench_cache/; a staged grch38sim<len>_* index of at least the wanted size is used when present.  H2G_BENCH_GENOME overrides the
size (e.g. 256e6 for a quick run); the size actually used is named in config.workload.  configs[1] (E. coli-size, single-end)
runs as the extra leg "ecoli_se"; the extra legs are skipped when the run is past H2G_BENCH_DEADLINE seconds (default 1150), the two
companion legs with an index build of their own (repeat_pe, graph256_pe) when they would end past H2G_BENCH_BIG_DEADLINE (1250): both are measured by
`bench.py --only-legs` / tools/queued_steps.py in profiles/ whether or not a run reaches them.

Launch:  python bench.py [--gpus N --steps K --warmup W]        (N > 1: N ranks, one per GPU — under torch.distributed.run, or spawned by this
program when it is started on its own; fewer than N visible devices is an error, never a silent N = 1)
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

# (before torch / HIP initialise: one h2g stream is 1 + M HIP streams that must run side by side; ROCclr's default 4 hardware queues would serialise them)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming copy)
SEED = 20260925 + 2
REF = os.path.join(ROOT, "oracle", "_ref")


def small_index(cache, genome_len):
    """configs[1]: the seeded E. coli-size substitute (cached .ht2 files of the reference's own builder)"""
    from hisat2_amd import synth
    base = os.path.join(cache, f"rnd{genome_len}_s{SEED}")
    contigs = synth.make_genome([genome_len], SEED)
    if not all(os.path.exists(f"{base}.{k}.ht2") for k in range(1, 7)):
        os.makedirs(cache, exist_ok=True)
        builder = os.path.join(REF, "hisat2-build-s")
        if not os.path.exists(builder):
            raise SystemExit("bench.py: no cached index and oracle/_ref/hisat2-build-s is missing; run __graft_entry__.build() where /root/reference exists")
        fa = base + ".fa"
        synth.write_fasta(fa, contigs, names=["ecoli_substitute"])
        subprocess.run([builder, "-q", "-p", str(_builder_threads()), fa, base + ".tmp"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for k in range(1, 9):
            os.replace(f"{base}.tmp.{k}.ht2", f"{base}.{k}.ht2")
        os.remove(fa)
    return base, contigs


build_index = small_index     # name used by tools/


def _builder_threads():
    import build_bench_index as BB
    return BB.usable_cpus()


def headline_index(cache, want_total):
    """(base, total bases, how it was obtained): the largest staged GRCh38-scale index, else build `want_total` — inside a budget: the box's
    builder time varies 2.3x between leases (1010 s and 2320 s measured, profiles/r04_NOTES.md §4) and a run that never prints is worth less than
    one that says plainly it fell back.  H2G_BENCH_BUILD_BUDGET seconds (default 1250) for the build; past it the run continues on a
    H2G_BENCH_FALLBACK_GENOME (default 256 Mbp, ~80 s to build) genome of the same contig profile and the line says so in metric and config."""
    import glob
    import build_bench_index as BB
    staged = []
    for f in glob.glob(os.path.join(cache, "grch38sim*_s%d.1.ht2" % BB.SEED)):
        tot = int(os.path.basename(f)[len("grch38sim"):].split("_")[0])
        if BB.have(BB.index_base(tot, cache)):
            staged.append(tot)
    if staged and max(staged) >= want_total:
        tot = max(staged)
        return BB.index_base(tot, cache), tot, "staged"
    budget = float(os.environ.get("H2G_BENCH_BUILD_BUDGET", "1250"))
    t0 = time.time()
    try:
        base = BB.build(want_total, cache=cache, timeout=budget)
        return base, want_total, "built in %.0f s" % (time.time() - t0)
    except subprocess.TimeoutExpired:
        pass
    small = int(float(os.environ.get("H2G_BENCH_FALLBACK_GENOME", "256e6")))
    if small >= want_total:
        raise SystemExit("bench.py: the %d bp index did not build within %.0f s" % (want_total, budget))
    t1 = time.time()
    base = BB.build(small, cache=cache)
    return base, small, "FALLBACK: the %d bp build passed its %.0f s budget on this box and was stopped; this index built in %.0f s" % (want_total, budget, time.time() - t1)


def reference_pairs(base, m1, m2, opts, threads, sam_path=None, upto=None):
    """runs oracle/_ref/hisat2-align-s on pairs; returns wall seconds"""
    from hisat2_amd import synth
    exe = os.path.join(REF, "hisat2-align-s")
    cmd = [exe, "-f", "--no-spliced-alignment", "-p", str(threads), "-x", base, "-1", m1, "-2", m2, "-S", sam_path or "/dev/null"] + list(opts)
    if upto is not None:
        cmd += ["-u", str(upto)]
    t0 = time.perf_counter()
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return time.perf_counter() - t0


def host_cpu_limits():
    """what bounds the reference's thread scaling on this box besides its own locks: the affinity mask and the cgroup CPU quota"""
    lim = {"sched_affinity": len(os.sched_getaffinity(0))}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            lim[path] = open(path).read().strip()
        except OSError:
            pass
    try:
        lim["loadavg"] = open("/proc/loadavg").read().split()[:3]
    except OSError:
        pass
    return lim


def body(path):
    return [l for l in open(path) if not l.startswith("@")]


def whole_batch_parity(api, base, st, codes1, offs1, codes2, offs2, names, khits, f1, f2, ref_opts=(), threads=None, tmp=None, paired=True):
    """Device-side parity at the size that is timed: EVERY record the device produced for the batch the steps ran on — the dense
    fetch of the stream the timed region used, h2g_align_pairs_fetch_dense / h2g_align_fetch_dense — through the product's sink + SAM text
    (include/h2g_sam.h), against the SAM of oracle/_ref/hisat2-align-s over the same read files, complete lines byte for byte.  The reference binary is
    the checker; nothing here is inside a timed region.  -> dict(pairs_checked | reads_checked, sam_lines, sam_lines_differing,
    digest_equal, sha256 of both bodies)."""
    import hashlib
    L = api.lib()
    vp = C.c_void_p
    L.h2g_sam_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.h2g_sam_close.argtypes = [vp]
    L.h2g_sam_set_threads.argtypes = [vp, C.c_int]
    L.h2g_sam_format_paired_dense.argtypes = [vp] + [vp] * 10 + [C.c_size_t, vp, vp, vp, vp, vp, C.c_uint32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.h2g_sam_format_unpaired_dense.argtypes = [vp] + [vp] * 5 + [C.c_size_t, vp, vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    n = len(names)
    threads = threads or min(32, os.cpu_count() or 1)
    exe = os.path.join(REF, "hisat2-align-s")
    ref_sam = os.path.join(tmp, "whole_ref.sam")
    t0 = time.perf_counter()
    cmd = [exe, "-f", "--no-spliced-alignment", "-p", str(min(threads, 16)), "--reorder", "-x", base] + (["-1", f1, "-2", f2] if paired else ["-U", f1]) + ["-S", ref_sam] + list(ref_opts)
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    nb = "".join(names).encode()
    noffs = np.concatenate([[0], np.cumsum([len(q) for q in names])]).astype(np.uint32)
    h = vp()
    if L.h2g_sam_open(base.encode(), C.byref(h)) != 0:
        return {"error": "h2g_sam_open"}
    L.h2g_sam_set_threads(h, threads)
    used = C.c_size_t(0)
    ptr = lambda x: x.ctypes.data if isinstance(x, np.ndarray) else C.addressof(x)
    if paired:
        res, a1, o1, a2, o2 = st.align_pairs_fetch_dense()
        nrec = int(o1[-1] + o2[-1])
        cap = n * 1400 + 6 * int(codes1.size + codes2.size) + 4096
        for _ in range(2):
            buf = np.empty(cap, dtype=np.uint8)
            rc = L.h2g_sam_format_paired_dense(h, codes1.ctypes.data, offs1.ctypes.data, None, nb, noffs.ctypes.data, codes2.ctypes.data, offs2.ctypes.data, None, nb, noffs.ctypes.data,
                                               n, ptr(res), ptr(a1), o1.ctypes.data, ptr(a2), o2.ctypes.data, khits, buf.ctypes.data, cap, C.byref(used))
            if rc == 0:
                break
            cap = used.value + 16
        del a1, a2
    else:
        res, a1, o1 = st.align_fetch_dense()
        nrec = int(o1[-1])
        cap = n * 700 + 3 * int(codes1.size) + 4096
        for _ in range(2):
            buf = np.empty(cap, dtype=np.uint8)
            rc = L.h2g_sam_format_unpaired_dense(h, codes1.ctypes.data, offs1.ctypes.data, None, nb, noffs.ctypes.data, n, ptr(res), ptr(a1), o1.ctypes.data, buf.ctypes.data, cap, C.byref(used))
            if rc == 0:
                break
            cap = used.value + 16
        del a1
    L.h2g_sam_close(h)
    if rc != 0:
        return {"error": "h2g_sam_format_*_dense rc %d" % rc}
    got = buf[:used.value].tobytes()
    del buf
    t_fmt = time.perf_counter() - t0
    with open(ref_sam, "rb") as f:
        want = f.read()
    k = 0
    while want.startswith(b"@", k):                      # the header lines (@HD @SQ @PG) are the file's, not the batch's
        k = want.index(b"\n", k) + 1
    want = want[k:]
    os.remove(ref_sam)
    dg, dw = hashlib.sha256(got).hexdigest(), hashlib.sha256(want).hexdigest()
    out = {("pairs_checked" if paired else "reads_checked"): n, "records_fetched": nrec, "sam_bytes": len(want), "digest_equal": dg == dw, "sha256_device_path": dg, "sha256_reference": dw,
           "against": "oracle/_ref/hisat2-align-s -p %d --reorder over the whole timed batch (complete SAM lines, byte for byte); device side = the dense fetch of the stream the timed steps ran on, "
                      "formatted by h2g_sam_format_%s_dense" % (min(threads, 16), "paired" if paired else "unpaired"),
           "reference_wall_s": t_ref, "fetch_and_format_s": t_fmt}
    if dg == dw:
        out["sam_lines"] = want.count(b"\n"); out["sam_lines_differing"] = 0
    else:
        gl, wl = got.split(b"\n"), want.split(b"\n")
        out["sam_lines"] = len(wl) - 1
        out["sam_lines_differing"] = sum(1 for x, y in zip(gl, wl) if x != y) + abs(len(gl) - len(wl))
        bad = [y.split(b"\t", 1)[0].decode() for x, y in zip(gl, wl) if x != y][:8]
        out["first_differing_reads"] = bad
    return out



KERNEL_SOURCES = ("h2g_k_go_fast.hip", "h2g_k_go_fast_am.hip", "h2g_fast.h", "h2g_core.h", "h2g_align.h", "h2g_graph.h", "h2g_go_args.h", "Makefile")


def _strip_comments(src):
    """C / C++ / make source without comments, blank lines and leading / trailing blanks: what the compiler sees of it"""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c in "\"'":                                           # string / character literal: copied as it is
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == "\\" else 1
            out.append(src[i:j + 1]); i = j + 1
        elif src.startswith("//", i):
            while i < n and src[i] != "\n":
                i += 1
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c); i += 1
    lines = [l.strip() for l in "".join(out).splitlines()]
    return "\n".join(l for l in lines if l)


def kernel_sources_sha16():
    """what the dominant kernel is compiled from (hisat2_amd/csrc: the fast pass's sources, the headers they include, the build flags), comments and
    blank space apart.  A PMC record is only attached to the line when it was taken on exactly this code; otherwise `traffic` is null (a stale
    figure is no figure)."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        txt = open(os.path.join(ROOT, "hisat2_amd", "csrc", f), "r", errors="replace").read()
        if f == "Makefile":
            txt = "\n".join(l.split("#")[0].rstrip() for l in txt.splitlines() if l.split("#")[0].strip())
        else:
            txt = _strip_comments(txt)
        h.update(f.encode()); h.update(txt.encode())
    return h.hexdigest()[:16]


def newest_pmc_record():
    """(path, record) of the newest profiles/rNN_pmc_traffic.json, or (None, None)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        return files[-1], json.load(open(files[-1]))
    except (OSError, ValueError):
        return files[-1], None


def attach_pmc_traffic(roofline, npairs, total, fast_on):
    """HBM traffic of the dominant kernel from the newest committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need separate passes and cannot be
    collected inside this process), per launch like `achieved` — attached ONLY when the record was taken on exactly the kernel sources of this tree and
    on this workload; otherwise `traffic` stays null and the line says why (a stale figure is no figure)."""
    roofline["kernel_sources_sha16"] = kernel_sources_sha16()
    path, pm = newest_pmc_record()
    if pm is None:
        roofline["traffic_note"] = "no PMC record under profiles/"
        return False
    name = os.path.relpath(path, ROOT)
    same_run = pm.get("pairs_per_launch") == npairs and pm.get("genome") == total and pm.get("kernel", "").startswith("k_go_fast") == fast_on
    if same_run and pm.get("kernel_sources_sha16") == roofline["kernel_sources_sha16"]:
        roofline["traffic"] = int(pm["traffic_bytes_per_launch"])
        roofline["traffic_source"] = pm.get("source")
        roofline["traffic_calibration"] = pm.get("calibration")
        roofline["traffic_record"] = name
        return True
    roofline["traffic_note"] = ("%s was taken on kernel sources %s / %s pairs / %s bp; this run is %s / %d / %d: not attached"
                                % (name, pm.get("kernel_sources_sha16"), pm.get("pairs_per_launch"), pm.get("genome"), roofline["kernel_sources_sha16"], npairs, total))
    return False


def _free_port():
    import socket
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    return port


def spawn_ranks(a, torch):
    """`python bench.py --gpus N` started on its own: re-launches itself as N ranks (torch.distributed.run, rendezvous on 127.0.0.1), the same command
    the driver uses.  Refuses when fewer than N devices are visible (a dry run needs none).  Returns the launcher's exit code."""
    if not a.dry_run:
        ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if ndev < a.gpus:
            print("bench.py: %d GPUs requested (--gpus %d), %d visible: refusing to run fewer ranks than asked for" % (a.gpus, a.gpus, ndev), file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this driver
    return subprocess.run(cmd, env=env).returncode


def dry_run(a, torch, shard, world, rank):
    """The launch path without a device: rendezvous (gloo), the id-range shards, the max-over-ranks timing and the one collective of the path (the
    counter sum), with synthetic per-rank counters.  tests/test_bench_launch.py runs it with --gpus 2 here."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n_global = a.pairs if a.strong else a.pairs * world
    lo, hi = shard.shard_range(n_global, rank, world)
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, float(lo), float(hi)], dtype=torch.float64)
    parts = [torch.zeros_like(t) for _ in range(world)]
    if world > 1:
        dist.all_gather(parts, t)
    else:
        parts = [t]
    summ = shard.all_reduce_sum(np.array([hi - lo, rank + 1], dtype=np.int64), dist if world > 1 else None)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "world_size": world, "backend": "gloo" if world > 1 else None, "scaling": "strong" if a.strong else "weak",
                          "pairs_global": int(summ[0]), "rank_sum": int(summ[1]), "shards": [[int(x[1]), int(x[2])] for x in parts],
                          "max_over_ranks_s": max(float(x[0]) for x in parts)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    t_start = time.time()
    deadline = float(os.environ.get("H2G_BENCH_DEADLINE", "1150"))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=1_000_000, help="read pairs per GPU per step")
    ap.add_argument("--batches", type=int, default=10, help="distinct resident batches of --pairs pairs the timed steps walk over (1..16)")
    ap.add_argument("--genome", type=float, default=float(os.environ.get("H2G_BENCH_GENOME", "3.1e9")))
    ap.add_argument("--strong", action="store_true", help="one global batch of --pairs split over the ranks instead of --pairs per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-whole-parity", action="store_true", help="skip the whole-batch device-vs-reference SAM comparison of each leg (the 3 000 / 20 000-pair command-line samples stay)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs (E. coli SE, graph index, micro-benchmarks)")
    ap.add_argument("--rank-queries", type=int, default=1 << 28)
    ap.add_argument("--only-legs", default="", help="comma-separated big legs (repeat_pe, graph256_pe) to run INSTEAD of the headline: prints {leg: ...} and exits")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="pairs available to the reference CPU runs (each thread count takes what ~6 s of it)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: exercises the launch path only (spawn, rendezvous over gloo, shard ranges, the counter all-reduce) and prints a line marked dry_run")
    a = ap.parse_args()

    import torch
    from hisat2_amd import shard

    # --gpus N is the launch contract: either this process is one of N ranks started by torch.distributed.run (WORLD_SIZE == N), or — started on its
    # own with N > 1 — it becomes that launch: N ranks over 127.0.0.1, one per device.  N ranks on fewer visible devices is an error, never a silent N = 1.
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(spawn_ranks(a, torch))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus" % (a.gpus, world))
    if a.dry_run:
        return dry_run(a, torch, shard, world, rank)
    from hisat2_amd import api, synth
    import build_bench_index as BB
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if torch.cuda.device_count() < world or local >= torch.cuda.device_count():
        raise SystemExit("bench.py: %d GPUs requested (--gpus / WORLD_SIZE), %d visible: one rank per device, no sharing" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(hours=2))   # "nccl" == RCCL on ROCm; (rank 0 may build the index behind the first barrier)

    if world > 1:
        # N > 1 measures the sharded hot path only: the CPU baseline, the whole-batch parity against the reference binary, the command-line figures and the companion legs are
        # single-GPU items (they run at N = 1: SURVEY §8(d)) and would only keep N - 1 ranks waiting at the last barrier
        a.no_extras = True
        a.no_cpu_baseline = True
    cache = os.path.join(ROOT, ".bench_cache")
    if a.only_legs:
        out = {}
        for name in a.only_legs.split(","):
            t0 = time.time()
            out[name] = (spliced_leg if name == "spliced_pe" else ONLY_LEGS[name] if name in ONLY_LEGS else BIG_LEGS[name][0])(a, api, synth, local, cache)
            out[name]["leg_seconds"] = time.time() - t0
        print(json.dumps(out))
        return
    want_total = int(a.genome)
    how = None
    if rank == 0:
        base, total, how = headline_index(cache, want_total)
    if dist is not None:
        dist.barrier()
    if rank != 0:
        base, total, how = headline_index(cache, want_total)
    contigs = BB.genome(total)

    # the global read set is N x pairs (weak scaling: per-GPU work fixed) or `pairs` (--strong); rank r owns the id range
    # shard_range(r) of it — exactly what `-s/--skip -u/--upto` restarts of the command line do.  No data-path collective.
    n_global = a.pairs if a.strong else a.pairs * world
    lo, hi = shard.shard_range(n_global, rank, world)
    npairs = hi - lo
    # configs[2] is 10 M pairs: the timed region walks over `--batches` DISTINCT batches of --pairs pairs (default 10), all resident in HBM before it starts
    # (h2g_stream_select_batch: read sets with result rows of their own on one stream), step i on batch i mod B — not one batch K times.  Batch b holds the global
    # ids [b x n_global + lo, b x n_global + hi) of a read set of B x n_global pairs.
    nbatch = max(1, min(int(a.batches), 16))
    ix = api.Index(base, device=local)
    st = api.Stream(ix, max_reads=npairs, max_bases=npairs * 101)
    last_b = (a.warmup + a.steps - 1) % nbatch                     # the batch of the last timed step: the one whose records the parity check reads
    t_upload = 0.0
    for b in range(nbatch):
        bm1, bm2 = synth.make_pairs(contigs, npairs, 101, SEED + 7 + 1000 * rank + 100003 * b, sub_rate=0.005)
        bc1, bo1 = synth.flatten_reads(bm1)
        bc2, bo2 = synth.flatten_reads(bm2)
        bnames = [str(b * n_global + lo + i) for i in range(npairs)]   # FASTA names = decimal ids (they feed genRandSeed, pat.h:55)
        st.select_batch(b)
        t_up0 = time.perf_counter()
        st.set_reads(bc1, bo1); st.set_read_names(bnames); st.set_mates(bc2, bo2, bnames)   # inputs resident in HBM before the timed region
        t_upload = time.perf_counter() - t_up0
        if b == last_b:
            m1, m2, c1, o1, c2, o2, names, id0 = bm1, bm2, bc1, bo1, bc2, bo2, bnames, b * n_global + lo
        if b == 0 and a.warmup > 0:
            # the first of the W warm-up steps runs here, on batch 0, before the other batches are selected: a stream's first run allocates its pools and this batch's
            # result rows, and a batch selected afterwards gets rows of the same size at selection (h2g_stream_select_batch) — nothing is allocated inside the timed region
            st.align_pairs_run(st.align_params()); st.sync()
    del bm1, bm2, bc1, bc2, bnames
    params = st.align_params()

    def barrier():
        st.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    step_no = 1 if a.warmup > 0 else 0
    for _ in range(a.warmup - step_no):
        st.select_batch(step_no % nbatch); st.align_pairs_run(params); step_no += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st.select_batch(step_no % nbatch); st.align_pairs_run(params); step_no += 1
    barrier()
    dt = time.perf_counter() - t0
    cnt = st.counters()                        # counters + HIP-event kernel times (stream events) of the LAST step
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the final alignment-count reduction over RCCL/xGMI — the only collective on this path
    summ = np.array([npairs, int(cnt.n_aligned), int(cnt.n_overflow), int(cnt.n_second_pass), int(cnt.n_side), int(cnt.n_sa_steps), int(cnt.n_rank)], dtype=np.int64)
    summ = shard.all_reduce_sum(summ, dist, device="cuda")

    if rank == 0:
        out = {}
        total_reads = 2 * int(summ[0])
        value = total_reads * a.steps / dt
        # dominant kernel = k_go_fast (the fast pass: every pair enters it, 99 % complete on it).  Algorithmic bytes =
        # 64 B x (unique sides visited by its search loops + its SA-walk steps) of the LAST launch on this rank; duration = HIP events
        # on the launch stream around that launch (h2g_counters.ms_fast_kernel)
        ms_fast = float(cnt.ms_fast_kernel)
        ms_machine = float(cnt.ms_align_kernel)
        fast_on = ms_fast > 0 and int(cnt.n_fast) > 0
        ms_kernel = ms_fast if fast_on else ms_machine
        alg_bytes = ((int(cnt.n_fast_side) + int(cnt.n_fast_sa_steps)) if fast_on else (int(cnt.n_side) + int(cnt.n_sa_steps))) * 64
        achieved = alg_bytes / (ms_kernel * 1e-3) / 1e9 if ms_kernel > 0 else 0.0
        alg_all = (int(cnt.n_side) + int(cnt.n_sa_steps)) * 64
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "kernel": ("k_go_fast_am (h2g_k_go_fast_am.hip: alignMate in the pass)" if int(os.environ.get("H2G_FAST_AM", api.DEFAULT_ALIGN_MATE)) else "k_go_fast (h2g_k_go_fast.hip)") if fast_on else "k_go<false> (h2g_go_kernels.h)", "kernel_ms": ms_kernel,
                    "tail_hand_off": int(os.environ.get("H2G_FAST_TAIL", api.DEFAULT_TAIL)),
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "pairs_completed_by_the_kernel": int(cnt.n_fast), "pairs_handed_on": int(cnt.n_fast_bail),
                    "machine_pass_ms": ms_machine, "machine_pass_note": "k_go<false> over the handed-on pairs (%.2f %%), on one of 8 machine streams next to the fast passes of the following steps" % (100.0 * int(cnt.n_fast_bail) / max(1, npairs)),
                    "whole_step": {"algorithmic_bytes": alg_all, "GB/s": alg_all / (dt / a.steps) / 1e9, "frac": alg_all / (dt / a.steps) / 1e9 / HBM_PEAK_GBS},
                    "sides_per_pair": int(cnt.n_side) / npairs, "sa_steps_per_pair": int(cnt.n_sa_steps) / npairs,
                    "note": "latency chains over scattered 64 B index lines + per-read control; a read stays in its lane between trips (state in registers, hot words in LDS), 44 % of the slot-trips load their slot (DESIGN.md §3.1c)"}
        if fast_on and float(cnt.ms_drain_kernel) > 0:
            # the end of the batch: the fast launch's workgroups leave once they can fetch no more and hold <= 512 reads; what they held (and the pairs that need alignMate) is finished by the
            # drain launch next to the FOLLOWING step's fast launch — its own kernel, its own counters (not in algorithmic_bytes_per_launch above)
            d_bytes = (int(cnt.n_drain_side) + int(cnt.n_drain_sa_steps)) * 64
            roofline["drain_launch"] = {"kernel": "k_go_fast_am_drain (h2g_k_go_fast_am.hip, the loop of h2g_k_go_fast.hip with ADOPT; alignMate in it)" if int(os.environ.get("H2G_FAST_MATE_HANDOVER", "1")) else "k_go_fast_drain",
                                        "kernel_ms": float(cnt.ms_drain_kernel), "reads_taken_up": int(cnt.n_adopted), "algorithmic_bytes_per_launch": d_bytes,
                                        "GB/s": d_bytes / (float(cnt.ms_drain_kernel) * 1e-3) / 1e9, "workgroups": int(os.environ.get("H2G_DRAIN_GRID", "64")),
                                        "note": "runs on the drain stream beside the next step's fast launch; the machine's pass over this step's hand-ons waits for it (DESIGN.md §3.1d)"}
        # HBM traffic of the same kernel on the same workload from this round's committed rocprofv3 --pmc passes (FETCH_SIZE and
        # WRITE_SIZE need separate passes and cannot be collected inside this process); per launch like `achieved`
        attach_pmc_traffic(roofline, npairs, total, fast_on)
        out.update({
            "metric": "reads/sec, 101 bp PE, " + ("GRCh38-size" if total >= 3_000_000_000 else "REDUCED-SIZE (%d bp, GRCh38 contig profile)" % total) + " linear index, --no-spliced-alignment: HI_Aligner::go per pair on the GPU (inputs and report events resident in HBM), SAM-identical to hisat2",
            "value": value, "unit": "reads/s", "n_gpus": world, "rccl_world_size": (dist.get_world_size() if dist is not None else 1), "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if a.strong else "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"configs[2]: GRCh38-{'SIZE' if total >= 3_000_000_000 else 'PROFILE (reduced size)'} linear index over a seeded uniform-random {total} bp genome (24 contigs; {how}), "
                                   f"{nbatch} distinct batches of {npairs} synthetic 101 bp --fr pairs per GPU resident in HBM ({nbatch * npairs} pairs; step i runs batch i mod {nbatch}), --no-spliced-alignment -k 5, 1xMI355X per rank",
                       "distinct_batches": nbatch,
                       "genome_bases": total, "index_device_bytes": int(ix.info.device_bytes), "pairs_per_gpu": npairs, "read_len": 101, "sub_rate": 0.005,
                       "fragment": "N(300, 30) clipped to [150, 600]",
                       "stage": "HI_Aligner::go for both mates + pairing; report events stay in HBM (finishRead / SAM text are host code, SURVEY §8(f) N1)",
                       "pipelining": "the general machine's pass over step k's hand-ons runs on machine stream k mod 8 next to the fast passes of the following steps (up to 8 such passes in flight, buffers 9 deep, 16 hardware queues); all K steps complete inside the timed region",
                       "sharding": f"pairs by id range across {world} GPU(s) ({'one global batch split' if a.strong else 'fixed work per GPU'}), index replicated; RCCL all-reduce of the summary counters only"},
            "roofline": roofline,
            "counters": {"pairs": int(summ[0]), "pairs_with_concordant": int(summ[1]), "pairs_still_flagged_overflow": int(summ[2]),
                         "pairs_second_pass": int(summ[3]), "second_pass_rate": float(summ[3]) / max(1, int(summ[0])),
                         "sides_per_pair": float(summ[4]) / int(summ[0]), "sa_steps_per_pair": float(summ[5]) / int(summ[0])},
        })
        exe = os.path.join(REF, "hisat2-align-s")
        cli = os.path.join(ROOT, "hisat2_amd", "hisat2-align-amd")
        tmp = tempfile.mkdtemp(prefix="h2bench")
        f1, f2 = os.path.join(tmp, "1.fa"), os.path.join(tmp, "2.fa")
        parity_failed = False
        if os.path.exists(exe) and not a.no_cpu_baseline:
            synth.write_reads_fasta(f1, m1, start_id=id0); synth.write_reads_fasta(f2, m2, start_id=id0)
            if not a.no_whole_parity:
                # every record of the batch the timed steps ran on (the stream still holds the last step's results) against the reference binary
                out["parity_whole_batch"] = whole_batch_parity(api, base, st, c1, o1, c2, o2, names, int(params.khits), f1, f2, tmp=tmp)
                parity_failed = not out["parity_whole_batch"].get("digest_equal", False)
        if world > 1:
            out["single_gpu_items"] = "cpu_baseline, parity_whole_batch, cli_end_to_end and the companion legs are measured by the N = 1 run"
        # host buffers in, host buffers out: upload + both passes + compact fetch of the report events (never `value`)
        pool = api.PinnedPool()
        pc1 = pool.array("c1", c1.size); pc1[:] = c1
        pc2 = pool.array("c2", c2.size); pc2[:] = c2
        st.align_pairs_fetch_compact(pinned=pool)                  # (the page-locked result buffers exist before the timed pass, as a streaming caller's do)
        packed = st.pack_names(names)                              # (name bytes + offsets, the form the C ABI takes: a parser's output, not part of the transfer)
        reps = []
        for _ in range(3):
            t1 = time.perf_counter()
            st.set_reads(pc1, o1); st.set_read_names(packed); st.set_mates(pc2, o2, packed)
            t_up = time.perf_counter() - t1
            st.align_pairs_run(params)
            pres, pr1, pb1, pr2, pb2 = st.align_pairs_fetch_compact(pinned=pool)
            reps.append((time.perf_counter() - t1, t_up))
        t_e2e, t_up = sorted(reps)[1]
        out["pcie_inclusive"] = {"pairs": npairs, "seconds": t_e2e, "reads_per_s": 2 * npairs / t_e2e, "upload_s": t_up, "upload_s_first_pageable": t_upload,
                                 "compact_record_bytes": int(pb1[-1] + pb2[-1]),
                                 "note": "h2g_set_reads + h2g_set_read_names + h2g_set_mates + h2g_align_pairs_run + h2g_align_pairs_fetch_compact (40 B + 12 B per edit held "
                                         "per record instead of 424 B), base codes and results in page-locked host memory (h2g_host_alloc), one host thread, nothing overlapped: median of 3"}
        del pres, pr1, pr2
        pool.close()
        if os.path.exists(exe) and not a.no_cpu_baseline:
            ns = npairs
            # parity on THIS config: the first pairs through the whole drop-in path (reads file -> hisat2-align-amd -> SAM) must be
            # byte-identical to the reference's SAM
            nv = min(3000, ns)
            ref_sam, amd_sam = os.path.join(tmp, "ref.sam"), os.path.join(tmp, "amd.sam")
            reference_pairs(base, f1, f2, [], 1, ref_sam, upto=nv)
            r = subprocess.run([cli, "-f", "--no-spliced-alignment", "-p", "8", "-x", base, "-1", f1, "-2", f2, "-u", str(nv), "-S", amd_sam,
                                "--h2g-stats", os.path.join(tmp, "stats.json")], capture_output=True, text=True)
            if r.returncode != 0:
                raise SystemExit("bench.py: hisat2-align-amd failed on the parity sample (rc %d): " % r.returncode + r.stderr[:1500] + " ... " + r.stderr[-300:])
            wa, wb = body(amd_sam), body(ref_sam)
            ndiff = sum(1 for x, y in zip(wa, wb) if x != y) + abs(len(wa) - len(wb))
            out["parity"] = {"pairs_checked": nv, "sam_lines": len(wb), "sam_lines_differing": ndiff,
                             "against": "oracle/_ref/hisat2-align-s -p 1 (complete SAM lines, byte for byte)", **json.load(open(os.path.join(tmp, "stats.json")))}
            if ndiff:
                raise SystemExit(f"bench.py: {ndiff} SAM lines of the parity sample differ from the reference")
            # the reference on this box's host cores, same index, same reads: thread counts 1 .. nproc, each on as many pairs
            # as ~6 s of work at that width; wall time minus the index load (a `-u 1` run at the same width); median of 3 at the best width
            ncpu = os.cpu_count() or 1
            widths = sorted({t for t in (1, 16, 64, 128, ncpu) if t <= ncpu})
            last_rate = 12_000.0                         # pairs/s guess for sizing the first sample; then what the previous width measured
            scan, best = {}, None
            for pth in widths:
                if time.time() - t_start > deadline:
                    break
                # ~5 s of work at the previous width's rate (the reference does not speed up beyond a few dozen threads on this input:
                # its read-batch and output locks, pat.cpp:76-107 / outq.cpp — the scan shows it)
                nsamp = int(min(ns, max(20_000, 5.0 * last_rate * (4 if pth == 16 else 1))))
                t_load = reference_pairs(base, f1, f2, [], pth, upto=1)
                t_run = max(reference_pairs(base, f1, f2, [], pth, upto=nsamp) - t_load, 1e-6)
                rate = 2 * nsamp / t_run
                scan[str(pth)] = {"reads_per_s": rate, "reads_per_s_per_core": rate / pth, "pairs": nsamp, "seconds": t_run, "index_load_s": t_load}
                last_rate = max(2_000.0, nsamp / t_run)
                if best is None or rate > best[0]:
                    best = (rate, pth, t_run, t_load, nsamp)
            reps = [best[0]]
            for _ in range(2):
                if time.time() - t_start > deadline:
                    break
                t_run = max(reference_pairs(base, f1, f2, [], best[1], upto=best[4]) - best[3], 1e-6)
                reps.append(2 * best[4] / t_run)
            med = sorted(reps)[len(reps) // 2]
            out["cpu_baseline"] = {"value": med, "unit": "reads/s", "cores": best[1], "kind": "reference", "host_cpus": ncpu, "host_cpu_limits": host_cpu_limits(),
                                   "reads_per_s_per_core_at_1_thread": scan.get("1", {}).get("reads_per_s_per_core"), "repeats_at_best_width": reps,
                                   "threads_scan": scan,
                                   "sample": f"first {best[4]} pairs of the bench batch, oracle/_ref/hisat2-align-s -p {best[1]} --no-spliced-alignment -S /dev/null on the same "
                                             f"{total} bp index, {best[2]:.1f} s of alignment (index load {best[3]:.1f} s timed by a -u 1 run and subtracted); median of {len(reps)}; "
                                             "every width's own sample is in threads_scan"}
            # end to end on the same sample: reads files -> SAM file, both programs
            t0c = time.perf_counter()
            r = subprocess.run([cli, "-f", "--no-spliced-alignment", "-p", "32", "-x", base, "-1", f1, "-2", f2, "-S", os.path.join(tmp, "e2e.sam")],
                               capture_output=True, text=True, env=dict(os.environ, H2G_CLI_TIMING="1"))
            t_cli = time.perf_counter() - t0c
            out["cli_end_to_end"] = {"pairs": ns, "wall_s": t_cli, "reads_per_s_wall": 2 * ns / t_cli, "host_threads": 32,
                                     "timing": r.stderr.strip().splitlines()[-1] if r.returncode == 0 and r.stderr.strip() else r.stderr[-300:]}
        shutil.rmtree(tmp, ignore_errors=True)
        st.close()
        if not a.no_extras and time.time() - t_start > deadline:
            out["extras_skipped"] = "past H2G_BENCH_DEADLINE = %.0f s (index build included)" % deadline
        elif not a.no_extras:
            try:                                   # the extra legs never cost the headline its line
                out.update(extras(a, api, synth, ix, local, cache))
            except Exception as e:             # noqa: BLE001
                out["extras_error"] = repr(e)[:400]
            # the two legs with an index build of their own run while the budget lasts (H2G_BENCH_BIG_DEADLINE seconds since the start; both
            # are measured with --only-legs in profiles/r04_legs.json whether or not this run reaches them)
            big_deadline = float(os.environ.get("H2G_BENCH_BIG_DEADLINE", "1250"))
            for name, (fn, need) in BIG_LEGS.items():
                if time.time() - t_start + need > big_deadline:
                    out.setdefault("big_legs_skipped", []).append(name)
                    continue
                try:
                    out[name] = fn(a, api, synth, local, cache)
                except Exception as e:         # noqa: BLE001
                    out[name] = {"error": repr(e)[:400]}
        if not a.no_extras and time.time() - t_start < float(os.environ.get("H2G_BENCH_CHAIN_DEADLINE", "1250")):
            # last key of the line: chains of dependent rank queries / graph LF steps with 1-8 chains per lane at the occupancy of the compact-state pass
            # (k_rank_chain, k_glf_chain; tools/chain_bench.py) — in a process of its own with a time limit: measurement kernels never cost the headline its line
            try:
                gbase = os.path.join(cache, f"rnd4900000_s{SEED}_snp", "g")
                cmd = [sys.executable, os.path.join(ROOT, "tools", "chain_bench.py"), str(1 << 21), "64", gbase if os.path.exists(gbase + ".8.ht2") else "none", "compact"]
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(local))))
                out["chain_microbench"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.stdout.strip() else {"error": (r.stderr or "no output")[-300:]}
            except Exception as e:             # noqa: BLE001
                out["chain_microbench"] = {"error": repr(e)[:300]}
        if not a.no_extras and total >= 1_000_000_000:
            # the headline's companion AT THE METRIC'S SIZE on a genome with repeats: same contig profile and size, synth.make_repeat_genome, linear index built
            # here by the reference's builder — only when the time this run has left covers that build (measured: 1.5 x the random genome's; H2G_BENCH_HARD_LIMIT seconds in all);
            # the 256 Mbp repeat leg above is the figure that is always there
            hard = float(os.environ.get("H2G_BENCH_HARD_LIMIT", "1680"))
            try:
                built = float(how.split("built in ")[1].split(" s")[0]) if how and "built in " in how else 600.0
            except (IndexError, ValueError):
                built = 600.0
            left = hard - (time.time() - t_start)
            # genome + FASTA, the build (measured: the repeat-structured 256 Mbp index builds in 0.9-1.5 x the random one's time), reads + timed runs + the reference over one batch;
            # when the metric's size does not fit the time left, the companion runs at the LARGEST size that does (build time is linear in the genome: multiples of 0.25 Gbp,
            # at least 1 Gbp) and its workload says so — a repeat-structured figure four times the 256 Mbp leg's size instead of none
            # (1.5 x: on one box the repeat-structured 3.1 Gbp index built in 0.94 x the random genome's time, on another it was not done after 1.29 x — a shared host's load — and a
            # build that runs into its time limit costs the run ten minutes and leaves nothing: profiles/r06_last/)
            need = lambda g: (1.5 * built + 100.0) * g / total + 150.0
            glen_c = total
            if left < need(total):
                glen_c = int(max(0.0, (left - 150.0) / ((1.5 * built + 100.0) / total)) // 250_000_000) * 250_000_000
            if glen_c < 1_000_000_000:
                out["repeat_grch38size_pe"] = {"skipped": "%.0f s left of %.0f, about %.0f needed at the metric's size, %.0f at 1 Gbp (the random genome's index took %.0f s on this box)" % (left, hard, need(total), need(1e9), built)}
            else:
                name = "repeat_grch38size_pe" if glen_c == total else "repeat_%dMbp_pe" % (glen_c // 1_000_000)
                if glen_c != total:
                    out["repeat_grch38size_pe"] = {"skipped": "%.0f s left of %.0f, about %.0f needed at the metric's size: run at %d bp instead (%s)" % (left, hard, need(total), glen_c, name)}
                try:
                    out[name] = repeat_leg(a, api, synth, local, cache, glen=glen_c, build_timeout=left - 200.0, nbatch=min(4, max(1, int(a.batches))))
                except Exception as e:         # noqa: BLE001
                    out[name] = {"error": repr(e)[:400]}
        legs_failed = [k for k, v in out.items() if isinstance(v, dict) and isinstance(v.get("parity_whole_batch"), dict) and not v["parity_whole_batch"].get("digest_equal", False)]
        if parity_failed or legs_failed:
            out["parity_failed"] = (["headline"] if parity_failed else []) + legs_failed
        print(json.dumps(out), flush=True)
        if parity_failed or legs_failed:      # the line is printed (it says what differs), and the run FAILS: a fast kernel whose results differ from the reference's is not done
            ix.close()
            raise SystemExit("bench.py: the device's records differ from the reference's SAM on the whole batch: %s" % out["parity_failed"])
    else:
        st.close()
    ix.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def extras(a, api, synth, ix_big, local, cache):
    """the other legs: configs[1] (E. coli-size, single-end), the SNP-graph index, Occ-rank and Smith-Waterman micro-kernels"""
    import sam_util as SU
    ex = {}
    exe = os.path.join(REF, "hisat2-align-s")
    base, contigs = small_index(cache, 4_900_000)
    n = 1_000_000
    reads, truth = synth.make_reads(contigs, n, 101, SEED + 1000, sub_rate=0.005)
    codes, offs = synth.flatten_reads(reads)
    ix = api.Index(base, device=local)
    st = api.Stream(ix, max_reads=n, max_bases=codes.size)
    st.set_reads(codes, offs); st.set_read_names([str(i) for i in range(n)])
    for _ in range(10):
        st.align_run()
    st.sync()
    t0 = time.perf_counter()
    for _ in range(32):
        st.align_run()
    st.sync()
    dt = (time.perf_counter() - t0) / 32
    c = st.counters()
    leg = {"workload": "configs[1]: E. coli-size (4.9 Mbp seeded substitute) linear index, 1 M synthetic 101 bp SE reads", "reads": n,
           "ms_per_step": dt * 1e3, "reads_per_s": n / dt, "kernel_ms": float(c.ms_fast_kernel) or float(c.ms_align_kernel), "machine_pass_ms": float(c.ms_align_kernel),
           "aligned": int(c.n_aligned), "fast_pass_completed": int(c.n_fast), "handed_on": int(c.n_fast_bail),
           "second_pass": int(c.n_second_pass), "still_flagged": int(c.n_overflow),
           "roofline_frac": ((int(c.n_fast_side) + int(c.n_fast_sa_steps)) * 64 / (float(c.ms_fast_kernel) * 1e-3) / 1e9 / HBM_PEAK_GBS) if float(c.ms_fast_kernel) > 0
                            else (int(c.n_side) + int(c.n_sa_steps)) * 64 / (float(c.ms_align_kernel) * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if os.path.exists(exe) and not a.no_cpu_baseline:
        tmp = tempfile.mkdtemp(prefix="h2benchs")
        if a.no_whole_parity:
            nv = 3000
            synth.write_reads_fasta(os.path.join(tmp, "r.fa"), reads[:nv])
            subprocess.run([exe, "-f", "-p", "1", "--no-spliced-alignment", "-x", base, "-U", os.path.join(tmp, "r.fa"), "-S", os.path.join(tmp, "r.sam")],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            rn, want = SU.parse_sam(os.path.join(tmp, "r.sam"))
            res, aln = st.align_fetch(0, nv)
            qn = [str(i) for i in range(nv)]
            got = SU.render_selected(res, aln, rn, [reads[i] for i in range(nv)], qn)
            leg["sam_checked_reads"] = nv
            leg["sam_mismatching_reads"] = sum(1 for q in qn if got[q] != want[q])
            if leg["sam_mismatching_reads"]:
                raise SystemExit("bench.py: the E. coli-size leg differs from the reference SAM")
        else:      # all 1 000 000 reads of the timed batch: the device's records -> SAM against the reference binary
            synth.write_reads_fasta(os.path.join(tmp, "r.fa"), reads)
            leg["parity_whole_batch"] = whole_batch_parity(api, base, st, codes, offs, None, None, [str(i) for i in range(n)], int(st.align_params().khits),
                                                           os.path.join(tmp, "r.fa"), None, tmp=tmp, paired=False)
        shutil.rmtree(tmp, ignore_errors=True)
    ex["ecoli_se"] = leg
    # Smith-Waterman kernels (a23-a25, opt-in path of the reference): the first 65536 reads framed around their true position
    nsw = 65536
    swq = [api.SwQuery(i, int(truth[i][2]), int(truth[i][0]), int(truth[i][1]), -20, i + 1) for i in range(nsw)]
    st.sw_align(swq[:1024])
    swres, sw_ms = st.sw_align(swq, repeats=3)
    sw_cells = sum(101 * int(r.refr - r.refl + 1) for r in swres)
    ex["sw_microbench"] = {"problems": nsw, "kernel_ms": sw_ms, "problems_per_s": nsw / (sw_ms * 1e-3), "GCUPS": sw_cells / (sw_ms * 1e-3) / 1e9,
                           "found": sum(1 for r in swres if r.found), "cells_per_problem": sw_cells / nsw}
    st.close()
    # go() on a SNP-GRAPH index (BASELINE configs[3] shape at config-2 size): a seeded variant every ~250 bp, reads from the alternate haplotype
    builder = os.path.join(REF, "hisat2-build-s")
    if os.path.exists(builder):
        gtmp = os.path.join(cache, f"rnd4900000_s{SEED}_snp")
        gbase = os.path.join(gtmp, "g")
        var = synth.make_snps(contigs, SEED + 5, every=250, names=["ecoli_substitute"])
        if not os.path.exists(gbase + ".8.ht2"):
            os.makedirs(gtmp, exist_ok=True)
            synth.write_fasta(gbase + ".fa", contigs, names=["ecoli_substitute"])
            synth.write_snps(gbase + ".snp", var)
            subprocess.run([builder, "-q", "-p", str(_builder_threads()), "--snp", gbase + ".snp", gbase + ".fa", gbase], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        alt = synth.apply_snps(contigs, var, names=["ecoli_substitute"])
        gnp = 500_000
        gm1, gm2 = synth.make_pairs(alt, gnp, 101, SEED + 4343, frag_mean=300, frag_sd=30, sub_rate=0.005)
        gq = [str(i) for i in range(gnp)]
        gc1, go1 = synth.flatten_reads(gm1)
        gc2, go2 = synth.flatten_reads(gm2)
        gix = api.Index(gbase, device=local)
        gst = api.Stream(gix, max_reads=gnp, max_bases=gc1.size)
        gst.set_reads(gc1, go1); gst.set_read_names(gq); gst.set_mates(gc2, go2, gq)
        for _ in range(10):                                 # (the pipeline is 8 machine passes deep: filled before the clock starts)
            gst.align_pairs_run()
        gst.sync()
        t0 = time.perf_counter()
        for _ in range(32):
            gst.align_pairs_run()
        gst.sync()
        gdt = (time.perf_counter() - t0) / 32
        gc_ = gst.counters()
        ex["graph_index_pe"] = {"workload": "configs[3] shape at E. coli size: SNP-graph index (a variant every ~250 bp), 500 k pairs from the alternate haplotype",
                                "pairs": gnp, "variants": len(var), "ms_per_step": gdt * 1e3, "reads_per_s": 2 * gnp / gdt, "kernel_ms": float(gc_.ms_align_kernel),
                                "pairs_with_concordant": int(gc_.n_aligned), "second_pass": int(gc_.n_second_pass), "still_flagged": int(gc_.n_overflow),
                                "ranks_per_pair": int(gc_.n_rank) / gnp, "sides_per_pair": int(gc_.n_side) / gnp, "sa_steps_per_pair": int(gc_.n_sa_steps) / gnp,
                                "pairs_completed_by_the_fast_pass": int(gc_.n_fast), "pairs_handed_on": int(gc_.n_fast_bail), "fast_kernel_ms": float(gc_.ms_fast_kernel),
                                "roofline": {"bound": "hbm", "kernel": "k_go_fast_graph (h2g_k_go_fast_graph.hip: the compact-state pass over the graph form of the state) + k_go<true> over the pairs it hands on",
                                             "achieved": (int(gc_.n_side) + int(gc_.n_sa_steps)) * 128 / (gdt) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": (int(gc_.n_side) + int(gc_.n_sa_steps)) * 128 / (gdt) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                             "achieved_counting_rank_queries": (int(gc_.n_rank) + int(gc_.n_sa_steps)) * 128 / (gdt) / 1e9,
                                             "algorithmic": "(sides of the searches' BWT rows + SA-walk steps) x 128 B graph sides of the whole step (a lower bound of the unique sides: the M / F / header sides are not counted), "
                                                            "over the steady-state step time (fast pass and machine pass overlap across steps)"}}
        gst.close(); gix.close()
    ix.close()
    if os.path.exists(builder) and os.path.exists(exe):
        ex["spliced_pe"] = spliced_leg(a, api, synth, local, cache)
    # Occ-rank micro-kernel at GRCh38 scale: 0.98 GB of synthetic sides, uniform rows, one countBt2Side per query
    for graph, nsides, key in ((False, 15_300_000, "rank_microbench"), (True, 7_650_000, "rank_microbench_graph")):
        rix = api.Index(synth_sides=nsides, seed=SEED, device=local, graph=graph)
        rst = api.Stream(rix)
        micro = {"sides": nsides, "bytes": nsides * (128 if graph else 64), "queries": a.rank_queries}
        for v, name in (((0, "lane_per_side"), (1, "8_lanes_per_side")) if graph else ((0, "lane_per_side"), (1, "4_lanes_per_side"), (2, "8_lanes_per_side"))):
            rst.rank_synth(a.rank_queries, SEED, variant=v, repeats=1)
            ms, ck = rst.rank_synth(a.rank_queries, SEED, variant=v, repeats=3)
            gbs = a.rank_queries * (128 if graph else 64) / (ms * 1e-3) / 1e9
            # second denominator: the measured copy bandwidth of the chip (MI355X guide: 6.29 TB/s); a 64 B side is half a 128 B
            # sector pair, so the linear kernel's ceiling in these units is half of that
            micro[name] = {"ms": ms, "GB/s": gbs, "frac_of_8TBs": gbs / HBM_PEAK_GBS, "frac_of_copy_6.29TBs": gbs / 6290.0, "checksum": int(ck)}
            if not graph and v == 0 and a.rank_queries >= (1 << 28):
                # 2^20 sampled outputs of THIS run against GFM::mapLF on the CPU — the oracle's own h2o_rank over the same side array, rebuilt on the host
                # (tests/rank_synth_check.py; the oracle is the checker here, nothing it computes is measured)
                try:
                    import rank_synth_check as RC
                    import h2o_py as HO
                    stride = a.rank_queries >> 20
                    got = rst.rank_synth_sample(stride, 1 << 20)
                    want = RC.sampled_expect(HO.load(), nsides, SEED, a.rank_queries, stride, 1 << 20)
                    nbad = int((want != got).sum())
                    micro["sampled_vs_oracle_mapLF"] = {"samples": int(len(want)), "differing": nbad, "every": stride, "against": "oracle/h2o.c h2o_rank (SideLocus::initFromRow gfm.h:376 + countBt2Side :2958 + fchr)"}
                    # the same kernel without the micro-benchmark's own output stream (k_rank_v0_sampled: every result into the checksum, every 256th stored)
                    rst.rank_synth(a.rank_queries, SEED, variant=10, repeats=1)
                    ms10, ck10 = rst.rank_synth(a.rank_queries, SEED, variant=10, repeats=3)
                    got10 = rst.rank_synth_sample(256, a.rank_queries >> 8)[::max(1, stride // 256)][:1 << 20]
                    gbs10 = a.rank_queries * 64 / (ms10 * 1e-3) / 1e9
                    micro["lane_per_side_no_output_stream"] = {"ms": ms10, "GB/s": gbs10, "frac_of_8TBs": gbs10 / HBM_PEAK_GBS, "checksum": int(ck10), "checksum_equals_lane_per_side": int(ck10) == int(ck),
                                                               "sampled_differing": int((want != got10).sum()),
                                                               "note": "a rank in the aligner feeds the next step of its chain; 4 B per query written back is the micro-benchmark's own traffic"}
                    if nbad or micro["lane_per_side_no_output_stream"]["sampled_differing"] or int(ck10) != int(ck):
                        raise SystemExit("bench.py: the rank micro-benchmark's outputs differ from the oracle's mapLF (%d of %d sampled; checksums %s / %s)" % (nbad, len(want), ck, ck10))
                except SystemExit:
                    raise
                except Exception as e:         # noqa: BLE001
                    micro["sampled_vs_oracle_mapLF"] = {"error": repr(e)[:300]}
        rst.close(); rix.close()
        ex[key] = micro
    return ex


BAIL_REASONS = "none input longpool subsample coords nghits edits depth localhits gsearch nres searched redundant mate npairs partial straddle other indel tail iedges gwalk".split()


def fast_bail_reasons(api, st):
    """hand-ons of the last fast pass by reason (h2g_fast.h FB_*): the development hook reads the counter block of the last run"""
    import ctypes as C
    v = (C.c_ulonglong * 136)()
    L = api.lib()
    L.h2g_go_fast_prof.argtypes = [C.c_void_p, C.c_void_p]
    if L.h2g_go_fast_prof(st.h, v) != 0:
        return None
    return {BAIL_REASONS[k]: int(v[48 + k]) for k in range(len(BAIL_REASONS)) if v[48 + k]}


def sam_parity(base, f1, f2, nv, tmp, opts=()):
    """the first nv pairs through the drop-in command line and through the reference: differing SAM lines (bodies)"""
    exe = os.path.join(REF, "hisat2-align-s")
    cli = os.path.join(ROOT, "hisat2_amd", "hisat2-align-amd")
    ref_sam, amd_sam = os.path.join(tmp, "ref.sam"), os.path.join(tmp, "amd.sam")
    subprocess.run([exe, "-f", "--no-spliced-alignment", "-p", "8", "--reorder", "-x", base, "-1", f1, "-2", f2, "-u", str(nv), "-S", ref_sam] + list(opts), check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([cli, "-f", "--no-spliced-alignment", "-p", "8", "-x", base, "-1", f1, "-2", f2, "-u", str(nv), "-S", amd_sam, "--h2g-stats", os.path.join(tmp, "stats.json")] + list(opts),
                       capture_output=True, text=True)
    if r.returncode != 0:
        return {"error": "hisat2-align-amd rc %d: %s" % (r.returncode, r.stderr[-600:])}
    wa, wb = body(amd_sam), body(ref_sam)
    return {"pairs_checked": nv, "sam_lines": len(wb), "sam_lines_differing": sum(1 for x, y in zip(wa, wb) if x != y) + abs(len(wa) - len(wb)),
            "against": "oracle/_ref/hisat2-align-s -p 8 --reorder (complete SAM lines)", **json.load(open(os.path.join(tmp, "stats.json")))}


def timed_pairs(api, synth, base, local, m1, m2, steps=32, whole_parity=True, more=()):
    """steady-state step of queued runs over (m1, m2) and the batches of `more` ((m1, m2) tuples of the same size): all resident (h2g_stream_select_batch), step i on batch
    i mod B; the counters and the whole-batch parity are batch 0's, run once more on its own"""
    c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
    n = len(m1)
    names = [str(i) for i in range(n)]
    ix = api.Index(base, device=local)
    st = api.Stream(ix, max_reads=n, max_bases=c1.size)
    st.set_reads(c1, o1); st.set_read_names(names); st.set_mates(c2, o2, names)
    nb = 1 + len(more)
    for b, (x1, x2) in enumerate(more, start=1):
        d1, p1 = synth.flatten_reads(x1); d2, p2 = synth.flatten_reads(x2)
        xn = [str(b * n + i) for i in range(n)]
        st.select_batch(b)
        st.set_reads(d1, p1); st.set_read_names(xn); st.set_mates(d2, p2, xn)
    k = 0
    for _ in range(10):                                # (the pipeline is 8 machine passes deep: filled before the clock starts)
        st.select_batch(k % nb); st.align_pairs_run(); k += 1
    st.sync()
    t0 = time.perf_counter()
    for _ in range(steps):                             # 32 steps: the drain of the last machine passes (inside the timed region) weighs 1 / 32
        st.select_batch(k % nb); st.align_pairs_run(); k += 1
    st.sync()
    dt = (time.perf_counter() - t0) / steps
    # the counters of ONE run on its own: in a queue of runs the machine pass of run k - 1 may finish a deferred read that run k also deferred, and
    # run k's count of aligned pairs then misses it (the result itself is the same either way; the headline's workload has no second pass)
    st.select_batch(0)
    st.align_pairs_run(); st.sync()
    c = st.counters()
    leg = {"pairs": n, "distinct_batches": nb, "ms_per_step": dt * 1e3, "reads_per_s": 2 * n / dt, "fast_kernel_ms": float(c.ms_fast_kernel), "machine_pass_ms": float(c.ms_align_kernel),
           "kernel_times_are": "of one run on its own (the step time above is the steady state of queued runs)",
           "pairs_completed_by_the_fast_pass": int(c.n_fast), "pairs_handed_on": int(c.n_fast_bail), "hand_on_rate": int(c.n_fast_bail) / n,
           "hand_ons_by_reason": fast_bail_reasons(api, st), "pairs_second_pass": int(c.n_second_pass), "second_pass_rate": int(c.n_second_pass) / n,
           "pairs_still_flagged_overflow": int(c.n_overflow), "pairs_with_concordant": int(c.n_aligned),
           "ranks_per_pair": int(c.n_rank) / n, "sides_per_pair": int(c.n_side) / n, "sa_steps_per_pair": int(c.n_sa_steps) / n,
           "index_device_bytes": int(ix.info.device_bytes), "_fast_alg_bytes": (int(c.n_fast_side) + int(c.n_fast_sa_steps)) * 64}
    if whole_parity and os.path.exists(os.path.join(REF, "hisat2-align-s")):
        # the whole batch on the device (the stream's results of the run above) against the reference binary, every SAM line
        tmp = tempfile.mkdtemp(prefix="h2whole")
        f1, f2 = os.path.join(tmp, "1.fa"), os.path.join(tmp, "2.fa")
        synth.write_reads_fasta(f1, m1); synth.write_reads_fasta(f2, m2)
        leg["parity_whole_batch"] = whole_batch_parity(api, base, st, c1, o1, c2, o2, names, int(st.align_params().khits), f1, f2, tmp=tmp)
        shutil.rmtree(tmp, ignore_errors=True)
    st.close(); ix.close()
    return leg


def repeat_leg(a, api, synth, local, cache, glen=256_000_000, npairs=1_000_000, nparity=20_000, build_timeout=None, nbatch=1):
    """The headline's companion on a genome WITH repeats: synth.make_repeat_genome — Alu-like and LINE-like families at 8-20 %
    divergence in a quarter of the bases, tandem arrays, segmental duplications — 24 human-profile contigs, linear index built on the box,
    1 M x 2 x 101 bp pairs: the step time, how many pairs the fast pass hands on and why, the second-pass rate, and the SAM of 20 000 pairs
    against the reference."""
    import build_bench_index as BB
    d = os.path.join(cache, f"rep{glen}_s{SEED}")
    base = os.path.join(d, "g")
    contigs = synth.make_repeat_genome(BB.contig_lens(glen), SEED + 77)
    t_build = None
    if not os.path.exists(base + ".8.ht2"):
        os.makedirs(d, exist_ok=True)
        synth.write_fasta(base + ".fa", contigs)
        t0 = time.time()
        try:
            subprocess.run([os.path.join(REF, "hisat2-build-s"), "-q", "-p", str(_builder_threads()), base + ".fa", base + ".tmp"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=build_timeout)
            for k in range(1, 9):
                os.replace(f"{base}.tmp.{k}.ht2", f"{base}.{k}.ht2")
        except subprocess.TimeoutExpired:
            return {"skipped": "the reference's builder did not finish the %d bp repeat-structured index within the %.0f s this run had left for it" % (glen, build_timeout)}
        finally:
            for p_ in [base + ".fa"] + [f"{base}.tmp.{k}.ht2" for k in range(1, 9)]:
                if os.path.exists(p_):
                    os.remove(p_)
        t_build = time.time() - t0
    m1, m2 = synth.make_pairs(contigs, npairs, 101, SEED + 78, sub_rate=0.005)
    more = [synth.make_pairs(contigs, npairs, 101, SEED + 78 + 100003 * b, sub_rate=0.005) for b in range(1, nbatch)]
    leg = {"workload": f"repeat-structured {glen} bp genome (interspersed families of ~300 bp and 1-6 kbp at 8-20 % divergence in a quarter of the bases, tandem arrays, segmental "
                       f"duplications; 24 contigs), linear index, {npairs} x 2 x 101 bp pairs, --no-spliced-alignment -k 5", "index_build_s": t_build}
    leg.update(timed_pairs(api, synth, base, local, m1, m2, whole_parity=not a.no_whole_parity and not a.no_cpu_baseline, more=more))
    del more
    # roofline of this leg: the fast kernel over its own algorithmic bytes, and the whole step (fast + machine passes, steady state)
    alg_fast = leg.pop("_fast_alg_bytes")
    alg_all = (leg["sides_per_pair"] + leg["sa_steps_per_pair"]) * npairs * 64
    pm_rep = None
    for rec in ("r06_rep_pmc_traffic.json", "r05_rep_pmc_traffic.json"):
        try:
            pm_rep = json.load(open(os.path.join(ROOT, "profiles", rec)))
            break
        except (OSError, ValueError):
            pass
    leg["roofline"] = {"bound": "hbm", "kernel": "k_go_fast (h2g_k_go_fast.hip) over the pairs it completes", "kernel_ms": leg["fast_kernel_ms"],
                       "achieved": alg_fast / (leg["fast_kernel_ms"] * 1e-3) / 1e9 if leg["fast_kernel_ms"] > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": alg_fast / (leg["fast_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if leg["fast_kernel_ms"] > 0 else 0.0,
                       "traffic": (int(pm_rep["traffic_bytes_per_launch"]) if pm_rep and pm_rep.get("kernel_sources_sha16") == kernel_sources_sha16() and pm_rep.get("pairs_per_launch") == npairs and int(pm_rep.get("genome", 256000000)) == glen else None),
                       "traffic_record": "profiles/r06_rep_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this leg's fast kernel; attached only on the sources it was taken on)",
                       "algorithmic": "64 B x (unique sides + SA-walk steps) of the fast kernel's own searches and walks",
                       "whole_step": {"algorithmic_bytes": alg_all, "GB/s": alg_all / (leg["ms_per_step"] * 1e-3) / 1e9, "frac": alg_all / (leg["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "note": "fast pass + the general machine's passes over the hand-ons (8 in flight) + second passes, steady state of queued runs"}}
    tmp = tempfile.mkdtemp(prefix="h2rep")
    f1, f2 = os.path.join(tmp, "1.fa"), os.path.join(tmp, "2.fa")
    synth.write_reads_fasta(f1, m1[:nparity]); synth.write_reads_fasta(f2, m2[:nparity])
    leg["parity"] = sam_parity(base, f1, f2, nparity, tmp)
    shutil.rmtree(tmp, ignore_errors=True)
    return leg


def graph256_leg(a, api, synth, local, cache, glen=256_000_000, every=250, npairs=1_000_000, nparity=20_000):
    """configs[3]'s shape beyond toy size: a SNP-graph index over a 256 Mbp genome (a variant about every 250 bp: ~1 M single-base variants,
    deletions and insertions; built on the box by the reference's builder, ~18 GB of builder memory), 1 M pairs from the alternate haplotype."""
    import build_graph_bench_index as GB
    import build_bench_index as BB
    base, info = GB.build(glen, every, cache=cache)
    if info is None and os.path.exists(base + ".build.json"):
        info = json.load(open(base + ".build.json"))
    contigs = BB.genome(glen)
    alt = synth.apply_snps(contigs, GB.variants(glen, every, contigs), names=GB.names(glen))
    m1, m2 = synth.make_pairs(alt, npairs, 101, SEED + 79, frag_mean=300, frag_sd=30, sub_rate=0.005)
    leg = {"workload": f"configs[3] shape: SNP-graph index over a seeded {glen} bp genome, a variant every ~{every} bp, {npairs} x 2 x 101 bp pairs from the alternate haplotype, --no-spliced-alignment",
           "index_build": info}
    leg.update(timed_pairs(api, synth, base, local, m1, m2, whole_parity=not a.no_whole_parity and not a.no_cpu_baseline))
    leg.pop("_fast_alg_bytes", None)
    # algorithmic bytes (round 6): the graph searches count the sides of their BWT rows (one when top and bot share a side; the sides their M / F bit vectors and header
    # back-scans add are not counted) and every SA-walk step is at least its row's side: a LOWER bound of the unique sides SURVEY §8(d) asks for.  Through round 5 the numerator
    # was rank queries (an upper bound); it is kept next to it.
    alg = (leg["sides_per_pair"] + leg["sa_steps_per_pair"]) * npairs * 128
    alg_queries = (leg["ranks_per_pair"] + leg["sa_steps_per_pair"]) * npairs * 128
    pm_g = None
    for rec in ("r06_graph_pmc_traffic.json", "r05_graph_pmc_traffic.json"):
        try:
            pm_g = json.load(open(os.path.join(ROOT, "profiles", rec)))
            break
        except (OSError, ValueError):
            pass
    leg["roofline"] = {"bound": "hbm", "kernel": "k_go_fast_graph + k_go<true> over the hand-ons (whole step)", "achieved": alg / (leg["ms_per_step"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": alg / (leg["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "achieved_counting_rank_queries": alg_queries / (leg["ms_per_step"] * 1e-3) / 1e9,
                       "traffic": (int(pm_g["traffic_bytes_per_launch"]) if pm_g and pm_g.get("kernel_sources_sha16") == kernel_sources_sha16() and pm_g.get("pairs_per_launch") == npairs
                                   and int(pm_g.get("genome", 256000000)) == glen else None),
                       "traffic_record": "profiles/r06_graph_pmc_traffic.json: FETCH_SIZE + WRITE_SIZE of k_go_fast_graph per launch (a LOWER bound: the counter tallies a 128 B side request at 64 B, profiles/r04_rank_pmc.json; "
                                         "the record also holds 2 x FETCH_SIZE + WRITE_SIZE); attached only on the sources it was taken on",
                       "algorithmic": "(sides of the searches' BWT rows + SA-walk steps) x 128 B graph sides: a lower bound of the unique sides (the M / F / header sides are not counted)"}
    tmp = tempfile.mkdtemp(prefix="h2g256")
    f1, f2 = os.path.join(tmp, "1.fa"), os.path.join(tmp, "2.fa")
    synth.write_reads_fasta(f1, m1[:nparity]); synth.write_reads_fasta(f2, m2[:nparity])
    leg["parity"] = sam_parity(base, f1, f2, nparity, tmp)
    shutil.rmtree(tmp, ignore_errors=True)
    return leg


def graph_big_leg(a, api, synth, local, cache):
    """the SNP-graph leg at H2G_GRAPH_LEG_GENOME bases (default 1e9: ~4 M variants, ~70 GB of builder memory, several minutes of build) — `--only-legs graph_big_pe` in a lease"""
    return graph256_leg(a, api, synth, local, cache, glen=int(float(os.environ.get("H2G_GRAPH_LEG_GENOME", "1e9"))))


BIG_LEGS = {"repeat_pe": (repeat_leg, 200.0), "graph256_pe": (graph256_leg, 260.0)}      # name -> (function, seconds it needs on a 16-core box incl. its index build)
ONLY_LEGS = {"graph_big_pe": graph_big_leg}                                               # legs that only ever run under --only-legs (too long for the default run)


def spliced_leg(a, api, synth, local, cache):
    """BASELINE configs[4]'s shape (genome_snp_tran index, paired, the reference's default mode) at E. coli size: a seeded genome with an
    intron planted about every 1.3 kb, an index built with --snp --ss --exon (a third of the introns as index splice sites), pairs drawn
    from the spliced transcript.  Two figures: go() with spliced alignment on the device for one resident batch (--no-temp-splicesite:
    the kernel's own rate and roofline line), and the command line in the default temporary-splice-site mode (waves of 1000 x -p reads,
    each wave's junctions merged before the next) next to the reference at the same -p, with the SAM bodies compared."""
    import hashlib
    import numpy as np
    exe = os.path.join(REF, "hisat2-align-s")
    builder = os.path.join(REF, "hisat2-build-s")
    glen, rdlen, npairs = 4_900_000, 101, 1_000_000
    rng = np.random.default_rng(SEED + 99)
    g = rng.integers(0, 4, size=glen, dtype=np.uint8)
    introns, pos = [], 2000
    while pos < glen - 12000:
        L = int(rng.choice([60, 90, 150, 400, 1200, 5000, 9000]))
        a0, b0 = pos, pos + L
        kind = int(rng.integers(0, 10))
        if kind < 8:
            g[a0:a0 + 2] = [2, 3]; g[b0 - 2:b0] = [0, 2]          # GT..AG
        elif kind == 8:
            g[a0:a0 + 2] = [2, 1]; g[b0 - 2:b0] = [0, 2]          # GC..AG
        introns.append((a0, b0))
        pos = b0 + int(rng.integers(120, 700))
    keep = np.ones(glen, dtype=bool)
    for a0, b0 in introns:
        keep[a0:b0] = False
    tx = g[keep]
    fl = np.maximum(rdlen, rng.normal(280, 40, size=npairs).astype(np.int64))
    s0 = rng.integers(0, len(tx) - 1000, size=npairs)
    ar = np.arange(rdlen)
    left = tx[s0[:, None] + ar]
    right = (3 - tx[(s0 + fl)[:, None] - 1 - ar])               # reverse complement of the fragment's other end
    sub = rng.random((npairs, rdlen)) < 0.005
    left = np.where(sub, (left + rng.integers(1, 4, size=left.shape)) & 3, left).astype(np.uint8)
    sub = rng.random((npairs, rdlen)) < 0.005
    right = np.where(sub, (right + rng.integers(1, 4, size=right.shape)) & 3, right).astype(np.uint8)
    flip = rng.random(npairs) < 0.5
    m1 = np.where(flip[:, None], right, left)
    m2 = np.where(flip[:, None], left, right)
    d = os.path.join(cache, f"spl{glen}_s{SEED}")
    base = os.path.join(d, "g")
    if not os.path.exists(base + ".8.ht2"):
        os.makedirs(d, exist_ok=True)
        synth.write_fasta(base + ".fa", [g], names=["chr1"])
        with open(base + ".ss", "w") as f:
            for a0, b0 in introns[::3]:
                f.write("chr1\t%d\t%d\t+\n" % (a0 - 1, b0))
        with open(base + ".exon", "w") as f:
            prev = 0
            for a0, b0 in introns:
                f.write("chr1\t%d\t%d\n" % (max(prev, a0 - 400), a0 - 1)); prev = b0
        synth.write_snps(base + ".snp", synth.make_snps([g], SEED + 5, every=400, names=["chr1"]))
        subprocess.run([builder, "-q", "-p", str(_builder_threads()), "--snp", base + ".snp", "--ss", base + ".ss", "--exon", base + ".exon", base + ".fa", base],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    leg = {"workload": f"configs[4] shape at E. coli size: --snp --ss --exon index over a {glen} bp genome with {len(introns)} planted introns, {npairs} x 2 x {rdlen} bp pairs from the spliced transcript",
           "pairs": npairs, "introns": len(introns)}
    # (1) the device alone: one resident batch, spliced alignment, no temporary sites
    ix = api.Index(base, device=local)
    c1, o1 = synth.flatten_reads(m1)
    c2, o2 = synth.flatten_reads(m2)
    q = [str(i) for i in range(npairs)]
    st = api.Stream(ix, max_reads=npairs, max_bases=c1.size)
    st.set_reads(c1, o1); st.set_read_names(q); st.set_mates(c2, o2, q)
    p = st.align_params()
    p.no_spliced_alignment = 0; p.no_temp_splicesite = 1
    st.align_pairs_run(p); st.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        st.align_pairs_run(p)
    st.sync()
    dt = (time.perf_counter() - t0) / 3
    c = st.counters()
    alg = (int(c.n_rank) + int(c.n_sa_steps)) * 128                           # graph sides are 128 B lines, one per rank query / walk step (the graph units count ranks)
    leg["device_no_temp_splicesite"] = {"ms_per_step": dt * 1e3, "reads_per_s": 2 * npairs / dt, "kernel_ms": float(c.ms_align_kernel), "pairs_with_concordant": int(c.n_aligned),
                                        "second_pass": int(c.n_second_pass), "still_flagged": int(c.n_overflow), "ranks_per_pair": int(c.n_rank) / npairs,
                                        "sa_steps_per_pair": int(c.n_sa_steps) / npairs,
                                        "roofline": {"bound": "hbm", "kernel": "k_go<true> spliced unit (h2g_go_kernels.h)", "achieved": alg / (float(c.ms_align_kernel) * 1e-3) / 1e9,
                                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (float(c.ms_align_kernel) * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None}}
    st.close(); ix.close()
    # (2) the command line in the reference's default mode against the reference at the same -p
    tmp = tempfile.mkdtemp(prefix="h2benchspl")
    ncli = npairs
    f1, f2 = os.path.join(tmp, "r1.fa"), os.path.join(tmp, "r2.fa")
    synth.write_reads_fasta(f1, m1[:ncli]); synth.write_reads_fasta(f2, m2[:ncli])
    P = 16
    cli = os.path.join(ROOT, "hisat2_amd", "hisat2-align-amd")
    t0 = time.perf_counter()
    subprocess.run([cli, "-f", "-p", str(P), "-x", base, "-1", f1, "-2", f2, "-S", os.path.join(tmp, "amd.sam")], check=True, stderr=subprocess.DEVNULL)
    t_amd = time.perf_counter() - t0
    leg["default_mode_command_line"] = {"pairs": ncli, "p": P, "window_reads": 1000 * P, "wall_s": t_amd, "reads_per_s_wall": 2 * ncli / t_amd}
    t0 = time.perf_counter()                                                  # the window is the reference's -p, not this program's: 64 000 reads == hisat2 -p 64
    subprocess.run([cli, "-f", "-p", str(P), "--ss-window", "64000", "-x", base, "-1", f1, "-2", f2, "-S", os.path.join(tmp, "amd64.sam")], check=True, stderr=subprocess.DEVNULL)
    t_amd64 = time.perf_counter() - t0
    leg["default_mode_command_line_window_64000"] = {"pairs": ncli, "wall_s": t_amd64, "reads_per_s_wall": 2 * ncli / t_amd64}
    if not a.no_cpu_baseline:
        t0 = time.perf_counter()
        subprocess.run([exe, "-f", "-p", str(P), "--reorder", "-x", base, "-1", f1, "-2", f2, "-S", os.path.join(tmp, "ref.sam")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t_ref = time.perf_counter() - t0
        same = hashlib.md5("".join(body(os.path.join(tmp, "amd.sam"))).encode()).hexdigest() == hashlib.md5("".join(body(os.path.join(tmp, "ref.sam"))).encode()).hexdigest()
        leg["default_mode_command_line"].update({"reference_wall_s": t_ref, "reference_reads_per_s_wall": 2 * ncli / t_ref,
                                                 "sam_identical_to": f"hisat2-align-s -p {P} --reorder" if same else None})
        if not same:
            raise SystemExit("bench.py: the spliced leg's SAM differs from the reference's")
    shutil.rmtree(tmp, ignore_errors=True)
    return leg


if __name__ == "__main__":
    main()
