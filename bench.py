#!/usr/bin/env python3
"""bench.py — throughput of the MI355X seed-and-extend hot path (BASELINE.json metric: reads/sec; Occ-rank HBM GB/s).

One "step" = one pass of the hot path over one resident batch of synthetic reads = HI_Aligner::go for every read
(hi_aligner.h:4048: FM backward search on both strands, SA-offset resolution, ungapped extension, local-index
search, indel joins, recursion, sink feedback, selectByScore) — h2g_align_run.  The fused seed stage of round-1a
(partialSearch -> getGenomeCoords -> extend(mm=0)) is timed once more and reported under "seed_stage".
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
        nthr = min(os.cpu_count() or 1, 64)
        subprocess.run([builder, "-q", "-p", str(nthr), fa, base + ".tmp"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
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


def cpu_reference(base, reads, sample, nver):
    """Reference leg: the REAL reference aligner (oracle/_ref/hisat2-align-s, built from /root/reference by
    oracle/Makefile.ref) on the host cores of this box, same reads, same flags; wall time minus a no-read run
    (index load).  Also returns the SAM of the first `nver` reads for the parity check."""
    import sam_util as SU
    from hisat2_amd import synth
    exe = os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")
    if not os.path.exists(exe):
        return None, None
    import tempfile
    tmp = tempfile.mkdtemp(prefix="h2bench")
    fa = os.path.join(tmp, "sample.fa")
    synth.write_reads_fasta(fa, reads[:sample])
    ncpu = os.cpu_count() or 1
    # the reference's reader/writer locks stop scaling long before 256 threads (127 k reads/s at -p 8 on
    # a cache-resident index), so scan a few thread counts on the bounded sample and report the best one
    scan = {}
    best = None
    for pth in [t for t in (1, 4, 16, 64) if t <= ncpu]:
        cmd = [exe, "-f", "--no-spliced-alignment", "-p", str(pth), "-x", base, "-U", fa, "-S", "/dev/null"]
        t0 = time.perf_counter()
        subprocess.run(cmd + ["-u", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t_load = time.perf_counter() - t0
        t0 = time.perf_counter()
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = max(time.perf_counter() - t0 - t_load, 1e-6)
        scan[str(pth)] = sample / dt
        if best is None or sample / dt > best[0]:
            best = (sample / dt, pth, dt, t_load)
    cores, dt, t_load = best[1], best[2], best[3]
    sam = os.path.join(tmp, "ver.sam")
    subprocess.run([exe, "-f", "--no-spliced-alignment", "-p", "1", "-x", base, "-U", fa, "-u", str(nver), "-S", sam],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    refnames, want = SU.parse_sam(sam)
    # the whole drop-in path on the same sample: reads file -> hisat2-align-amd (parse, GPU go(), C++ sink + SAM text) -> SAM
    # file, diffed line by line against the reference's own SAM of the sample
    cli = os.path.join(ROOT, "hisat2_amd", "hisat2-align-amd")
    cli_leg = None
    if os.path.exists(cli):
        full = os.path.join(tmp, "full.sam")
        subprocess.run([exe, "-f", "--no-spliced-alignment", "-p", str(cores), "--reorder", "-x", base, "-U", fa, "-S", full], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        mine = os.path.join(tmp, "amd.sam")
        t0 = time.perf_counter()
        r = subprocess.run([cli, "-f", "--no-spliced-alignment", "-p", "16", "-x", base, "-U", fa, "-S", mine], capture_output=True, text=True,
                           env=dict(os.environ, H2G_CLI_TIMING="1"))
        t_cli = time.perf_counter() - t0
        if r.returncode == 0:
            a = [l for l in open(mine) if not l.startswith("@")]
            b = [l for l in open(full) if not l.startswith("@")]
            ndiff = sum(1 for x, y in zip(a, b) if x != y) + abs(len(a) - len(b))
            cli_leg = {"reads": sample, "wall_s": t_cli, "reads_per_s_wall": sample / t_cli, "timing": r.stderr.strip().splitlines()[-1],
                       "sam_lines": len(b), "sam_lines_differing": ndiff,
                       "host_threads": 16, "note": "wall time of the process incl. index load + upload, FASTA parsing, SAM formatting and file write"}
            if ndiff:
                raise SystemExit(f"bench.py: hisat2-align-amd SAM differs from the reference on {ndiff} lines")
        else:
            cli_leg = {"error": r.stderr[-400:]}
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return ({"value": sample / dt, "cli_end_to_end": cli_leg, "unit": "reads/s", "cores": cores, "kind": "reference", "host_cpus": ncpu, "threads_scan_reads_per_s": scan,
             "sample": f"first {sample} reads of the bench batch, oracle/_ref/hisat2-align-s -p {cores} --no-spliced-alignment -S /dev/null, {dt:.2f} s (index load {t_load:.2f} s subtracted); best of the thread counts scanned"},
            (refnames, want))


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
    ap.add_argument("--cpu-sample", type=int, default=300_000)
    ap.add_argument("--rank-queries", type=int, default=1 << 26)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", type=int, default=5000)
    ap.add_argument("--pairs", type=int, default=500_000)
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
    reads, truth = synth.make_reads(contigs, a.reads, 101, SEED + 1000 * (rank + 1), sub_rate=0.005)
    codes, offs = synth.flatten_reads(reads)
    ix = api.Index(base, device=local)
    st = api.Stream(ix, max_reads=a.reads, max_bases=codes.size)
    st.set_reads(codes, offs)                 # inputs resident in HBM before the timed region
    st.set_read_names([str(i) for i in range(a.reads)])   # FASTA names = decimal ids (feed genRandSeed, pat.h:55)
    params = st.seed_params(no_spliced=True)  # config 2 runs --no-spliced-alignment
    aparams = st.align_params()

    def barrier():
        st.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    st.seed_extend_run(params)                 # round-1a seed stage, timed by HIP events only (reported aside)
    st.sync()
    for _ in range(a.warmup):
        st.align_run(aparams)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st.align_run(aparams)
    barrier()
    dt = time.perf_counter() - t0
    cnt = st.counters()                        # counters + HIP-event kernel times of the LAST step
    ares, aaln = st.align_fetch(0, min(a.verify, a.reads))
    allres, _ = st.align_fetch(with_alignments=False)
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
    asum = np.array([int((allres["nselect"] > 0).sum()), int((allres["nselect"] > 1).sum()), int((allres["overflow"] != 0).sum()),
                     int(allres["nrank"].sum()), int(allres["nsteps"].sum()), int(cnt.n_side)], dtype=np.int64)
    asum = shard.all_reduce_sum(asum, dist, device="cuda")   # the alignment-summary reduction (RCCL over xGMI)

    if rank == 0:
        total_reads = a.reads * world
        value = total_reads * a.steps / dt
        # dominant kernel = k_align (the whole go() state machine).  Algorithmic bytes =
        # 64 B x (unique sides visited by the search loops + SA-walk steps) of the LAST launch on this rank.
        ms_align = float(cnt.ms_align)
        alg_bytes = int(cnt.n_side) * 64 + int(cnt.n_sa_steps) * 64
        achieved = alg_bytes / (ms_align * 1e-3) / 1e9 if ms_align > 0 else 0.0
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "k_align", "kernel_ms": ms_align,
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "note": "lane-per-read state machine, latency/divergence-bound at this index size (1.2 MB of sides is L2-resident); the HBM-scale Occ-rank number is rank_microbench"}
        # HBM traffic of the same kernel on the same workload from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
        # need separate passes; they cannot be collected inside this process).  FETCH_SIZE is in KB and counts 128 B requests
        # at 64 B on gfx950 => x2 (MI355X guide, HBM section); WRITE_SIZE is taken as reported (uncalibrated).
        try:
            def _pmc(fn, ctr):
                for ln in open(os.path.join(ROOT, "profiles", fn)):
                    if "k_align<" in ln and ", false>" in ln and "k_align<3," not in ln and ctr in ln:
                        return float(ln.split()[-1])
                return None
            fk, wk = _pmc("r01_m_pmc_fetch.txt", "FETCH_SIZE"), _pmc("r01_m_pmc_write.txt", "WRITE_SIZE")
            if fk is not None and wk is not None and a.reads == 1_000_000:
                roofline["traffic"] = int(fk * 1024 * 2 + wk * 1024)
                roofline["traffic_source"] = ("profiles/r01_m_pmc_fetch.txt + r01_m_pmc_write.txt (rocprofv3 --pmc, same workload, per launch): "
                                              "2 x FETCH_SIZE + WRITE_SIZE; ~8x the algorithmic bytes = per-lane workspace traffic, DESIGN.md §3")
        except OSError:
            pass
        seed_stage = {"ms_search": float(cnt.ms_search), "ms_resolve_extend": float(cnt.ms_resolve_extend),
                      "reads_per_s": a.reads / ((float(cnt.ms_search) + float(cnt.ms_resolve_extend)) * 1e-3)}
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
        # same micro-kernel on GRAPH sides (128 B = one L2 line): 7.65 M synthetic sides = the same 0.98 GB
        gix = api.Index(synth_sides=7_650_000, seed=SEED, device=local, graph=True)
        gst = api.Stream(gix)
        micro_g = {}
        for v, name in ((0, "lane_per_side"), (1, "8_lanes_per_side")):
            gst.rank_synth(a.rank_queries, SEED, variant=v, repeats=1)
            ms, ck = gst.rank_synth(a.rank_queries, SEED, variant=v, repeats=3)
            gbs = a.rank_queries * 128 / (ms * 1e-3) / 1e9
            micro_g[name] = {"ms": ms, "GB/s": gbs, "frac_of_8TBs": gbs / HBM_PEAK_GBS, "checksum": int(ck)}
        gst.close()
        gix.close()
        # Smith-Waterman kernels (a23-a25, opt-in path of the reference): 101 x 141 cells x {H,E,F} per problem; problems = the first 65536 bench reads framed around their true position, as hybridSearch frames a seed hit
        nsw = min(65536, a.reads)
        swq = [api.SwQuery(i, int(truth[i][2]), int(truth[i][0]), int(truth[i][1]), -20, i + 1) for i in range(nsw)]
        st.sw_align(swq[:1024])
        swres, sw_ms = st.sw_align(swq, repeats=3)
        sw_found = sum(1 for r in swres if r.found)
        sw_cells = sum(101 * int(r.refr - r.refl + 1) for r in swres)
        sw_micro = {"problems": nsw, "kernel_ms": sw_ms, "problems_per_s": nsw / (sw_ms * 1e-3), "GCUPS": sw_cells / (sw_ms * 1e-3) / 1e9,
                    "found": sw_found, "cells_per_problem": sw_cells / nsw,
                    "note": "k_sw_fill: register-systolic wavefront per problem (one __shfl_up per anti-diagonal step), H/E/F streamed to HBM anti-diagonal-major (92.5 KB/problem, coalesced 64 B stores); k_sw_backtrace: one lane per problem"}
        # go() on a SNP-GRAPH index (BASELINE configs[3] shape at config-2 size): the same genome with a seeded variant every
        # ~250 bp (hisat2-build-s --snp), reads drawn from the alternate haplotype; 2000 reads checked against the reference
        graph_leg = None
        builder = os.path.join(ROOT, "oracle", "_ref", "hisat2-build-s")
        if os.path.exists(builder):
            import tempfile, shutil
            gtmp = os.path.join(cache, f"rnd{a.genome_len}_s{SEED}_snp")
            gbase = os.path.join(gtmp, "g")
            var = synth.make_snps(contigs, SEED + 5, every=250, names=["ecoli_substitute"])
            if not os.path.exists(gbase + ".8.ht2"):
                os.makedirs(gtmp, exist_ok=True)
                synth.write_fasta(gbase + ".fa", contigs, names=["ecoli_substitute"])
                synth.write_snps(gbase + ".snp", var)
                subprocess.run([builder, "-q", "-p", str(min(os.cpu_count() or 1, 64)), "--snp", gbase + ".snp", gbase + ".fa", gbase], check=True,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            alt = synth.apply_snps(contigs, var, names=["ecoli_substitute"])
            greads, _ = synth.make_reads(alt, a.reads, 101, SEED + 4242, sub_rate=0.005)
            gcodes, goffs = synth.flatten_reads(greads)
            gix = api.Index(gbase, device=local)
            gst = api.Stream(gix, max_reads=a.reads, max_bases=gcodes.size)
            gst.set_reads(gcodes, goffs); gst.set_read_names([str(i) for i in range(a.reads)])
            gst.align_run(); gst.sync()
            t0g = time.perf_counter()
            for _ in range(3):
                gst.align_run()
            gst.sync()
            gdt = (time.perf_counter() - t0g) / 3
            gc_ = gst.counters()
            graph_leg = {"reads": a.reads, "variants": len(var), "ms_per_step": gdt * 1e3, "reads_per_s": a.reads / gdt, "kernel_ms": float(gc_.ms_align),
                         "aligned": int(gc_.n_aligned), "overflow": int(gc_.n_overflow), "numSides": int(gix.info.numSides), "sideSz": int(gix.info.sideSz)}
            # paired-end on the graph index (the shape of BASELINE configs[3]): half as many pairs
            gnp = a.reads // 2
            gm1, gm2 = synth.make_pairs(alt, gnp, 101, SEED + 4343, frag_mean=300, frag_sd=30, sub_rate=0.005)
            gq = [str(i) for i in range(gnp)]
            gc1, go1 = synth.flatten_reads(gm1)
            gc2, go2 = synth.flatten_reads(gm2)
            gpst = api.Stream(gix, max_reads=gnp, max_bases=gc1.size)
            gpst.set_reads(gc1, go1); gpst.set_read_names(gq); gpst.set_mates(gc2, go2, gq)
            gpst.align_pairs_run(); gpst.sync()
            t0g = time.perf_counter()
            for _ in range(3):
                gpst.align_pairs_run()
            gpst.sync()
            gpdt = (time.perf_counter() - t0g) / 3
            gpc = gpst.counters()
            graph_leg["paired_end"] = {"pairs": gnp, "ms_per_step": gpdt * 1e3, "reads_per_s": 2 * gnp / gpdt, "kernel_ms": float(gpc.ms_align),
                                       "pairs_with_concordant": int(gpc.n_aligned), "pairs_overflow": int(gpc.n_overflow)}
            gpst.close()
            exe = os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")
            if os.path.exists(exe) and not a.no_cpu_baseline:
                import sam_util as SU
                nvg = 2000
                tmpg = tempfile.mkdtemp(prefix="h2benchg")
                synth.write_reads_fasta(os.path.join(tmpg, "r.fa"), greads[:nvg])
                subprocess.run([exe, "-f", "-p", "1", "--no-spliced-alignment", "-x", gbase, "-U", os.path.join(tmpg, "r.fa"), "-S", os.path.join(tmpg, "r.sam")],
                               check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                rn, wantg = SU.parse_sam(os.path.join(tmpg, "r.sam"))
                gres, galn = gst.align_fetch(0, nvg)
                qg = [str(i) for i in range(nvg)]
                gotg = SU.render_selected(gres, galn, rn, [greads[i] for i in range(nvg)], qg)
                nbadg = sum(1 for q in qg if gotg[q] != wantg[q])
                shutil.rmtree(tmpg, ignore_errors=True)
                graph_leg.update({"sam_checked_reads": nvg, "sam_mismatching_reads": nbadg})
                if nbadg:
                    raise SystemExit(f"bench.py: {nbadg} of {nvg} graph-index reads differ from the reference SAM")
            gst.close(); gix.close()
        nver = 2000
        verify_sample(base, reads, got, nver)          # seed stage vs oracle/h2o.c
        cpu_ref, ref_sam = (None, None)
        if not a.no_cpu_baseline:
            cpu_ref, ref_sam = cpu_reference(base, reads, min(a.cpu_sample, a.reads), len(ares))
        parity = {"seed_stage_vs_oracle_reads": nver, "bit_exact": True}
        if ref_sam is not None:
            import sam_util as SU
            qn = [str(i) for i in range(len(ares))]
            gotsam = SU.render_selected(ares, aaln, ref_sam[0], [reads[i] for i in range(len(ares))], qn)
            nbad = sum(1 for q in qn if gotsam[q] != ref_sam[1][q])
            parity.update({"sam_checked_reads": len(ares), "sam_mismatching_reads": nbad,
                           "against": "oracle/_ref/hisat2-align-s (FLAG, RNAME, POS, CIGAR, AS:i per line)"})
            if nbad:
                raise SystemExit(f"bench.py: {nbad} of {len(ares)} reads differ from the reference SAM")
        # paired-end extra (BASELINE north star is PE): 500 k pairs = the same 1 M reads, HI_Aligner::go with pairReads /
        # alignMate on the GPU; the first 2000 pairs are checked line-by-line against the reference's -1/-2 SAM
        pe = None
        if a.pairs > 0:
            import fuzz_pairs as FP
            import pe_sink as PS
            m1, m2 = synth.make_pairs(contigs, a.pairs, 101, SEED + 77, sub_rate=0.005)
            c1, o1 = synth.flatten_reads(m1)
            c2, o2 = synth.flatten_reads(m2)
            qn = [str(i) for i in range(a.pairs)]
            pst = api.Stream(ix, max_reads=a.pairs, max_bases=c1.size)
            pst.set_reads(c1, o1); pst.set_read_names(qn); pst.set_mates(c2, o2, qn)
            pst.align_pairs_run(); pst.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                pst.align_pairs_run()
            pst.sync()
            pdt = (time.perf_counter() - t0) / 3
            pc = pst.counters()
            pe = {"pairs": a.pairs, "ms_per_step": pdt * 1e3, "pairs_per_s": a.pairs / pdt, "reads_per_s": 2 * a.pairs / pdt,
                  "kernel_ms": float(pc.ms_align), "pairs_with_concordant": int(pc.n_aligned), "pairs_overflow": int(pc.n_overflow)}
            exe = os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")
            if os.path.exists(exe) and not a.no_cpu_baseline:
                import tempfile, shutil
                nv = min(2000, a.pairs)
                tmp = tempfile.mkdtemp(prefix="h2benchpe")
                synth.write_reads_fasta(os.path.join(tmp, "1.fa"), m1[:nv]); synth.write_reads_fasta(os.path.join(tmp, "2.fa"), m2[:nv])
                subprocess.run([exe, "-f", "-p", "1", "--no-spliced-alignment", "-x", base, "-1", os.path.join(tmp, "1.fa"), "-2",
                                os.path.join(tmp, "2.fa"), "-S", os.path.join(tmp, "pe.sam")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                rn, want = FP.parse_pe_sam(os.path.join(tmp, "pe.sam"))
                pres, pa1, pa2 = pst.align_pairs_fetch(0, nv)
                nbad = sum(1 for i in range(nv) if PS.finish_pair(pres[i], pa1, pa2, i * api.PAIR_RES_CAP, rn, (m1[i], m2[i])) != want[str(i)])
                shutil.rmtree(tmp, ignore_errors=True)
                pe.update({"sam_checked_pairs": nv, "sam_mismatching_pairs": nbad})
                if nbad:
                    raise SystemExit(f"bench.py: {nbad} of {nv} pairs differ from the reference SAM")
            pst.close()
        out = {
            "metric": "reads/sec, 101 bp SE (whole job): HI_Aligner::go per read on the GPU, bit-identical FLAG/POS/CIGAR/AS",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "configs[1]: E. coli-size linear GFM, 1M synthetic 101 bp SE reads per GPU, 1xMI355X",
                       "genome": f"seeded uniform-random {a.genome_len} bp substitute for NC_008253 (no network)",
                       "reads_per_gpu": a.reads, "read_len": 101, "sub_rate": 0.005, "mode": "--no-spliced-alignment -k 5",
                       "stage": "full HI_Aligner::go + selectByScore (alignments stay in HBM; SAM text formatting is the host's, SURVEY §8(f) N1)",
                       "sharding": f"reads by id range across {world} GPU(s), index replicated; RCCL all-reduce of summary counters only"},
            "roofline": roofline,
            "rank_microbench": {"sides": 15_300_000, "bytes": 15_300_000 * 64, "queries": a.rank_queries, **micro},
            "rank_microbench_graph": {"sides": 7_650_000, "bytes": 7_650_000 * 128, "queries": a.rank_queries, **micro_g},
            "sw_microbench": sw_micro,
            "graph_index": graph_leg,
            "seed_stage": seed_stage,
            "paired_end": pe,
            "counters": {"reads_aligned": int(asum[0]), "reads_multi": int(asum[1]), "reads_overflow": int(asum[2]),
                         "ranks_per_read": float(asum[3]) / total_reads, "sa_steps_per_read": float(asum[4]) / total_reads,
                         "sides_per_read_rank0": float(asum[5]) / a.reads, "seed_reads_with_anchor": int(summ[0])},
            "parity": parity,
        }
        if cpu_ref is not None:
            cli_leg = cpu_ref.pop("cli_end_to_end", None)
            if cli_leg is not None:
                out["cli_end_to_end"] = cli_leg
            out["cpu_baseline"] = cpu_ref
            out["cpu_baseline_port"] = cpu_baseline(base, reads, min(200000, a.reads))
        elif not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(base, reads, min(200000, a.reads))
        print(json.dumps(out))
    st.close()
    ix.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
