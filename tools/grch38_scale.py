#!/usr/bin/env python3
"""One-off GRCh38-SCALE run (BASELINE configs[2] shape): a seeded uniform-random 3.1 Gbp genome in 24 contigs, index built
on the GPU box by oracle/_ref/hisat2-build-s, then 1 M SE reads and 500 k pairs through h2g_align_run /
h2g_align_pairs_run, a sample checked line-by-line against oracle/_ref/hisat2-align-s, and the reference timed on
the host cores over a bounded sample.  Writes gpurun_out/grch38_scale.json.  (No real GRCh38: there is no network.)"""
import json, os, subprocess, sys, time, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hisat2_amd import api, synth

SEED = 20260925 + 38
TOTAL = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
BUILD_TIMEOUT = int(sys.argv[2]) if len(sys.argv) > 2 else 1800
out = {"genome_len": TOTAL}
def log(*a):
    print(*a, flush=True)
def save():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "grch38_scale.json"), "w"), indent=1)

mem = {l.split(":")[0]: int(l.split()[1]) for l in open("/proc/meminfo") if l.split()[1].isdigit()}
out["host"] = {"cores": os.cpu_count(), "mem_total_gb": mem.get("MemTotal", 0) / 1e6, "mem_avail_gb": mem.get("MemAvailable", 0) / 1e6}
log(out["host"])
if mem.get("MemAvailable", 0) / 1e6 < 40:
    raise SystemExit("not enough host memory for a 3.1 Gbp build")
# human-like contig size profile (24 "chromosomes"), scaled to TOTAL
prof = np.array([248, 242, 198, 190, 181, 171, 159, 145, 138, 133, 135, 133, 114, 107, 102, 90, 83, 80, 58, 64, 46, 50, 156, 57], dtype=np.float64)
lens = (prof / prof.sum() * TOTAL).astype(np.int64)
t0 = time.time()
contigs = synth.make_genome([int(x) for x in lens], SEED)
work = "/tmp/h2g_grch38"
os.makedirs(work, exist_ok=True)
base = os.path.join(work, "rnd38")
fa = base + ".fa"
synth.write_fasta(fa, contigs)
out["t_genome_s"] = time.time() - t0
log("genome written in %.1f s" % out["t_genome_s"])
builder = os.path.join(ROOT, "oracle", "_ref", "hisat2-build-s")
nthr = min(os.cpu_count() or 1, 64)
t0 = time.time()
try:
    subprocess.run([builder, "-q", "-p", str(nthr), fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=BUILD_TIMEOUT)
except subprocess.TimeoutExpired:
    out["error"] = f"index build exceeded {BUILD_TIMEOUT} s"; save(); raise SystemExit(out["error"])
except subprocess.CalledProcessError as e:
    out["error"] = "index build failed: " + e.stderr.decode()[-400:]; save(); raise SystemExit(out["error"])
out["t_index_build_s"] = time.time() - t0
out["index_bytes"] = {k: os.path.getsize(f"{base}.{k}.ht2") for k in range(1, 9)}
os.remove(fa)
log("index built in %.1f s" % out["t_index_build_s"], out["index_bytes"])
save()

t0 = time.time()
ix = api.Index(base)
out["t_index_load_s"] = time.time() - t0
out["index_info"] = {k: int(getattr(ix.info, k)) for k in ("len", "gbwtLen", "numSides", "nLocal", "device_bytes", "minK", "nFrag")}
log("index resident:", out["index_info"], "load %.1f s" % out["t_index_load_s"])
nreads = 1_000_000
reads, _ = synth.make_reads(contigs, nreads, 101, SEED + 1, sub_rate=0.005)
codes, offs = synth.flatten_reads(reads)
st = api.Stream(ix, max_reads=nreads, max_bases=codes.size)
qn = [str(i) for i in range(nreads)]
st.set_reads(codes, offs); st.set_read_names(qn)
st.align_run(); st.sync()
t0 = time.perf_counter()
for _ in range(3):
    st.align_run()
st.sync()
dt = (time.perf_counter() - t0) / 3
c = st.counters()
out["se"] = {"reads": nreads, "ms_per_step": dt * 1e3, "reads_per_s": nreads / dt, "kernel_ms": float(c.ms_align), "aligned": int(c.n_aligned),
             "overflow": int(c.n_overflow), "ranks_per_read": c.n_rank / nreads, "sa_steps_per_read": c.n_sa_steps / nreads,
             "sides_per_read": c.n_side / nreads,
             "algorithmic_GBs": (int(c.n_side) + int(c.n_sa_steps)) * 64 / (float(c.ms_align) * 1e-3) / 1e9}
log("SE", out["se"])
p = st.seed_params(True)
st.seed_extend_run(p); st.sync(); st.seed_extend_run(p); st.sync()
c2 = st.counters()
out["seed_stage"] = {"ms_search": float(c2.ms_search), "ms_resolve_extend": float(c2.ms_resolve_extend),
                     "search_algorithmic_GBs": int(c2.n_side) * 64 / (float(c2.ms_search) * 1e-3) / 1e9 if c2.ms_search > 0 else None,
                     "sides": int(c2.n_side), "sa_steps": int(c2.n_sa_steps)}
log("seed stage", out["seed_stage"])
save()
# parity sample, SE
import sam_util as SU
exe = os.path.join(ROOT, "oracle", "_ref", "hisat2-align-s")
tmp = tempfile.mkdtemp(prefix="h2g38")
nv = 3000
synth.write_reads_fasta(os.path.join(tmp, "se.fa"), reads[:nv])
t0 = time.time()
subprocess.run([exe, "-f", "-p", "1", "--no-spliced-alignment", "-x", base, "-U", os.path.join(tmp, "se.fa"), "-S", os.path.join(tmp, "se.sam")],
               check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
out["t_ref_se_sample_s"] = time.time() - t0
refnames, want = SU.parse_sam(os.path.join(tmp, "se.sam"))
res, aln = st.align_fetch(0, nv)
got = SU.render_selected(res, aln, refnames, [reads[i] for i in range(nv)], qn[:nv])
nbad = sum(1 for q in qn[:nv] if got[q] != want[q])
out["se"]["sam_checked_reads"] = nv; out["se"]["sam_mismatching_reads"] = nbad
log("SE parity: %d of %d differ" % (nbad, nv))
save()
# CPU reference throughput at this scale (bounded sample, -p scan)
ns = 200_000
synth.write_reads_fasta(os.path.join(tmp, "cpu.fa"), reads[:ns])
cpu = {}
for thr in (16, 64):
    if thr > (os.cpu_count() or 1):
        continue
    cmd = [exe, "-f", "-p", str(thr), "--no-spliced-alignment", "-x", base, "-U", os.path.join(tmp, "cpu.fa"), "-S", "/dev/null"]
    t0 = time.perf_counter()
    subprocess.run(cmd + ["-u", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    t_load = time.perf_counter() - t0
    t0 = time.perf_counter()
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dtc = max(time.perf_counter() - t0 - t_load, 1e-6)
    cpu[thr] = {"reads": ns, "seconds": dtc, "index_load_seconds": t_load, "reads_per_s": ns / dtc}
out["cpu_reference"] = cpu
log("cpu reference", cpu)
save()
st.close()
# PE
npairs = 500_000
m1, m2 = synth.make_pairs(contigs, npairs, 101, SEED + 7, sub_rate=0.005)
c1, o1 = synth.flatten_reads(m1); c2_, o2 = synth.flatten_reads(m2)
qp = [str(i) for i in range(npairs)]
pst = api.Stream(ix, max_reads=npairs, max_bases=c1.size)
pst.set_reads(c1, o1); pst.set_read_names(qp); pst.set_mates(c2_, o2, qp)
pst.align_pairs_run(); pst.sync()
t0 = time.perf_counter()
for _ in range(3):
    pst.align_pairs_run()
pst.sync()
pdt = (time.perf_counter() - t0) / 3
pc = pst.counters()
out["pe"] = {"pairs": npairs, "ms_per_step": pdt * 1e3, "pairs_per_s": npairs / pdt, "kernel_ms": float(pc.ms_align),
             "concordant": int(pc.n_aligned), "overflow": int(pc.n_overflow)}
log("PE", out["pe"])
save()
import fuzz_pairs as FP
import pe_sink as PS
nvp = 1500
synth.write_reads_fasta(os.path.join(tmp, "1.fa"), m1[:nvp]); synth.write_reads_fasta(os.path.join(tmp, "2.fa"), m2[:nvp])
subprocess.run([exe, "-f", "-p", "1", "--no-spliced-alignment", "-x", base, "-1", os.path.join(tmp, "1.fa"), "-2", os.path.join(tmp, "2.fa"),
                "-S", os.path.join(tmp, "pe.sam")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
rn, wantp = FP.parse_pe_sam(os.path.join(tmp, "pe.sam"))
pres, pa1, pa2 = pst.align_pairs_fetch(0, nvp)
nbad = sum(1 for i in range(nvp) if PS.finish_pair(pres[i], pa1, pa2, i * api.PAIR_RES_CAP, rn, (m1[i], m2[i])) != wantp[str(i)])
out["pe"]["sam_checked_pairs"] = nvp; out["pe"]["sam_mismatching_pairs"] = nbad
log("PE parity: %d of %d differ" % (nbad, nvp))
save()
shutil.rmtree(tmp, ignore_errors=True)
shutil.rmtree(work, ignore_errors=True)
log("done")
