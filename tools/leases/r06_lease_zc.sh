#!/bin/bash
# Round 6, lease ZC: formatting on its own thread or on the main thread (H2G_CLI_ASYNC_FMT=0), three runs each, 4 M pairs on a 1 Gbp index, to a file — one box (16 CPUs by its cgroup)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_zc; mkdir -p $OUT
T0=$(date +%s)
python - <<'PY' > $OUT/ab.log 2>&1
import os, sys, time, subprocess, tempfile
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
from hisat2_amd import synth
import build_bench_index as BB
base, total, how = bench.headline_index(os.path.join(ROOT, ".bench_cache"), int(1e9))
contigs = BB.genome(total)
n = 4000000
m1, m2 = synth.make_pairs(contigs, n, 101, bench.SEED + 7, sub_rate=0.005)
tmp = tempfile.mkdtemp(prefix="h2cli")
f1, f2 = os.path.join(tmp, "r1.fa"), os.path.join(tmp, "r2.fa")
synth.write_reads_fasta(f1, m1); synth.write_reads_fasta(f2, m2)
exe = os.path.join(ROOT, "hisat2_amd", "hisat2-align-amd")
for rep in range(3):
    for mode in ("1", "0"):
        for dest in (os.path.join(tmp, "o.sam"), "/dev/null"):
            t0 = time.perf_counter()
            r = subprocess.run([exe, "-f", "--no-spliced-alignment", "-p", "32", "-x", base, "-1", f1, "-2", f2, "-S", dest], env=dict(os.environ, H2G_CLI_TIMING="1", H2G_CLI_ASYNC_FMT=mode), capture_output=True, text=True)
            dt = time.perf_counter() - t0
            print("async %s -> %-9s rc %d wall %.2f s = %.2f M reads/s | %s" % (mode, "file" if dest != "/dev/null" else dest, r.returncode, dt, 2 * n / dt / 1e6, [l for l in r.stderr.splitlines() if l.startswith("time:")]), flush=True)
PY
cut -c1-330 $OUT/ab.log | tail -14
echo "done after $(( $(date +%s) - T0 )) s"
