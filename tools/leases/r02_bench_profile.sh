#!/bin/bash
# One GPU-box visit: rocprofv3 kernel trace and the FETCH_SIZE / WRITE_SIZE passes of bench.py's headline leg (same command each time),
# then profiles-ready summaries and the per-launch traffic of the go() kernel as profiles/r02_pmc_traffic.json expects it.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
CMD="python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 1"
$CMD > $OUT/r02_bench_plain.json 2> $OUT/r02_bench_plain.err            # builds / caches the index
rocprofv3 --kernel-trace --stats -d /tmp/bp_trace -- $CMD > $OUT/r02_bench_traced.json 2> /tmp/bp_trace.err
python tools/rocpd_summary.py /tmp/bp_trace > $OUT/r02_bench_trace.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/bp_pmc
  rocprofv3 --pmc $c -d /tmp/bp_pmc -- $CMD > /dev/null 2> /tmp/bp_pmc.err
  echo "# rocprofv3 --pmc $c -- $CMD" > $OUT/r02_bench_pmc_$c.txt
  python tools/rocpd_summary.py /tmp/bp_pmc >> $OUT/r02_bench_pmc_$c.txt 2>&1
done
python - <<'PY'
import json, re
def mean(path, counter):
    for l in open(path):
        if "k_go<false" in l and counter in l:
            return float(l.split()[-1])
f = mean("gpurun_out/r02_bench_pmc_FETCH_SIZE.txt", "FETCH_SIZE"); w = mean("gpurun_out/r02_bench_pmc_WRITE_SIZE.txt", "WRITE_SIZE")
b = json.loads(open("gpurun_out/r02_bench_plain.json").read().strip().splitlines()[-1])
t = int(2 * f * 1024 + w * 1024)
json.dump({"pairs_per_launch": b["config"]["pairs_per_gpu"], "genome": b["config"]["genome_bases"], "traffic_bytes_per_launch": t,
           "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
           "source": "profiles/r02_bench_pmc_FETCH_SIZE.txt + r02_bench_pmc_WRITE_SIZE.txt (rocprofv3 --pmc, separate passes of the same bench.py command, mean per launch of k_go<false,...>): 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE"},
          open("gpurun_out/r02_pmc_traffic.json", "w"))
print("traffic per launch", t)
PY
