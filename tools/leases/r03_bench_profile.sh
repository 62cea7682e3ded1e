#!/bin/bash
# One GPU-box visit at the metric's size (GRCh38-size index built on the box): the default bench.py line, then rocprofv3 kernel trace
# and the FETCH_SIZE / WRITE_SIZE passes of the headline leg (same command each time), summaries and the per-launch traffic of the
# dominant kernel as profiles/r03_pmc_traffic.json expects it.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
STEPS=${STEPS:-20}
python bench.py --steps $STEPS --warmup 5 > $OUT/r03_bench.json 2> $OUT/r03_bench.err        # builds / caches the index
tail -c 600 $OUT/r03_bench.err
CMD="python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 1"
rocprofv3 --kernel-trace --stats -d /tmp/bp_trace -- $CMD > $OUT/r03_bench_traced.json 2> /tmp/bp_trace.err
python tools/rocpd_summary.py /tmp/bp_trace > $OUT/r03_bench_trace.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/bp_pmc
  rocprofv3 --pmc $c -d /tmp/bp_pmc -- $CMD > /dev/null 2> /tmp/bp_pmc.err
  echo "# rocprofv3 --pmc $c -- $CMD" > $OUT/r03_bench_pmc_$c.txt
  python tools/rocpd_summary.py /tmp/bp_pmc >> $OUT/r03_bench_pmc_$c.txt 2>&1
done
python - <<'PY'
import json
def mean(path, counter, kern):
    for l in open(path):
        if l.startswith(kern) and counter in l:
            return float(l.split()[-1])
out = {}
b = json.loads(open("gpurun_out/r03_bench.json").read().strip().splitlines()[-1])
for kern in ("k_go_fast", "void k_go<false, 2, 0>"):
    f = mean("gpurun_out/r03_bench_pmc_FETCH_SIZE.txt", "FETCH_SIZE", kern); w = mean("gpurun_out/r03_bench_pmc_WRITE_SIZE.txt", "WRITE_SIZE", kern)
    if f is not None and w is not None:
        out[kern] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "traffic_bytes_per_launch": int(2 * f * 1024 + w * 1024)}
k = out.get("k_go_fast", {})
json.dump({"pairs_per_launch": b["config"]["pairs_per_gpu"], "genome": b["config"]["genome_bases"], "kernel": "k_go_fast",
           "traffic_bytes_per_launch": k.get("traffic_bytes_per_launch"), "FETCH_SIZE_KB": k.get("FETCH_SIZE_KB"), "WRITE_SIZE_KB": k.get("WRITE_SIZE_KB"),
           "all_kernels": out,
           "source": "profiles/r03_bench_pmc_FETCH_SIZE.txt + r03_bench_pmc_WRITE_SIZE.txt (rocprofv3 --pmc, separate passes of the same bench.py command, mean per launch of k_go_fast): 2 x FETCH_SIZE (gfx950 correction for wide loads; upper bound for 64 B lines) + WRITE_SIZE"},
          open("gpurun_out/r03_pmc_traffic.json", "w"))
print(open("gpurun_out/r03_pmc_traffic.json").read())
PY
head -c 3000 $OUT/r03_bench.json
