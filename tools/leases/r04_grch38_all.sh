#!/bin/bash
# ONE lease at the metric's size: build the 3.1 Gbp index once, then
#  (1) the fast kernel's scheduling choices (tail hand-off x alignMate in the pass) and its time split, where every index line is an HBM miss
#  (2) the headline line with the best of them, (3) rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of the same command,
#  (4) profiles/r04_pmc_traffic.json keyed by the kernel sources' hash, (5) the command line on 10 M pairs.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r04_grch38; mkdir -p $OUT
T0=$(date +%s)
python tools/build_bench_index.py 3.1e9 > $OUT/build.log 2>&1; tail -1 $OUT/build.log | cut -c1-300
echo "index after $(( $(date +%s) - T0 )) s"
rocprofv3 -L 2>/dev/null | grep -i -E "TCC_EA0_(RD|WR)REQ|TCC_EA_(RD|WR)REQ|FETCH_SIZE|WRITE_SIZE|TCC_REQ|TCC_HIT|TCC_MISS" | head -40 > $OUT/counters_avail.txt
# (1)
for cfg in "0 0" "16 0" "0 1" "16 1"; do
  set -- $cfg; export H2G_FAST_TAIL=$1 H2G_FAST_AM=$2
  timeout 900 python tools/fast_perf.py pe 1000000 3.1e9 > $OUT/pe_t$1_am$2.log 2>&1; echo "tail=$1 am=$2: $(tail -1 $OUT/pe_t$1_am$2.log | cut -c1-330)"
done
unset H2G_FAST_TAIL H2G_FAST_AM
H2G_LIB=$PWD/hisat2_amd/csrc/obj/libh2g_prof.so timeout 900 python tools/fast_perf.py pe 1000000 3.1e9 > $OUT/pe_prof.log 2>&1; tail -24 $OUT/pe_prof.log | cut -c1-300
BEST=$(python - <<'PY'
import re, glob
best = None
for f in glob.glob("gpurun_out/r04_grch38/pe_t*_am*.log"):
    m = re.search(r"steady ([0-9.]+) ms", open(f).read())
    t, am = re.search(r"pe_t(\d+)_am(\d+)", f).groups()
    if m and (best is None or float(m.group(1)) < best[0]): best = (float(m.group(1)), t, am)
print(best[1], best[2])
PY
)
set -- $BEST; export H2G_FAST_TAIL=$1 H2G_FAST_AM=$2
echo "best: tail=$1 am=$2 after $(( $(date +%s) - T0 )) s" | tee $OUT/best.txt
# (2)
python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_best.json 2> $OUT/bench_best.err; head -c 1500 $OUT/bench_best.json; echo
# (3)
CMD="python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 1"
rocprofv3 --kernel-trace --stats -d /tmp/bp_trace -- $CMD > $OUT/bench_traced.json 2> /tmp/bp_trace.err
python tools/rocpd_summary.py /tmp/bp_trace > $OUT/bench_trace.txt 2>&1; head -8 $OUT/bench_trace.txt | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/bp_pmc
  rocprofv3 --pmc $c -d /tmp/bp_pmc -- $CMD > /dev/null 2> /tmp/bp_pmc.err
  echo "# rocprofv3 --pmc $c -- $CMD   (H2G_FAST_TAIL=$H2G_FAST_TAIL H2G_FAST_AM=$H2G_FAST_AM)" > $OUT/bench_pmc_$c.txt
  python tools/rocpd_summary.py /tmp/bp_pmc >> $OUT/bench_pmc_$c.txt 2>&1
done
grep -E "k_go_fast" $OUT/bench_pmc_FETCH_SIZE.txt $OUT/bench_pmc_WRITE_SIZE.txt | cut -c1-220
echo "profiles after $(( $(date +%s) - T0 )) s"
# (4)
python - <<'PY'
import json, sys, os
sys.path.insert(0, ".")
import bench
OUT = "gpurun_out/r04_grch38"
def mean(path, counter, kern):
    for l in open(path):
        if l.startswith(kern) and counter in l:
            return float(l.split()[-1])
b = json.loads(open(OUT + "/bench_best.json").read().strip().splitlines()[-1])
kern = "k_go_fast_am" if os.environ.get("H2G_FAST_AM") == "1" else "k_go_fast"
f = mean(OUT + "/bench_pmc_FETCH_SIZE.txt", "FETCH_SIZE", kern); w = mean(OUT + "/bench_pmc_WRITE_SIZE.txt", "WRITE_SIZE", kern)
rec = {"pairs_per_launch": b["config"]["pairs_per_gpu"], "genome": b["config"]["genome_bases"], "kernel": kern, "kernel_sources_sha16": bench.kernel_sources_sha16(),
       "tail": int(os.environ.get("H2G_FAST_TAIL", 0)), "align_mate": int(os.environ.get("H2G_FAST_AM", 0)),
       "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
       "traffic_bytes_per_launch": int((f + w) * 1024) if f and w else None,
       "traffic_upper_bound_bytes": int((2 * f + w) * 1024) if f and w else None,
       "calibration": "FETCH_SIZE counts fabric read requests x 64 B: exact for the scattered 64 B sides (k_rank_v0 at 2^28 queries: 17.51 GB reported for 17.18 GB of sides + 1.34 GB of query input), "
                      "half for 128 B requests (k_rank_g0: 18.43 GB reported for 34.36 + 1.34 GB) - profiles/r04_rank_pmc_*.txt.  traffic = FETCH_SIZE + WRITE_SIZE (every request at 64 B: a lower bound, "
                      "exact for the index lines); upper bound = 2 x FETCH_SIZE + WRITE_SIZE (every read request a 128 B one)",
       "source": "profiles/r04_bench_pmc_FETCH_SIZE.txt + r04_bench_pmc_WRITE_SIZE.txt: rocprofv3 --pmc, separate passes of `python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 1`, mean per launch"}
json.dump(rec, open(OUT + "/pmc_traffic.json", "w"))
print(json.dumps(rec)[:900])
PY
# (5) the command line at this size, 10 M pairs
H2G_CLI_GENOME=3.1e9 timeout 1500 python tools/cli_perf.py 10000000 > $OUT/cli_10M_pairs.log 2>&1; tail -3 $OUT/cli_10M_pairs.log | cut -c1-400
echo "done after $(( $(date +%s) - T0 )) s"
