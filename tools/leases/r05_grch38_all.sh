#!/bin/bash
# ONE lease at the metric's size (the 3.1 Gbp index cannot travel: built on the box first).  In order of worth:
#  (1) the headline line (whole-batch parity of the timed batch against the reference binary, pcie-inclusive, cpu baseline, command line on 1 M pairs),
#  (2) rocprofv3 kernel trace of the same command, (3) FETCH_SIZE / WRITE_SIZE passes, (4) profiles/r05_pmc_traffic.json keyed by the kernel sources' hash,
#  (5) the pipeline-depth ledger at this size, (6) the fast kernel's time split, (7) the command line on 10 M pairs.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r05_grch38; mkdir -p $OUT
T0=$(date +%s)
python tools/build_bench_index.py 3.1e9 > $OUT/build.log 2>&1; tail -1 $OUT/build.log | cut -c1-300
echo "index after $(( $(date +%s) - T0 )) s" | tee $OUT/timeline.txt
# (1)
timeout 900 python bench.py --no-extras --steps 20 --warmup 5 > $OUT/bench_headline.json 2> $OUT/bench_headline.err; echo "bench rc $?"; head -c 2500 $OUT/bench_headline.json; echo
echo "headline after $(( $(date +%s) - T0 )) s" | tee -a $OUT/timeline.txt
# (2)
CMD="python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2"
rm -rf /tmp/bp_trace
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/bp_trace -- $CMD > $OUT/bench_traced.json 2> /tmp/bp_trace.err
python tools/rocpd_summary.py /tmp/bp_trace > $OUT/bench_trace.txt 2>&1; head -8 $OUT/bench_trace.txt | cut -c1-200
# (3)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/bp_pmc
  timeout 600 rocprofv3 --pmc $c -d /tmp/bp_pmc -- $CMD > /dev/null 2> /tmp/bp_pmc.err
  echo "# rocprofv3 --pmc $c -- $CMD" > $OUT/bench_pmc_$c.txt
  python tools/rocpd_summary.py /tmp/bp_pmc >> $OUT/bench_pmc_$c.txt 2>&1
done
grep -E "k_go" $OUT/bench_pmc_FETCH_SIZE.txt $OUT/bench_pmc_WRITE_SIZE.txt | grep SIZE | cut -c1-220
echo "profiles after $(( $(date +%s) - T0 )) s" | tee -a $OUT/timeline.txt
# (4)
python - <<'PY'
import json, sys, os
sys.path.insert(0, ".")
import bench
OUT = "gpurun_out/r05_grch38"
def mean(path, counter, kern):
    for l in open(path):
        if l.startswith(kern + "(") and counter in l:
            return float(l.split()[-1])
try:
    b = json.loads(open(OUT + "/bench_traced.json").read().strip().splitlines()[-1])
    kern = "k_go_fast"
    f = mean(OUT + "/bench_pmc_FETCH_SIZE.txt", "FETCH_SIZE", kern); w = mean(OUT + "/bench_pmc_WRITE_SIZE.txt", "WRITE_SIZE", kern)
    rec = {"pairs_per_launch": b["config"]["pairs_per_gpu"], "genome": b["config"]["genome_bases"], "kernel": kern, "kernel_sources_sha16": bench.kernel_sources_sha16(),
           "tail": 16, "align_mate": 0, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
           "traffic_bytes_per_launch": int((f + w) * 1024) if f and w else None,
           "traffic_upper_bound_bytes": int((2 * f + w) * 1024) if f and w else None,
           "calibration": "FETCH_SIZE counts fabric read requests x 64 B: exact for the scattered 64 B sides (k_rank_v0 at 2^28 queries: 17.51 GB reported for 17.18 GB of sides + 1.34 GB of query input), "
                          "half for 128 B requests (k_rank_g0: 18.43 GB reported for 34.36 + 1.34 GB) - profiles/r04_rank_pmc_*.txt.  traffic = FETCH_SIZE + WRITE_SIZE (every request at 64 B: a lower bound, "
                          "exact for the index lines); upper bound = 2 x FETCH_SIZE + WRITE_SIZE (every read request a 128 B one)",
           "source": "profiles/r05_bench_pmc_FETCH_SIZE.txt + r05_bench_pmc_WRITE_SIZE.txt: rocprofv3 --pmc, separate passes of `python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2`, mean per launch",
           "note": "taken in the one GRCh38-size lease of round 5 (tools/r05_grch38_all.sh) on the kernel sources with this hash"}
    json.dump(rec, open(OUT + "/pmc_traffic.json", "w"), indent=1)
    print(json.dumps(rec)[:700])
except Exception as e:
    print("pmc record failed:", repr(e))
PY
# (5)
timeout 500 python tools/queued_steps.py rnd 3.1e9 1000000 "8,128,0,0;2,96,-1,0;4,128,0,0;8,128,-1,0" > $OUT/mstreams_grch38.jsonl 2> $OUT/mstreams_grch38.err; cut -c1-330 $OUT/mstreams_grch38.jsonl; tail -2 $OUT/mstreams_grch38.err
echo "sweep after $(( $(date +%s) - T0 )) s" | tee -a $OUT/timeline.txt
# (6)
H2G_LIB=$PWD/hisat2_amd/csrc/obj_prof/libh2g_prof.so timeout 500 python tools/fast_perf.py pe 1000000 3.1e9 > $OUT/fast_prof_grch38.log 2>&1; tail -24 $OUT/fast_prof_grch38.log | cut -c1-300
echo "prof after $(( $(date +%s) - T0 )) s" | tee -a $OUT/timeline.txt
# (7)
H2G_CLI_GENOME=3.1e9 timeout 900 python tools/cli_perf.py 10000000 > $OUT/cli_10M_pairs.log 2>&1; tail -3 $OUT/cli_10M_pairs.log | cut -c1-500
echo "done after $(( $(date +%s) - T0 )) s" | tee -a $OUT/timeline.txt
