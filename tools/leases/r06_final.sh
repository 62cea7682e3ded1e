#!/bin/bash
# Round 6, the FINAL lease (the sources that ship: drain launch, parked alignMate pairs, slow-form queues of the graph pass).
# ONE lease at the metric's size (the 3.1 Gbp index cannot travel: built on the box first).  In order of worth:
#  (1) the headline line (whole-batch parity of the timed batch against the reference binary, pcie-inclusive, cpu baseline, command line on 1 M pairs),
#  (2) rocprofv3 kernel trace of the same command, (3) FETCH_SIZE / WRITE_SIZE passes, (4) profiles/r06_pmc_traffic.json keyed by the kernel sources' hash,
#  (5) the pipeline-depth ledger at this size, (6) the fast kernel's time split, (7) the command line on 10 M pairs.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_final; mkdir -p $OUT
T0=$(date +%s)
# (0) bench.py's loop over 10 distinct batches on a fresh box, small genome: 20 / 5 must read the steady state now that a batch's rows are written at selection
H2G_BENCH_GENOME=256e6 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_256_first.json 2> $OUT/bench_256_first.err
python -c "
import json; d = json.loads(open('$OUT/bench_256_first.json').read().strip().splitlines()[-1]); print('256 Mbp 20/5:', d['ms_per_step'], d['roofline']['kernel_ms'])"
echo "check after $(( $(date +%s) - T0 )) s" | tee $OUT/timeline.txt
# (1) the DRIVER's command: index build, headline, parity, cpu baseline, every extra leg, the repeat-structured companion at the metric's size while time lasts
timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc $?"; head -c 3000 $OUT/bench_driver.json; echo
echo "driver command after $(( $(date +%s) - T0 )) s" | tee -a $OUT/timeline.txt
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r06_final/bench_driver.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step")}, {k: d["roofline"].get(k) for k in ("kernel_ms", "frac", "traffic")}, d.get("parity_whole_batch", {}).get("digest_equal"), d["config"]["workload"][:160])
    for leg in ("repeat_pe", "graph256_pe", "spliced_pe", "ecoli_se", "repeat_grch38size_pe"):
        v = d.get(leg)
        if isinstance(v, dict):
            print(leg, {k: v.get(k) for k in ("ms_per_step", "fast_kernel_ms", "hand_on_rate", "index_build_s", "skipped", "error")}, (v.get("parity_whole_batch") or {}).get("digest_equal"))
    print("rank:", {k: (v.get("GB/s") if isinstance(v, dict) else v) for k, v in d.get("rank_microbench", {}).items()})
    print("cpu:", {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "cores")}, "cli:", d.get("cli_end_to_end", {}).get("reads_per_s_wall"), "pcie:", d.get("pcie_inclusive", {}).get("reads_per_s"))
except Exception as e:
    print("driver line:", repr(e))
PY
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
OUT = "gpurun_out/r06_final"
def mean(path, counter, kern):
    for l in open(path):
        if l.startswith(kern + "(") and counter in l:
            return float(l.split()[-1])
try:
    b = json.loads(open(OUT + "/bench_traced.json").read().strip().splitlines()[-1])
    kern = "k_go_fast"
    f = mean(OUT + "/bench_pmc_FETCH_SIZE.txt", "FETCH_SIZE", kern); w = mean(OUT + "/bench_pmc_WRITE_SIZE.txt", "WRITE_SIZE", kern)
    fd = mean(OUT + "/bench_pmc_FETCH_SIZE.txt", "FETCH_SIZE", "k_go_fast_am_drain"); wd = mean(OUT + "/bench_pmc_WRITE_SIZE.txt", "WRITE_SIZE", "k_go_fast_am_drain")
    rec = {"pairs_per_launch": b["config"]["pairs_per_gpu"], "genome": b["config"]["genome_bases"], "kernel": kern, "kernel_sources_sha16": bench.kernel_sources_sha16(),
           "tail": 16, "align_mate": 0, "drain_launch": {"kernel": "k_go_fast_am_drain", "FETCH_SIZE_KB": fd, "WRITE_SIZE_KB": wd, "traffic_bytes_per_launch": int((fd + wd) * 1024) if fd and wd else None}, "distinct_batches": b["config"].get("distinct_batches"), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
           "traffic_bytes_per_launch": int((f + w) * 1024) if f and w else None,
           "traffic_upper_bound_bytes": int((2 * f + w) * 1024) if f and w else None,
           "calibration": "FETCH_SIZE counts fabric read requests x 64 B: exact for the scattered 64 B sides (k_rank_v0 at 2^28 queries: 17.51 GB reported for 17.18 GB of sides + 1.34 GB of query input), "
                          "half for 128 B requests (k_rank_g0: 18.43 GB reported for 34.36 + 1.34 GB) - profiles/r04_rank_pmc_*.txt.  traffic = FETCH_SIZE + WRITE_SIZE (every request at 64 B: a lower bound, "
                          "exact for the index lines); upper bound = 2 x FETCH_SIZE + WRITE_SIZE (every read request a 128 B one)",
           "source": "profiles/r06_final/bench_pmc_FETCH_SIZE.txt + bench_pmc_WRITE_SIZE.txt: rocprofv3 --pmc, separate passes of `python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2`, mean per launch",
           "note": "taken in the final GRCh38-size lease of round 6 (tools/leases/r06_final.sh) on the kernel sources with this hash; traffic_bytes_per_launch is k_go_fast's own (the drain launch's is beside it)"}
    json.dump(rec, open(OUT + "/pmc_traffic.json", "w"), indent=1)
    print(json.dumps(rec)[:700])
except Exception as e:
    print("pmc record failed:", repr(e))
PY
# (5)
timeout 500 python tools/queued_steps.py rnd 3.1e9 1000000 "8,128,0,0,-1,64,-1;8,128,0,0,0,64,0;8,128,0,0,-1,64,-1;8,128,0,0,512,64,0" > $OUT/mstreams_grch38.jsonl 2> $OUT/mstreams_grch38.err; cut -c1-330 $OUT/mstreams_grch38.jsonl; tail -2 $OUT/mstreams_grch38.err
echo "sweep after $(( $(date +%s) - T0 )) s" | tee -a $OUT/timeline.txt
# (6)
H2G_LIB=$PWD/hisat2_amd/csrc/obj_prof/libh2g_prof.so timeout 500 python tools/fast_perf.py pe 1000000 3.1e9 > $OUT/fast_prof_grch38.log 2>&1; tail -24 $OUT/fast_prof_grch38.log | cut -c1-300
echo "prof after $(( $(date +%s) - T0 )) s" | tee -a $OUT/timeline.txt
# (7)
H2G_CLI_GENOME=3.1e9 timeout 900 python tools/cli_perf.py 10000000 > $OUT/cli_10M_pairs.log 2>&1; tail -3 $OUT/cli_10M_pairs.log | cut -c1-500
echo "done after $(( $(date +%s) - T0 )) s" | tee -a $OUT/timeline.txt
# (8) the 256 Mbp legs on the shipped sources: kernel trace + FETCH_SIZE / WRITE_SIZE of the repeat-structured and SNP-graph legs (-> r06_{rep,graph}_pmc_traffic.json, attached by source hash)
for leg in rep graph; do
  CMD="python tools/queued_steps.py $leg 256e6 1000000 8,128,0,0"
  rm -rf /tmp/bp_trace
  timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/bp_trace -- $CMD > $OUT/${leg}_traced.jsonl 2> /tmp/bp_trace.err
  python tools/rocpd_summary.py /tmp/bp_trace > $OUT/${leg}_trace.txt 2>&1; head -7 $OUT/${leg}_trace.txt | cut -c1-200
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/bp_pmc
    timeout 500 rocprofv3 --pmc $c -d /tmp/bp_pmc -- $CMD > $OUT/${leg}_pmc_run.jsonl 2> /tmp/bp_pmc.err
    echo "# rocprofv3 --pmc $c -- $CMD" > $OUT/${leg}_pmc_$c.txt
    python tools/rocpd_summary.py /tmp/bp_pmc >> $OUT/${leg}_pmc_$c.txt 2>&1
  done
  grep -E "k_go_fast" $OUT/${leg}_pmc_FETCH_SIZE.txt $OUT/${leg}_pmc_WRITE_SIZE.txt | grep SIZE | cut -c1-220
  echo "$leg profiles after $(( $(date +%s) - T0 )) s" | tee -a $OUT/timeline.txt
done
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
OUT = "gpurun_out/r06_final"
def mean(path, counter, kern):
    for l in open(path):
        if l.startswith(kern + "(") and counter in l:
            return float(l.split()[-1])
for leg, kern, dk in (("rep", "k_go_fast", "k_go_fast_am_drain"), ("graph", "k_go_fast_graph", "k_go_fast_graph_drain")):
    try:
        f = mean(OUT + "/%s_pmc_FETCH_SIZE.txt" % leg, "FETCH_SIZE", kern); w = mean(OUT + "/%s_pmc_WRITE_SIZE.txt" % leg, "WRITE_SIZE", kern)
        fd = mean(OUT + "/%s_pmc_FETCH_SIZE.txt" % leg, "FETCH_SIZE", dk); wd = mean(OUT + "/%s_pmc_WRITE_SIZE.txt" % leg, "WRITE_SIZE", dk)
        rec = {"leg": leg, "pairs_per_launch": 1000000, "genome": 256000000, "kernel": kern, "kernel_sources_sha16": bench.kernel_sources_sha16(), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
               "traffic_bytes_per_launch": int((f + w) * 1024), "traffic_upper_bound_bytes": int((2 * f + w) * 1024),
               "drain_launch": {"kernel": dk, "FETCH_SIZE_KB": fd, "WRITE_SIZE_KB": wd, "traffic_bytes_per_launch": int((fd + wd) * 1024) if fd and wd else None},
               "source": "profiles/r06_final/%s_pmc_FETCH_SIZE.txt + %s_pmc_WRITE_SIZE.txt: rocprofv3 --pmc, separate passes of `python tools/queued_steps.py %s 256e6 1000000 8,128,0,0`, mean per launch" % (leg, leg, leg),
               "calibration": "FETCH_SIZE counts fabric read requests x 64 B (exact for 64 B sides, half for 128 B graph sides: profiles/r04_rank_pmc.json); traffic = FETCH_SIZE + WRITE_SIZE is a lower bound"}
        json.dump(rec, open(OUT + "/%s_pmc_traffic.json" % leg, "w"), indent=1)
        print(json.dumps(rec)[:500])
    except Exception as e:
        print(leg, "pmc record failed:", repr(e))
PY
# (9) the SNP graph at 1 Gbp (its own index build: ~5 min)
H2G_GRAPH_LEG_GENOME=1e9 timeout 1500 python bench.py --only-legs graph_big_pe > $OUT/graph_1gbp.json 2> $OUT/graph_1gbp.err; echo "graph 1 Gbp rc $?"; head -c 1500 $OUT/graph_1gbp.json; echo; tail -2 $OUT/graph_1gbp.err | cut -c1-300
echo "graph 1 Gbp after $(( $(date +%s) - T0 )) s" | tee -a $OUT/timeline.txt
# (10) the whole GPU suite + smoke on these sources
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; tail -4 $OUT/gputests.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
echo "all after $(( $(date +%s) - T0 )) s" | tee -a $OUT/timeline.txt
