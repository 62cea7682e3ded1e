#!/bin/bash
# Round 6, lease H: the SNP-graph leg at 1 Gbp with its whole-batch parity; trace + FETCH_SIZE / WRITE_SIZE of the graph and repeat-structured 256 Mbp legs on
# the final kernel sources (-> profiles/r06_{graph,rep}_pmc_traffic.json, attached to the legs' roofline blocks by source hash); the whole GPU suite on the final sources.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_h; mkdir -p $OUT
T0=$(date +%s)
H2G_GRAPH_LEG_GENOME=1e9 timeout 2400 python bench.py --only-legs graph_big_pe > $OUT/graph_1gbp.json 2> $OUT/graph_1gbp.err; echo "graph 1 Gbp rc $?"; head -c 1800 $OUT/graph_1gbp.json; echo; tail -3 $OUT/graph_1gbp.err | cut -c1-300
echo "graph 1 Gbp after $(( $(date +%s) - T0 )) s"
for leg in rep graph; do
  CMD="python tools/queued_steps.py $leg 256e6 1000000 8,128,0,0"
  rm -rf /tmp/bp_trace
  timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/bp_trace -- $CMD > $OUT/${leg}_traced.jsonl 2> /tmp/bp_trace.err
  python tools/rocpd_summary.py /tmp/bp_trace > $OUT/${leg}_trace.txt 2>&1; head -6 $OUT/${leg}_trace.txt | cut -c1-200
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/bp_pmc
    timeout 500 rocprofv3 --pmc $c -d /tmp/bp_pmc -- $CMD > $OUT/${leg}_pmc_run.jsonl 2> /tmp/bp_pmc.err
    echo "# rocprofv3 --pmc $c -- $CMD" > $OUT/${leg}_pmc_$c.txt
    python tools/rocpd_summary.py /tmp/bp_pmc >> $OUT/${leg}_pmc_$c.txt 2>&1
  done
  grep -E "k_go_fast" $OUT/${leg}_pmc_FETCH_SIZE.txt $OUT/${leg}_pmc_WRITE_SIZE.txt | grep SIZE | cut -c1-220
  echo "$leg profiles after $(( $(date +%s) - T0 )) s"
done
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
OUT = "gpurun_out/r06_h"
def mean(path, counter, kern):
    for l in open(path):
        if l.startswith(kern + "(") and counter in l:
            return float(l.split()[-1])
for leg, kern in (("rep", "k_go_fast"), ("graph", "k_go_fast_graph")):
    try:
        f = mean(OUT + "/%s_pmc_FETCH_SIZE.txt" % leg, "FETCH_SIZE", kern); w = mean(OUT + "/%s_pmc_WRITE_SIZE.txt" % leg, "WRITE_SIZE", kern)
        rec = {"leg": leg, "pairs_per_launch": 1000000, "genome": 256000000, "kernel": kern, "kernel_sources_sha16": bench.kernel_sources_sha16(), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
               "traffic_bytes_per_launch": int((f + w) * 1024), "traffic_upper_bound_bytes": int((2 * f + w) * 1024),
               "source": "profiles/r06_h_%s_pmc_FETCH_SIZE.txt + r06_h_%s_pmc_WRITE_SIZE.txt: rocprofv3 --pmc, separate passes of `python tools/queued_steps.py %s 256e6 1000000 8,128,0,0`, mean per launch" % (leg, leg, leg),
               "calibration": "FETCH_SIZE counts fabric read requests x 64 B (exact for 64 B sides, half for 128 B graph sides: profiles/r04_rank_pmc.json); traffic = FETCH_SIZE + WRITE_SIZE is a lower bound"}
        json.dump(rec, open(OUT + "/%s_pmc_traffic.json" % leg, "w"), indent=1)
        print(json.dumps(rec)[:400])
    except Exception as e:
        print(leg, "pmc record failed:", repr(e))
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; tail -4 $OUT/gputests.log | cut -c1-300
echo "done after $(( $(date +%s) - T0 )) s"
