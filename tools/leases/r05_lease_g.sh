#!/bin/bash
# Round 5, lease G: the GPU suite on the final sources; trace + FETCH_SIZE / WRITE_SIZE of the repeat-structured and SNP-graph legs on those sources (-> profiles/r05_{rep,graph}_pmc_traffic.json,
# attached to the legs' roofline blocks by kernel-source hash); tail hand-off x alignMate-in-the-pass under the 8-deep pipeline on a 256 Mbp random genome.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r05_g; mkdir -p $OUT
T0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1; tail -4 $OUT/gputests.log | cut -c1-300
echo "tests after $(( $(date +%s) - T0 )) s"
for leg in rep graph; do
  CMD="python tools/queued_steps.py $leg 256e6 1000000 8,128,0,0"
  if [ $leg = rep ]; then
    rm -rf /tmp/bp_trace
    timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/bp_trace -- $CMD > $OUT/${leg}_traced.jsonl 2> /tmp/bp_trace.err
    python tools/rocpd_summary.py /tmp/bp_trace > $OUT/${leg}_trace.txt 2>&1; head -6 $OUT/${leg}_trace.txt | cut -c1-200
  fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/bp_pmc
    timeout 400 rocprofv3 --pmc $c -d /tmp/bp_pmc -- $CMD > $OUT/${leg}_pmc_run.jsonl 2> /tmp/bp_pmc.err
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
OUT = "gpurun_out/r05_g"
def mean(path, counter, kern):
    for l in open(path):
        if l.startswith(kern + "(") and counter in l:
            return float(l.split()[-1])
for leg, kern in (("rep", "k_go_fast"), ("graph", "k_go_fast_graph")):
    try:
        f = mean(OUT + "/%s_pmc_FETCH_SIZE.txt" % leg, "FETCH_SIZE", kern); w = mean(OUT + "/%s_pmc_WRITE_SIZE.txt" % leg, "WRITE_SIZE", kern)
        rec = {"leg": leg, "pairs_per_launch": 1000000, "genome": 256000000, "kernel": kern, "kernel_sources_sha16": bench.kernel_sources_sha16(), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
               "traffic_bytes_per_launch": int((f + w) * 1024), "traffic_upper_bound_bytes": int((2 * f + w) * 1024),
               "source": "profiles/r05_g_%s_pmc_FETCH_SIZE.txt + r05_g_%s_pmc_WRITE_SIZE.txt: rocprofv3 --pmc, separate passes of `python tools/queued_steps.py %s 256e6 1000000 8,128,0,0`, mean per launch" % (leg, leg, leg),
               "calibration": "FETCH_SIZE counts fabric read requests x 64 B (exact for 64 B sides, half for 128 B graph sides: profiles/r04_rank_pmc.json); traffic = FETCH_SIZE + WRITE_SIZE is a lower bound"}
        json.dump(rec, open(OUT + "/%s_pmc_traffic.json" % leg, "w"), indent=1)
        print(json.dumps(rec)[:400])
    except Exception as e:
        print(leg, "pmc record failed:", repr(e))
PY
for cfg in "16 0" "0 0" "16 1" "0 1"; do
  set -- $cfg
  H2G_FAST_TAIL=$1 H2G_FAST_AM=$2 timeout 300 python tools/queued_steps.py rnd 256e6 1000000 "8,128,0,0" > $OUT/rnd256_tail$1_am$2.jsonl 2> $OUT/rnd256_tail$1_am$2.err; echo "tail=$1 am=$2: $(tail -1 $OUT/rnd256_tail$1_am$2.jsonl | cut -c1-330)"
done
echo "done after $(( $(date +%s) - T0 )) s"
