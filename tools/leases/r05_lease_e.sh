#!/bin/bash
# Round 5, lease E: the GPU suite on the final sources (compact fetch, hidden block offsets, long edits, M = 8), then the whole bench.py flow on a reduced genome
# (40 Mbp: every leg, whole-batch parity, pcie-inclusive, command line) so that the GRCh38-size lease only has to repeat it at size.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r05_e; mkdir -p $OUT
T0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1; tail -6 $OUT/gputests.log | cut -c1-300
echo "tests after $(( $(date +%s) - T0 )) s"
H2G_BENCH_GENOME=40e6 timeout 1500 python bench.py --steps 10 --warmup 3 > $OUT/bench_40Mbp.json 2> $OUT/bench_40Mbp.err; echo "bench rc $?"; tail -3 $OUT/bench_40Mbp.err | cut -c1-400
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05_e/bench_40Mbp.json").read().strip().splitlines()[-1])
except Exception as e:
    print("no line:", e); raise SystemExit
def short(v, depth=0):
    if isinstance(v, dict):
        return {k: short(x, depth + 1) for k, x in v.items() if k not in ("note", "against", "workload", "sample", "threads_scan", "host_cpu_limits", "traffic_calibration", "calibration", "machine_pass_note", "algorithmic", "kernel_times_are")}
    if isinstance(v, str) and len(v) > 90:
        return v[:90] + "..."
    return v
print(json.dumps(short(d))[:6000])
PY
echo "done after $(( $(date +%s) - T0 )) s"
