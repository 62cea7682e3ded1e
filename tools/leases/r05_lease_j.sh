#!/bin/bash
# Round 5, lease J: why does bench.py's 20-step headline loop sit 4-5 ms per step above the 30-step sweep on a 256 Mbp genome (leases H, I: 19.0 against 14.4) when it did not at 3.1 Gbp (lease F: 16.1 against 15.7)?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r05_j; mkdir -p $OUT
T0=$(date +%s)
timeout 400 python tools/r05_mstreams.py rnd 256e6 1000000 "8,128,0,0" > $OUT/sweep.jsonl 2> $OUT/sweep.err; tail -1 $OUT/sweep.jsonl | cut -c1-300
for cfg in "20 5" "60 10" "20 12"; do
  set -- $cfg
  timeout 400 python bench.py --genome 256e6 --no-extras --no-cpu-baseline --steps $1 --warmup $2 > $OUT/bench_s$1_w$2.json 2> $OUT/bench_s$1_w$2.err
  python - "$OUT/bench_s$1_w$2.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], {k: d.get(k) for k in ("steps", "warmup", "ms_per_step")}, {k: d["roofline"].get(k) for k in ("kernel_ms", "machine_pass_ms", "pairs_handed_on")})
except Exception as e:
    print("no line:", e)
PY
done
H2G_MSTREAMS=2 H2G_FAST_RESERVE=-1 H2G_MACH_TOTAL=96 timeout 400 python bench.py --genome 256e6 --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_M2.json 2> $OUT/bench_M2.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05_j/bench_M2.json").read().strip().splitlines()[-1])
    print("M=2 (round 4's settings):", {k: d.get(k) for k in ("steps", "warmup", "ms_per_step")}, {k: d["roofline"].get(k) for k in ("kernel_ms", "machine_pass_ms")})
except Exception as e:
    print("no line:", e)
PY
echo "done after $(( $(date +%s) - T0 )) s"
