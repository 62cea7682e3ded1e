#!/bin/bash
# Round 6: the companion's fall-back path (a repeat-structured genome SMALLER than the headline's when the run's time does not cover the metric's size) at a reduced scale:
# headline genome 1.5 Gbp, the run's limit set so that the companion has to shrink
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_last3; mkdir -p $OUT
T0=$(date +%s)
H2G_BENCH_GENOME=1.5e9 H2G_BENCH_HARD_LIMIT=${HARD:-1350} timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_1500Mbp.json 2> $OUT/bench_1500Mbp.err; echo "bench rc $? after $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_last3/bench_1500Mbp.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step")}, d["config"]["workload"][:120])
for k, v in d.items():
    if k.startswith("repeat_") and isinstance(v, dict):
        print(k, {q: v.get(q) for q in ("workload", "ms_per_step", "fast_kernel_ms", "hand_on_rate", "index_build_s", "skipped", "error")}, (v.get("parity_whole_batch") or {}).get("digest_equal"))
PY
tail -3 $OUT/bench_1500Mbp.err | cut -c1-300
