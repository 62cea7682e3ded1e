#!/bin/bash
# Round 6, lease D: resident batches again (test fixed; result rows allocated at selection), the new tests (sampled rank check against the oracle, the -2-shorter error path),
# compile-flag variants of the graph fast unit, bench.py at 256 Mbp over 10 distinct batches.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_d; mkdir -p $OUT
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_batches.py tests/test_gpu_parity.py tests/test_gpu_sam.py -m gpu -x -q > $OUT/gputests_new.log 2>&1; tail -4 $OUT/gputests_new.log | cut -c1-300
echo "tests after $(( $(date +%s) - T0 )) s"
for lib in new gnosink gos; do
  if [ $lib = new ]; then unset H2G_LIB; else export H2G_LIB=$PWD/hisat2_amd/csrc/obj/libh2g_$lib.so; fi
  timeout 600 python tools/queued_steps.py graph 256e6 1000000 "8,128,0,0" > $OUT/graph_$lib.jsonl 2> $OUT/graph_$lib.err; echo "graph $lib: $(tail -1 $OUT/graph_$lib.jsonl | cut -c1-420)"
done
unset H2G_LIB
echo "graph after $(( $(date +%s) - T0 )) s"
H2G_BENCH_GENOME=256e6 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $OUT/bench_256Mbp.json 2> $OUT/bench_256Mbp.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r06_d/bench_256Mbp.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print({k: d.get(k) for k in ("value", "ms_per_step")}, {k: r.get(k) for k in ("kernel_ms", "frac", "traffic")}, d.get("parity_whole_batch", {}).get("digest_equal"), d["config"]["workload"][:200])
except Exception as e:
    print("bench line:", repr(e)); print(open("gpurun_out/r06_d/bench_256Mbp.err").read()[-1500:])
PY
echo "done after $(( $(date +%s) - T0 )) s"
