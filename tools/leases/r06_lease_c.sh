#!/bin/bash
# Round 6, lease C: resident batches (test + bench.py's timed loop over 10 distinct batches on a 256 Mbp genome, the driver's command); the graph fast unit with short in-edge
# lists, and its leaf functions inlined (variant), on the 256 Mbp SNP graph.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_c; mkdir -p $OUT
T0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_batches.py tests/test_gpu_fast_pass.py -m gpu -x -q > $OUT/gputests_batches.log 2>&1; tail -4 $OUT/gputests_batches.log | cut -c1-300
echo "tests after $(( $(date +%s) - T0 )) s"
for lib in new glfinl; do
  if [ $lib = new ]; then unset H2G_LIB; else export H2G_LIB=$PWD/hisat2_amd/csrc/obj/libh2g_$lib.so; fi
  timeout 600 python tools/queued_steps.py graph 256e6 1000000 "8,128,0,0" > $OUT/graph_$lib.jsonl 2> $OUT/graph_$lib.err; echo "graph $lib: $(tail -1 $OUT/graph_$lib.jsonl | cut -c1-420)"
done
unset H2G_LIB
echo "graph after $(( $(date +%s) - T0 )) s"
H2G_BENCH_GENOME=256e6 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_256Mbp.json 2> $OUT/bench_256Mbp.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r06_c/bench_256Mbp.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print({k: d.get(k) for k in ("value", "ms_per_step")}, {k: r.get(k) for k in ("kernel_ms", "frac", "traffic")}, d.get("parity_whole_batch", {}).get("digest_equal"), d["config"]["workload"][:200])
    for leg in ("repeat_pe", "graph256_pe", "spliced_pe", "ecoli_se"):
        v = d.get(leg)
        if isinstance(v, dict):
            print(leg, {k: v.get(k) for k in ("ms_per_step", "fast_kernel_ms", "hand_on_rate", "error")}, (v.get("parity_whole_batch") or {}).get("digest_equal"))
except Exception as e:
    print("bench line:", repr(e)); print(open("gpurun_out/r06_c/bench_256Mbp.err").read()[-1500:])
PY
echo "done after $(( $(date +%s) - T0 )) s"
