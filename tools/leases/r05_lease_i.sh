#!/bin/bash
# Round 5, lease I: the command line with its parser and writer threads (GPU suite: every CLI test against the reference), the headline loop with every buffer allocated
# by the first run, the command line on 10 M pairs (E. coli-size index).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r05_i; mkdir -p $OUT
T0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1; tail -4 $OUT/gputests.log | cut -c1-300
echo "tests after $(( $(date +%s) - T0 )) s"
timeout 600 python tools/cli_perf.py 10000000 > $OUT/cli_10M_pairs_ecoli.log 2>&1; tail -3 $OUT/cli_10M_pairs_ecoli.log | cut -c1-500
echo "cli after $(( $(date +%s) - T0 )) s"
H2G_BENCH_GENOME=256e6 timeout 900 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_256Mbp_headline.json 2> $OUT/bench_256Mbp_headline.err; echo "bench rc $?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05_i/bench_256Mbp_headline.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step")}, {k: d["roofline"].get(k) for k in ("frac", "kernel_ms", "machine_pass_ms")}, d.get("pcie_inclusive", {}).get("reads_per_s"))
except Exception as e:
    print("no line:", e)
PY
echo "done after $(( $(date +%s) - T0 )) s"
