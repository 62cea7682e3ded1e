#!/bin/bash
# Round 5, lease H: the GPU suite on the final sources (192-edit working hits, tiled scan), then the driver's own command on a 256 Mbp genome (every leg, every parity check).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r05_h; mkdir -p $OUT
T0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1; tail -4 $OUT/gputests.log | cut -c1-300
echo "tests after $(( $(date +%s) - T0 )) s"
H2G_BENCH_GENOME=256e6 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_256Mbp.json 2> $OUT/bench_256Mbp.err; echo "bench rc $?"; tail -3 $OUT/bench_256Mbp.err | cut -c1-400
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05_h/bench_256Mbp.json").read().strip().splitlines()[-1])
except Exception as e:
    print("no line:", e); raise SystemExit
keep = {}
keep["headline"] = {k: d.get(k) for k in ("value", "ms_per_step")}
keep["roofline"] = {k: d["roofline"].get(k) for k in ("frac", "kernel_ms", "traffic", "machine_pass_ms")}
for k in ("parity_whole_batch", "pcie_inclusive", "cli_end_to_end", "parity_failed", "big_legs_skipped", "extras_skipped", "extras_error"):
    v = d.get(k)
    if isinstance(v, dict):
        v = {a: b for a, b in v.items() if a in ("digest_equal", "pairs_checked", "reads_per_s", "seconds", "reads_per_s_wall", "wall_s", "sam_lines_differing")}
    keep[k] = v
for leg in ("ecoli_se", "repeat_pe", "graph256_pe", "graph_index_pe", "spliced_pe"):
    v = d.get(leg)
    if isinstance(v, dict):
        keep[leg] = {a: (b if not isinstance(b, dict) else {x: y for x, y in b.items() if x in ("digest_equal", "frac", "traffic", "ms_per_step", "sam_lines_differing")}) for a, b in v.items()
                     if a in ("ms_per_step", "reads_per_s", "parity_whole_batch", "roofline", "error", "device_no_temp_splicesite", "hand_on_rate", "parity")}
print(json.dumps(keep)[:5000])
PY
echo "done after $(( $(date +%s) - T0 )) s"
