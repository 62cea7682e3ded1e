#!/bin/bash
# Round 6, the last lease: the driver's own sequence on the tree that ships — `python bench.py` with its defaults (the PMC records of r06_final.sh attach by source hash), the whole GPU
# suite, smoke()
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_last; mkdir -p $OUT
T0=$(date +%s)
timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc $? after $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r06_last/bench_driver.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print({k: d.get(k) for k in ("value", "ms_per_step")}, {k: r.get(k) for k in ("kernel_ms", "frac", "traffic", "algorithmic_bytes_per_launch", "pairs_handed_on")}, d.get("parity_whole_batch", {}).get("digest_equal"))
    print("drain:", r.get("drain_launch"))
    for leg in ("repeat_pe", "graph256_pe", "ecoli_se", "repeat_grch38size_pe"):
        v = d.get(leg)
        if isinstance(v, dict):
            print(leg, {k: v.get(k) for k in ("ms_per_step", "fast_kernel_ms", "hand_on_rate", "index_build_s", "skipped", "error")}, (v.get("parity_whole_batch") or {}).get("digest_equal"), (v.get("roofline") or {}).get("traffic"))
    print("cpu:", {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "cores")}, "cli:", d.get("cli_end_to_end", {}), "pcie:", d.get("pcie_inclusive", {}).get("reads_per_s"))
    print("keys:", [k for k in d.keys()])
except Exception as e:
    print("driver line:", repr(e))
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; tail -4 $OUT/gputests.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
echo "all after $(( $(date +%s) - T0 )) s"
