#!/bin/bash
# Round 6, lease F: bench.py's own loop FIRST on a fresh box (256 Mbp, no extras: is its step the steady state's?), then again after the tests; the tests that changed (device-side
# combineWith vectors, the local index in registers under fast == machine); the wave-level time split with the register-resident local index.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_f; mkdir -p $OUT
T0=$(date +%s)
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print({k: d.get(k) for k in ("value", "ms_per_step")}, {k: r.get(k) for k in ("kernel_ms", "frac", "traffic")}, d.get("parity_whole_batch", {}).get("digest_equal"))
except Exception as e:
    print("bench line:", repr(e))
PY
}
H2G_BENCH_GENOME=256e6 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_first.json 2> $OUT/bench_first.err; show $OUT/bench_first.json
H2G_BENCH_GENOME=256e6 timeout 900 python bench.py --gpus 1 --steps 60 --warmup 10 --no-extras --no-cpu-baseline > $OUT/bench_first_60.json 2> $OUT/bench_first_60.err; show $OUT/bench_first_60.json
echo "bench after $(( $(date +%s) - T0 )) s"
timeout 900 python -m pytest tests/test_gpu_align.py tests/test_gpu_fast_pass.py tests/test_gpu_fast_stress.py tests/test_gpu_pairs.py -m gpu -x -q > $OUT/gputests.log 2>&1; tail -4 $OUT/gputests.log | cut -c1-300
echo "tests after $(( $(date +%s) - T0 )) s"
H2G_LIB=$PWD/hisat2_amd/csrc/obj_prof/libh2g_prof.so timeout 600 python tools/fast_perf.py pe 1000000 256e6 > $OUT/fast_prof_rnd256.log 2>&1; tail -14 $OUT/fast_prof_rnd256.log | cut -c1-700
timeout 600 python tools/queued_steps.py rnd 256e6 1000000 "8,128,0,0" > $OUT/rnd_new.jsonl 2> $OUT/rnd_new.err; echo "rnd: $(tail -1 $OUT/rnd_new.jsonl | cut -c1-420)"
H2G_BENCH_GENOME=256e6 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_after.json 2> $OUT/bench_after.err; show $OUT/bench_after.json
echo "done after $(( $(date +%s) - T0 )) s"
