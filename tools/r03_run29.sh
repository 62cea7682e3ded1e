cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
H2G_BENCH_GENOME=256e6 timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/r03_bench_256Mbp.json 2> $OUT/r03_bench_256Mbp.err
tail -c 300 $OUT/r03_bench_256Mbp.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_256Mbp.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['machine_pass_ms'], d['config']['workload'][:120])
PY
