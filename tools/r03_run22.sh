cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
H2G_BENCH_GENOME=40e6 timeout 1200 python bench.py --steps 10 --warmup 3 > $OUT/r03_small_bench.json 2> $OUT/r03_small_bench.err
tail -c 1500 $OUT/r03_small_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_small_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d.get('spliced_pe'), indent=1))
pass
PY
