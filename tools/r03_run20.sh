cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
L=$OUT/r03_run20.log; : > $L
H2G_STEADY=20 timeout 300 python tools/fast_perf.py pe 1000000 >> $L 2>&1
H2G_STEADY=20 timeout 300 python tools/fast_perf.py se 1000000 >> $L 2>&1
H2G_BENCH_GENOME=40e6 timeout 900 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/r03_small_bench.json 2> $OUT/r03_small_bench.err
grep -v "^index ready\|bails:" $L
python -c "
import json; d=json.loads(open('$OUT/r03_small_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['machine_pass_ms'], d['roofline']['frac'])"
