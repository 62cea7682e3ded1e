cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/r03_gputests.log 2>&1
tail -4 $OUT/r03_gputests.log
H2G_BENCH_GENOME=40e6 timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/r03_small_bench.json 2> $OUT/r03_small_bench.err
tail -c 800 $OUT/r03_small_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_small_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])
for k in ('graph_index_pe','rank_microbench','rank_microbench_graph','cli_end_to_end','ecoli_se'):
    print(k, json.dumps(d.get(k))[:900])
PY
timeout 900 python tools/cli_perf.py 10000000 > $OUT/r03_cli_10M.log 2>&1
cat $OUT/r03_cli_10M.log
