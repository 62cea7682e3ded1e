cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
H2G_BENCH_GENOME=40e6 timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/r03_small_bench.json 2> $OUT/r03_small_bench.err
tail -c 1500 $OUT/r03_small_bench.err; head -c 2500 $OUT/r03_small_bench.json; echo
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/r03_gputests.log 2>&1
tail -8 $OUT/r03_gputests.log
