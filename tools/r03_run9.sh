cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
H2G_BENCH_GENOME=40e6 timeout 900 python bench.py --steps 5 --warmup 2 --no-extras > $OUT/r03_small_bench.json 2> $OUT/r03_small_bench.err
tail -c 3000 $OUT/r03_small_bench.err; head -c 1500 $OUT/r03_small_bench.json
