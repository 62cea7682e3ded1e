cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
L=$OUT/r03_run23.log; : > $L
H2G_STEADY=20 timeout 300 python tools/fast_perf.py pe 1000000 >> $L 2>&1
H2G_STEADY=20 timeout 300 python tools/fast_perf.py se 1000000 >> $L 2>&1
timeout 900 python -m pytest tests/test_gpu_fast_pass.py -x -q -m gpu > $OUT/r03_run23_tests.log 2>&1
tail -3 $OUT/r03_run23_tests.log
grep -v "^index ready\|bails:" $L
