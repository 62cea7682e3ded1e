cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
L=$OUT/r03_run12.log; : > $L
for kn in "150 4" "300 4" "450 4" "600 4" "1000 4" "300 12"; do
  set -- $kn
  echo "== libh2g.so MACH_DIV=$1 MACH_MIN=$2" >> $L
  H2G_STEADY=20 H2G_MACH_DIV=$1 H2G_MACH_MIN=$2 timeout 300 python tools/fast_perf.py pe 1000000 >> $L 2>&1
done
echo "== se 300" >> $L
H2G_STEADY=20 H2G_MACH_DIV=300 timeout 300 python tools/fast_perf.py se 1000000 >> $L 2>&1
timeout 900 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_align.py -x -q -m gpu > $OUT/r03_run12_tests.log 2>&1
tail -5 $OUT/r03_run12_tests.log
grep -v "^index ready\|bails:" $L
