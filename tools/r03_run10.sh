cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
L=$OUT/r03_run10.log; : > $L
for lib in libh2g_v0.so libh2g.so libh2g_v2.so; do
  echo "== $lib" >> $L
  H2G_LIB=$PWD/hisat2_amd/$lib timeout 300 python tools/fast_perf.py pe 1000000 >> $L 2>&1
done
for kn in "600 4" "300 4" "75 4" "150 16"; do
  set -- $kn
  echo "== libh2g.so MACH_DIV=$1 MACH_MIN=$2" >> $L
  H2G_MACH_DIV=$1 H2G_MACH_MIN=$2 timeout 300 python tools/fast_perf.py pe 1000000 >> $L 2>&1
done
echo "== se" >> $L
timeout 300 python tools/fast_perf.py se 1000000 >> $L 2>&1
timeout 900 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_ext_search.py -x -q -m gpu > $OUT/r03_run10_tests.log 2>&1
tail -5 $OUT/r03_run10_tests.log
grep -v "^index ready" $L
