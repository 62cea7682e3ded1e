cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
L=$OUT/r03_run14.log; : > $L
for lib in libh2g.so libh2g_v1.so libh2g_v2.so libh2g_v3.so libh2g_v9.so; do
  echo "== $lib" >> $L
  H2G_STEADY=20 H2G_LIB=$PWD/hisat2_amd/$lib timeout 300 python tools/fast_perf.py pe 1000000 >> $L 2>&1
done
grep -v "^index ready\|bails:" $L
