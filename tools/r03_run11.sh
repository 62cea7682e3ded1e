cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
L=$OUT/r03_run11.log; : > $L
echo "== libh2g.so TAIL=0" >> $L
H2G_GO_TAIL=0 timeout 300 python tools/fast_perf.py pe 1000000 >> $L 2>&1
for lib in libh2g.so libh2g_v2.so libh2g_v3.so libh2g_v4.so libh2g_v5.so libh2g_v7.so; do
  echo "== $lib" >> $L
  H2G_LIB=$PWD/hisat2_amd/$lib timeout 300 python tools/fast_perf.py pe 1000000 >> $L 2>&1
done
echo "== v2 se" >> $L
H2G_LIB=$PWD/hisat2_amd/libh2g_v2.so timeout 300 python tools/fast_perf.py se 1000000 >> $L 2>&1
grep -v "^index ready\|bails:" $L
