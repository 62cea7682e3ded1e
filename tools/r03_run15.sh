cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
L=$OUT/r03_run15.log; : > $L
for lib in libh2g_k65.so libh2g_k48.so libh2g_k32.so libh2g_k24.so libh2g_k16.so libh2g_v9.so; do
  echo "== $lib" >> $L
  H2G_STEADY=20 H2G_LIB=$PWD/hisat2_amd/$lib timeout 300 python tools/fast_perf.py pe 1000000 >> $L 2>&1
done
echo "== se k24" >> $L
H2G_STEADY=20 H2G_LIB=$PWD/hisat2_amd/libh2g_k24.so timeout 300 python tools/fast_perf.py se 1000000 >> $L 2>&1
grep -v "^index ready\|bails:" $L
