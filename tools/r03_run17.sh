cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
L=$OUT/r03_run17.log; : > $L
for v in a b c; do
echo "== $v" >> $L
H2G_LIB=$PWD/hisat2_amd/libh2g_$v.so timeout 300 python tools/fast_perf.py se 1000000 >> $L 2>&1
done
grep -v "^index ready\|bails:" $L
