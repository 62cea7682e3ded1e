cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
L=$OUT/r03_run24.log; : > $L
H2G_LIB=$PWD/hisat2_amd/libh2g_v9.so timeout 300 python tools/fast_perf.py pe 1000000 >> $L 2>&1
grep -v "^index ready" $L
