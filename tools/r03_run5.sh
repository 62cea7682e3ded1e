cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
{
for lib in $LIBS; do echo "== $lib"; H2G_LIB=$PWD/hisat2_amd/$lib timeout 300 python tools/fast_perf.py ${MODE:-pe} 1000000 2>&1 | tail -16; done
} > $OUT/fast_perf5.log 2>&1
cat $OUT/fast_perf5.log
