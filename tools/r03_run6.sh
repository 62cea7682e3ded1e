cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
{
for n in 1000 5000 20000 100000; do for f in 0 1; do H2G_GO_FAST=$f timeout 300 python tools/fast_perf.py pe $n 2>&1 | tail -1 | cut -c1-260; done; done
} > $OUT/fast_perf6.log 2>&1
cat $OUT/fast_perf6.log
