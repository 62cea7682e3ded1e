cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
{
timeout 900 python -m pytest tests/test_gpu_pairs.py -x -q 2>&1 | tail -40
H2G_GO_FAST=0 timeout 900 python -m pytest tests/test_gpu_pairs.py -x -q 2>&1 | tail -3
} > $OUT/fast_test8.log 2>&1
cat $OUT/fast_test8.log
