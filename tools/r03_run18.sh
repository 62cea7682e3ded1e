cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_fast_pass.py -x -q -m gpu -s > $OUT/r03_run18.log 2>&1
tail -15 $OUT/r03_run18.log
