cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputests.log 2>&1
tail -15 gpurun_out/r03_gputests.log
