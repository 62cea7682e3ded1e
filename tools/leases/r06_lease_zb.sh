#!/bin/bash
# Round 6, lease ZB: the command line with the two mate files parsed side by side and without the walk through the destructors at exit: the whole GPU suite, then 4 M pairs on a 1 Gbp index
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_zb; mkdir -p $OUT
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; tail -4 $OUT/gputests.log | cut -c1-300
echo "tests after $(( $(date +%s) - T0 )) s"
H2G_LOAD_TIMING=1 H2G_CLI_GENOME=1e9 timeout 1200 python tools/cli_perf.py 4000000 > $OUT/cli_1gbp.log 2>&1; tail -2 $OUT/cli_1gbp.log | cut -c1-700
echo "done after $(( $(date +%s) - T0 )) s"
