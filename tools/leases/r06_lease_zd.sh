#!/bin/bash
# Round 6, lease ZD: the command line's parser with table-driven base conversion (no push_back per character): the command-line tests, then 4 M pairs on a 1 Gbp index, three runs
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_zd; mkdir -p $OUT
T0=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_sam.py tests/test_gpu_chr22.py tests/test_gpu_long_edits.py tests/test_gpu_zy_spliced.py -x -q > $OUT/gputests.log 2>&1; tail -3 $OUT/gputests.log | cut -c1-300
echo "tests after $(( $(date +%s) - T0 )) s"
for k in 1 2 3; do H2G_CLI_GENOME=1e9 timeout 1200 python tools/cli_perf.py 4000000 2>&1 | tail -2 | cut -c1-520; done > $OUT/cli_1gbp.log; cat $OUT/cli_1gbp.log
echo "done after $(( $(date +%s) - T0 )) s"
