#!/bin/bash
# Round 6, lease ZA: the command line with formatting on a thread of its own (fetch of batch k + 1 beside the text of batch k) and the staged index load: the whole GPU suite (every
# command-line test compares SAM with the reference's), then 10 M pairs on the GRCh38-size index (built here), file and /dev/null, with the load's split
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_za; mkdir -p $OUT
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; tail -4 $OUT/gputests.log | cut -c1-300
echo "tests after $(( $(date +%s) - T0 )) s"
H2G_LOAD_TIMING=1 H2G_CLI_GENOME=3.1e9 timeout 2400 python tools/cli_perf.py 10000000 > $OUT/cli_10M_pairs.log 2>&1; tail -3 $OUT/cli_10M_pairs.log | cut -c1-800
echo "cli after $(( $(date +%s) - T0 )) s"
H2G_LOAD_TIMING=1 H2G_CLI_GENOME=3.1e9 timeout 600 python tools/cli_perf.py 1000000 > $OUT/cli_1M_pairs.log 2>&1; tail -2 $OUT/cli_1M_pairs.log | cut -c1-800
echo "done after $(( $(date +%s) - T0 )) s"
