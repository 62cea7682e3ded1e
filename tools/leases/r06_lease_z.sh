#!/bin/bash
# Round 6, lease Z: the index load through page-locked staging buffers filled by six threads, the three big arrays of the global index straight from the mapped files — the tests that
# read an index back through the device (rank / search / coordinates golden vectors, SAM), then the command line on a 1 Gbp index (1.6 GB on the device) with the load's own split
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_z; mkdir -p $OUT
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sam.py tests/test_gpu_align.py -x -q > $OUT/gputests.log 2>&1; tail -3 $OUT/gputests.log | cut -c1-300
echo "tests after $(( $(date +%s) - T0 )) s"
H2G_LOAD_TIMING=1 H2G_CLI_GENOME=1e9 timeout 1200 python tools/cli_perf.py 4000000 > $OUT/cli_1gbp.log 2>&1; tail -3 $OUT/cli_1gbp.log | cut -c1-700
echo "done after $(( $(date +%s) - T0 )) s"
