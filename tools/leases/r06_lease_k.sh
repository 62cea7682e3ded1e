#!/bin/bash
# Round 6, lease K: the queued-run stress test aborted once in the full suite of lease H (SIGABRT inside h2g_align_fetch_dense, no message): how often, and what the runtime says
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_k; mkdir -p $OUT
T0=$(date +%s)
for i in $(seq 1 14); do
  AMD_LOG_LEVEL=1 timeout 300 python -m pytest tests/test_gpu_fast_stress.py -x -q > $OUT/stress_$i.log 2>&1; rc=$?
  echo "run $i rc $rc $(tail -1 $OUT/stress_$i.log | cut -c1-120) after $(( $(date +%s) - T0 )) s"
  if [ $rc -ne 0 ]; then grep -v "^  File\|^$" $OUT/stress_$i.log | head -30 | cut -c1-300; fi
done
