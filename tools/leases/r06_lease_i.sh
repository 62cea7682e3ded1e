#!/bin/bash
# Round 6, lease I: what a resident batch's first run costs (bench.py's 20-step window reads 18 ms per step where the steady state is 12-13)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_i; mkdir -p $OUT
timeout 900 python tools/batches_first.py 256e6 1000000 10 > $OUT/batches_first.jsonl 2> $OUT/batches_first.err; cut -c1-200 $OUT/batches_first.jsonl; tail -3 $OUT/batches_first.err
