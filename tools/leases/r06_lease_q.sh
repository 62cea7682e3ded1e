#!/bin/bash
# Round 6, lease Q: pairs that need alignMate parked for the drain launch (the alignMate build behind the plain fast launch): fast == machine digests, then steady steps with it off / on
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_q; mkdir -p $OUT
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_fast_pass.py -x -q > $OUT/gputests_fast.log 2>&1; tail -5 $OUT/gputests_fast.log | cut -c1-600
echo "tests after $(( $(date +%s) - T0 )) s"
S="8,128,0,0,512,32,0;8,128,0,0,512,32,1;8,128,0,0,512,48,1;8,128,0,0,0,32,0;8,128,0,0,512,32,1;8,128,0,0,512,32,0"
timeout 900 python tools/queued_steps.py rnd 256e6 1000000 "$S" > $OUT/rnd.jsonl 2> $OUT/rnd.err; cut -c1-360 $OUT/rnd.jsonl; tail -3 $OUT/rnd.err
echo "rnd after $(( $(date +%s) - T0 )) s"
S="8,128,0,0,512,32,0;8,128,0,0,512,32,1;8,128,0,0,512,64,1;8,128,0,0,0,32,0"
timeout 900 python tools/queued_steps.py rep 256e6 1000000 "$S" > $OUT/rep.jsonl 2> $OUT/rep.err; cut -c1-360 $OUT/rep.jsonl; tail -3 $OUT/rep.err
echo "done after $(( $(date +%s) - T0 )) s"
