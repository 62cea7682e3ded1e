#!/bin/bash
# Round 6, lease U: the end of the machine's pass through a drain launch of the same kernel (reads taken up in place).  (1) the WHOLE GPU suite with every new path forced on small batches
# (H2G_MACH_ORPHAN / H2G_FAST_ORPHAN ...: every parity and equality test then runs through them), (2) the suite on the defaults, (3) steady steps with the machine's drain off / on:
# repeat-structured, random, SNP graph at 256 Mbp
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_u; mkdir -p $OUT
T0=$(date +%s)
H2G_MACH_ORPHAN=48 H2G_MACH_DRAIN_DIV=2 H2G_FAST_ORPHAN=64 H2G_DRAIN_GRID=8 timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/gputests_forced.log 2>&1; tail -4 $OUT/gputests_forced.log | cut -c1-400
echo "forced suite after $(( $(date +%s) - T0 )) s"
timeout 1500 python -m pytest tests/test_gpu_fast_pass.py tests/test_gpu_fast_stress.py -x -q > $OUT/gputests_fast.log 2>&1; tail -3 $OUT/gputests_fast.log | cut -c1-400
echo "default fast tests after $(( $(date +%s) - T0 )) s"
S="8,128,0,0,-1,64,-1,0;8,128,0,0,-1,64,-1,256,4;8,128,0,0,-1,64,-1,128,4;8,128,0,0,-1,64,-1,256,8;8,128,0,0,-1,64,-1,512,4;8,128,0,0,-1,64,-1,0"
timeout 900 python tools/queued_steps.py rep 256e6 1000000 "$S" > $OUT/rep.jsonl 2> $OUT/rep.err; cut -c1-420 $OUT/rep.jsonl; tail -3 $OUT/rep.err
echo "rep after $(( $(date +%s) - T0 )) s"
S="8,128,0,0,-1,64,-1,0;8,128,0,0,-1,64,-1,256,4;8,128,0,0,-1,64,-1,128,4;8,128,0,0,-1,64,-1,0"
timeout 900 python tools/queued_steps.py rnd 256e6 1000000 "$S" > $OUT/rnd.jsonl 2> $OUT/rnd.err; cut -c1-420 $OUT/rnd.jsonl; tail -3 $OUT/rnd.err
timeout 900 python tools/queued_steps.py graph 256e6 1000000 "$S" > $OUT/graph.jsonl 2> $OUT/graph.err; cut -c1-420 $OUT/graph.jsonl; tail -3 $OUT/graph.err
echo "done after $(( $(date +%s) - T0 )) s"
