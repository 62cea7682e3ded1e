#!/bin/bash
# Round 6, lease N: the drain launch on a stream of its own (the graph unit's 5.6 KB of scratch per lane on eight more queues was refused by the runtime in lease M), and how many
# CUs the fast launch leaves it: steady step by (orphan threshold, drain grid, fast_reserve), 256 Mbp random genome / SNP graph
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_n; mkdir -p $OUT
T0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_fast_pass.py -x -q -k "case7 or case8 or case9 or case10" > $OUT/gputests_fast.log 2>&1; tail -3 $OUT/gputests_fast.log | cut -c1-400
echo "tests after $(( $(date +%s) - T0 )) s"
S="8,128,0,0,0;8,128,0,0,512,32;8,128,32,0,512,32;8,128,48,0,512,32;8,128,64,0,512,32;8,128,64,0,512,48;8,128,80,0,512,64;8,128,48,0,256,32;8,128,32,0,0;8,128,0,0,0"
timeout 900 python tools/queued_steps.py rnd 256e6 1000000 "$S" > $OUT/rnd.jsonl 2> $OUT/rnd.err; cut -c1-330 $OUT/rnd.jsonl; tail -3 $OUT/rnd.err
echo "rnd after $(( $(date +%s) - T0 )) s"
S="8,128,0,0,0;8,128,0,0,512,32;8,128,32,0,512,32;8,128,48,0,512,32;8,128,64,0,512,48;8,128,32,0,256,32"
timeout 900 python tools/queued_steps.py graph 256e6 1000000 "$S" > $OUT/graph.jsonl 2> $OUT/graph.err; cut -c1-330 $OUT/graph.jsonl; tail -3 $OUT/graph.err
echo "done after $(( $(date +%s) - T0 )) s"
