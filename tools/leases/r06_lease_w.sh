#!/bin/bash
# Round 6, lease W: the machine's drain launch once more, where the passes are bound by the CUs they hold and not by their latency (lease V: mach_total 192-384) — off / on per share
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_w; mkdir -p $OUT
T0=$(date +%s)
S="8,0,0,0,-1,64,-1,0;8,0,0,0,-1,64,-1,256,4;8,0,0,0,-1,64,-1,256,2;8,384,0,0,-1,64,-1,0;8,384,0,0,-1,64,-1,256,4;8,384,0,0,-1,64,-1,256,2;8,384,0,0,-1,64,-1,512,2;8,256,0,0,-1,64,-1,0;8,256,0,0,-1,64,-1,256,2;8,0,0,0,-1,64,-1,0"
timeout 900 python tools/queued_steps.py rep 256e6 1000000 "$S" > $OUT/rep.jsonl 2> $OUT/rep.err; cut -c1-400 $OUT/rep.jsonl; tail -3 $OUT/rep.err
echo "rep after $(( $(date +%s) - T0 )) s"
