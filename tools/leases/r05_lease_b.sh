#!/bin/bash
# Round 5, lease B: (1) is the fast==machine digest inequality of lease A (k_go_fast_am + tail on the repeat genome) a function of the pipeline depth?
# (2) the rest of the GPU suite past it; (3) the depth sweep again with more hardware queues than streams (GPU_MAX_HW_QUEUES: ROCclr maps streams onto
# 4 hardware queues by default — 1 + M streams beyond that share queues and serialise).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r05_b; mkdir -p $OUT
T0=$(date +%s)
for M in 2 4; do
  H2G_MSTREAMS=$M timeout 600 python -m pytest "tests/test_gpu_fast_pass.py" -q -k "case5 or case6 or case4" > $OUT/fastpass_M$M.log 2>&1; echo "M=$M: $(tail -1 $OUT/fastpass_M$M.log)"
done
GPU_MAX_HW_QUEUES=16 H2G_MSTREAMS=4 timeout 600 python -m pytest "tests/test_gpu_fast_pass.py" -q -k "case5" > $OUT/fastpass_M4_q16.log 2>&1; echo "M=4 q16: $(tail -1 $OUT/fastpass_M4_q16.log)"
echo "fastpass after $(( $(date +%s) - T0 )) s"
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fast_pass.py > $OUT/gputests_rest.log 2>&1; tail -4 $OUT/gputests_rest.log
echo "tests after $(( $(date +%s) - T0 )) s"
GPU_MAX_HW_QUEUES=16 timeout 700 python tools/r05_mstreams.py rep 256e6 1000000 "2,96,-1,0;4,96,-1,0;6,96,-1,0;8,96,-1,0;8,128,-1,0;4,128,-1,0" > $OUT/mstreams_rep_q16.jsonl 2> $OUT/mstreams_rep_q16.err; cut -c1-330 $OUT/mstreams_rep_q16.jsonl; tail -2 $OUT/mstreams_rep_q16.err
echo "rep after $(( $(date +%s) - T0 )) s"
GPU_MAX_HW_QUEUES=16 timeout 240 python tools/r05_mstreams.py rnd 4.9e6 1000000 "2,96,-1,0;4,96,-1,0;4,96,0,0" > $OUT/mstreams_rnd_q16.jsonl 2> $OUT/mstreams_rnd_q16.err; cut -c1-330 $OUT/mstreams_rnd_q16.jsonl
echo "done after $(( $(date +%s) - T0 )) s"
