#!/bin/bash
# Round 6, lease M: the end of a batch through the drain launch (k_go_fast_drain): fast == machine digests incl. the new cases, then the steady step with the drain launch
# off / on at several thresholds and grids, one process per index (256 Mbp random genome, 256 Mbp SNP graph, 256 Mbp repeat-structured genome)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_m; mkdir -p $OUT
T0=$(date +%s)
timeout 1200 python -m pytest tests/test_gpu_fast_pass.py -x -q > $OUT/gputests_fast.log 2>&1; tail -5 $OUT/gputests_fast.log | cut -c1-400
echo "tests after $(( $(date +%s) - T0 )) s"
S="8,128,0,0,0;8,128,0,0,512,32;8,128,0,0,512,16;8,128,0,0,512,64;8,128,0,0,256,32;8,128,0,0,768,48;8,128,0,0,0"
timeout 900 python tools/queued_steps.py rnd 256e6 1000000 "$S" > $OUT/rnd.jsonl 2> $OUT/rnd.err; cut -c1-330 $OUT/rnd.jsonl; tail -3 $OUT/rnd.err
echo "rnd after $(( $(date +%s) - T0 )) s"
timeout 900 python tools/queued_steps.py graph 256e6 1000000 "$S" > $OUT/graph.jsonl 2> $OUT/graph.err; cut -c1-330 $OUT/graph.jsonl; tail -3 $OUT/graph.err
echo "graph after $(( $(date +%s) - T0 )) s"
timeout 900 python tools/queued_steps.py rep 256e6 1000000 "8,128,0,0,0;8,128,0,0,512,32;8,128,0,0,512,64" > $OUT/rep.jsonl 2> $OUT/rep.err; cut -c1-330 $OUT/rep.jsonl; tail -3 $OUT/rep.err
echo "done after $(( $(date +%s) - T0 )) s"
