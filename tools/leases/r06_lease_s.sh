#!/bin/bash
# Round 6, lease S: the graph pass with its slow primitive forms on queues of their own (FQ_EXTEND_SLOW / FQ_WALK_SLOW): digests (graph cases of the fast-pass file + the graph suites),
# the steady step at 256 Mbp, the wave-level split (prof build, 4.9 Mbp graph)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_s; mkdir -p $OUT
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_fast_pass.py tests/test_gpu_fast_stress.py tests/test_gpu_batches.py -x -q > $OUT/gputests_fast.log 2>&1; tail -5 $OUT/gputests_fast.log | cut -c1-600
echo "tests after $(( $(date +%s) - T0 )) s"
H2G_LIB=$PWD/hisat2_amd/csrc/obj_prof/libh2g_prof.so timeout 600 python tools/fast_perf.py gpe 1000000 > $OUT/fast_prof_graph.log 2>&1; tail -16 $OUT/fast_prof_graph.log | cut -c1-1800
echo "prof after $(( $(date +%s) - T0 )) s"
S="8,128,0,0,512,64,-1;8,128,0,0,0,64,-1;8,128,0,0,512,64,-1"
timeout 900 python tools/queued_steps.py graph 256e6 1000000 "$S" > $OUT/graph.jsonl 2> $OUT/graph.err; cut -c1-400 $OUT/graph.jsonl; tail -3 $OUT/graph.err
echo "done after $(( $(date +%s) - T0 )) s"
