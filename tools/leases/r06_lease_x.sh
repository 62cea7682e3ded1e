#!/bin/bash
# Round 6, lease X: mapGLF of a row range from sides held in registers (map_glf_fused: two rounds of paired side loads instead of eight dependent requests) in the graph searches —
# digests of the graph cases, then A / B against the library before it (hisat2_amd/variants/libh2g_prefuse.so), 256 Mbp SNP graph, one box
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_x; mkdir -p $OUT
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_fast_pass.py tests/test_gpu_parity.py -x -q -k "graph or case2 or case3 or case9 or fast" > $OUT/gputests.log 2>&1; tail -3 $OUT/gputests.log | cut -c1-400
echo "tests after $(( $(date +%s) - T0 )) s"
for lib in prefuse new prefuse new; do
  if [ $lib = new ]; then unset H2G_LIB; else export H2G_LIB=$PWD/hisat2_amd/variants/libh2g_$lib.so; fi
  timeout 600 python tools/queued_steps.py graph 256e6 1000000 "8,128,0,0,-1,64,-1" >> $OUT/graph_$lib.jsonl 2> $OUT/graph_$lib.err; echo "graph $lib: $(tail -1 $OUT/graph_$lib.jsonl | cut -c1-420)"
done
echo "done after $(( $(date +%s) - T0 )) s"
