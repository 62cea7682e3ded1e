#!/bin/bash
# Round 6, lease T: (i) orphan threshold / drain grid with the alignMate pairs in the drain launch, 256 Mbp random genome; (ii) the graph unit with partial_search_graph_item inline
# (obj/libh2g_psginl.so) against the shipped one, 256 Mbp SNP graph — one box
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_t; mkdir -p $OUT
T0=$(date +%s)
S="8,128,0,0,512,64,1;8,128,0,0,384,64,1;8,128,0,0,640,64,1;8,128,0,0,512,96,1;8,128,0,0,384,48,1;8,128,0,0,768,96,1;8,128,0,0,512,64,1;8,128,0,0,0,64,0"
timeout 900 python tools/queued_steps.py rnd 256e6 1000000 "$S" > $OUT/rnd.jsonl 2> $OUT/rnd.err; cut -c1-330 $OUT/rnd.jsonl; tail -3 $OUT/rnd.err
echo "rnd after $(( $(date +%s) - T0 )) s"
for lib in shipped psginl shipped psginl; do
  if [ $lib = shipped ]; then unset H2G_LIB; else export H2G_LIB=$PWD/hisat2_amd/csrc/obj/libh2g_$lib.so; fi
  timeout 600 python tools/queued_steps.py graph 256e6 1000000 "8,128,0,0,-1,64,-1" >> $OUT/graph_$lib.jsonl 2> $OUT/graph_$lib.err; echo "graph $lib: $(tail -1 $OUT/graph_$lib.jsonl | cut -c1-420)"
done
echo "done after $(( $(date +%s) - T0 )) s"
