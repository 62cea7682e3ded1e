#!/bin/bash
# experiment build: the graph units with the per-lane scratch of the graph primitives (GraphWS) in PRIVATE memory (-DH2G_GWS_PRIVATE=1): hisat2_amd/csrc/obj_gp/libh2g_gp.so,
# loaded through H2G_LIB.  Everything else is the shipped library's objects (run `make -C hisat2_amd/csrc` first).
set -e
cd "$(dirname "$0")/../hisat2_amd/csrc"
mkdir -p obj_gp
F3="-DH2G_GWS_PRIVATE=1 --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage"
F2="${F3/-O3/-O2 -fno-slp-vectorize -mllvm -sink-insts-to-avoid-spills=1}"
( /opt/rocm/bin/hipcc $F2 -c -o obj_gp/h2g_k_go_fast_graph.o h2g_k_go_fast_graph.hip 2> obj_gp/h2g_k_go_fast_graph.log || { grep error obj_gp/h2g_k_go_fast_graph.log; exit 1; } ) &
for u in h2g_k_go_graph h2g_k_go_graph_spl; do ( /opt/rocm/bin/hipcc $F3 -c -o obj_gp/$u.o $u.hip 2> obj_gp/$u.log || { grep error obj_gp/$u.log; exit 1; } ) & done
wait
KEEP="h2g_kernels h2g_k_go_fast h2g_k_go_fast_am h2g_k_go_linear h2g_k_go_linear_big h2g_k_go_graph_big h2g_k_go_linear_spl h2g_k_go_linear_spl_big h2g_k_go_graph_spl_big h2g_sam"
OBJS=""; for u in $KEEP; do OBJS="$OBJS obj/$u.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o obj_gp/libh2g_gp.so $OBJS obj_gp/h2g_k_go_fast_graph.o obj_gp/h2g_k_go_graph.o obj_gp/h2g_k_go_graph_spl.o
grep -h "ScratchSize" obj_gp/*.log | sort | uniq -c
ls -la obj_gp/libh2g_gp.so
