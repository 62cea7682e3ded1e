#!/bin/bash
# the library with the wave-level time split compiled in (-DH2G_GO_PROF): hisat2_amd/csrc/obj_prof/libh2g_prof.so, loaded through H2G_LIB (never the shipped library)
set -e
cd "$(dirname "$0")/../hisat2_amd/csrc"
mkdir -p obj_prof
UNITS="h2g_kernels h2g_k_go_fast h2g_k_go_fast_am h2g_k_go_fast_graph h2g_k_go_linear h2g_k_go_graph h2g_k_go_linear_big h2g_k_go_graph_big h2g_k_go_linear_spl h2g_k_go_linear_spl_big h2g_k_go_graph_spl h2g_k_go_graph_spl_big"
FLAGS="-DH2G_GO_PROF --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function"
for u in $UNITS; do
  f="$FLAGS"
  case $u in h2g_k_go_fast*) f="${FLAGS/-O3/-O2 -fno-slp-vectorize -mllvm -sink-insts-to-avoid-spills=1}";; esac
  ( /opt/rocm/bin/hipcc $f -c -o obj_prof/$u.o $u.hip 2> obj_prof/$u.log || { cat obj_prof/$u.log; exit 1; } ) &
done
wait
/opt/rocm/bin/hipcc -x c++ -O2 -std=c++17 -fPIC -fvisibility=hidden -Wall -c -o obj_prof/h2g_sam.o h2g_sam.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o obj_prof/libh2g_prof.so obj_prof/*.o
ls -la obj_prof/libh2g_prof.so
