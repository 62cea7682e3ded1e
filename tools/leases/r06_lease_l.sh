#!/bin/bash
# Round 6, lease L: time-resolved lane fill / slots in flight of the fast pass (prof build: h2g_fast_prof.h FPROF_TBIN), 256 Mbp random genome and the 4.9 Mbp SNP graph at 1 M pairs
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_l; mkdir -p $OUT
T0=$(date +%s)
H2G_LIB=$PWD/hisat2_amd/csrc/obj_prof/libh2g_prof.so timeout 600 python tools/fast_perf.py pe 1000000 256e6 > $OUT/fast_prof_rnd256.log 2>&1; tail -26 $OUT/fast_prof_rnd256.log | cut -c1-1500
H2G_LIB=$PWD/hisat2_amd/csrc/obj_prof/libh2g_prof.so timeout 600 python tools/fast_perf.py gpe 1000000 > $OUT/fast_prof_graph.log 2>&1; tail -26 $OUT/fast_prof_graph.log | cut -c1-1500
echo "done after $(( $(date +%s) - T0 )) s"
