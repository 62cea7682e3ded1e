#!/bin/bash
# Round 6, lease B: the whole GPU suite on the sticky-lane loop + private GraphWS + the graph units' side count; wave-level time split of the new loop (prof build) on the
# 256 Mbp random genome and the 4.9 Mbp SNP graph.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_b; mkdir -p $OUT
T0=$(date +%s)
H2G_LIB=$PWD/hisat2_amd/csrc/obj_prof/libh2g_prof.so timeout 600 python tools/fast_perf.py pe 1000000 256e6 > $OUT/fast_prof_rnd256.log 2>&1; tail -22 $OUT/fast_prof_rnd256.log | cut -c1-700
H2G_LIB=$PWD/hisat2_amd/csrc/obj_prof/libh2g_prof.so timeout 600 python tools/fast_perf.py gpe 500000 > $OUT/fast_prof_graph.log 2>&1; tail -24 $OUT/fast_prof_graph.log | cut -c1-700
echo "prof after $(( $(date +%s) - T0 )) s"
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; tail -4 $OUT/gputests.log | cut -c1-300
echo "done after $(( $(date +%s) - T0 )) s"
