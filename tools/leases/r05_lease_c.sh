#!/bin/bash
# Round 5, lease C: defaults M = 8 / fast_reserve 0 / 16 hardware queues; overflowed reads listed by the pass itself (no partial rows); thin machine trips topped
# up from the other rings of the same primitive; records beyond 32 edits.  (1) the GPU suite, (2) the sweeps on the three legs, (3) the machine pass's time split.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r05_c; mkdir -p $OUT
T0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1; tail -6 $OUT/gputests.log | cut -c1-300
echo "tests after $(( $(date +%s) - T0 )) s"
timeout 700 python tools/r05_mstreams.py rep 256e6 1000000 "8,128,0,0;8,128,-1,0;8,96,0,0;4,128,0,0;8,192,0,0;6,96,0,0" > $OUT/mstreams_rep.jsonl 2> $OUT/mstreams_rep.err; cut -c1-330 $OUT/mstreams_rep.jsonl; tail -2 $OUT/mstreams_rep.err
echo "rep after $(( $(date +%s) - T0 )) s"
H2G_LIB=$PWD/hisat2_amd/csrc/obj_prof/libh2g_prof.so timeout 300 python tools/r05_mach_prof.py rep 256e6 1000000 > $OUT/mach_prof_rep.log 2>&1; tail -22 $OUT/mach_prof_rep.log | cut -c1-250
echo "prof after $(( $(date +%s) - T0 )) s"
timeout 240 python tools/r05_mstreams.py rnd 4.9e6 1000000 "8,128,0,0;2,96,-1,0;8,128,-1,0;4,128,0,0" > $OUT/mstreams_rnd.jsonl 2> $OUT/mstreams_rnd.err; cut -c1-330 $OUT/mstreams_rnd.jsonl
echo "rnd after $(( $(date +%s) - T0 )) s"
timeout 700 python tools/r05_mstreams.py graph 256e6 1000000 "8,128,0,0;8,128,0,1600;8,128,0,400;4,96,0,1600;2,96,-1,1600" > $OUT/mstreams_graph.jsonl 2> $OUT/mstreams_graph.err; cut -c1-330 $OUT/mstreams_graph.jsonl; tail -2 $OUT/mstreams_graph.err
echo "graph after $(( $(date +%s) - T0 )) s"
H2G_LIB=$PWD/hisat2_amd/csrc/obj_prof/libh2g_prof.so timeout 300 python tools/r05_mach_prof.py graph 256e6 1000000 > $OUT/mach_prof_graph.log 2>&1; tail -22 $OUT/mach_prof_graph.log | cut -c1-250
echo "done after $(( $(date +%s) - T0 )) s"
