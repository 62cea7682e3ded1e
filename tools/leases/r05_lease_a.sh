#!/bin/bash
# Round 5, lease A (no GRCh38-size build): the GPU suite on the M-deep machine pipeline, the pipeline-depth sweep on the three legs whose step the
# hand-on path bounds (repeat-structured 256 Mbp, SNP graph 256 Mbp, E. coli-size random), and the first trace + PMC of the graph / repeat legs.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r05_a; mkdir -p $OUT
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; tail -4 $OUT/gputests.log
echo "tests after $(( $(date +%s) - T0 )) s"
timeout 240 python tools/r05_mstreams.py rnd 4.9e6 1000000 "2,96,-1,0;4,96,-1,0;4,48,-1,0;4,96,0,0;6,96,-1,0" > $OUT/mstreams_rnd.jsonl 2> $OUT/mstreams_rnd.err; cut -c1-330 $OUT/mstreams_rnd.jsonl
echo "rnd after $(( $(date +%s) - T0 )) s"
timeout 700 python tools/r05_mstreams.py rep 256e6 1000000 > $OUT/mstreams_rep.jsonl 2> $OUT/mstreams_rep.err; cut -c1-330 $OUT/mstreams_rep.jsonl; tail -2 $OUT/mstreams_rep.err
echo "rep after $(( $(date +%s) - T0 )) s"
timeout 700 python tools/r05_mstreams.py graph 256e6 1000000 "2,96,-1,1600;4,96,-1,1600;4,48,-1,1600;4,96,-1,400;6,96,-1,1600;4,96,0,1600" > $OUT/mstreams_graph.jsonl 2> $OUT/mstreams_graph.err; cut -c1-330 $OUT/mstreams_graph.jsonl; tail -2 $OUT/mstreams_graph.err
echo "graph after $(( $(date +%s) - T0 )) s"
for leg in graph rep; do
  CMD="python tools/r05_mstreams.py $leg 256e6 1000000 4,96,-1,0"
  rm -rf /tmp/bp_trace
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/bp_trace -- $CMD > $OUT/${leg}_traced.jsonl 2> /tmp/bp_trace.err
  python tools/rocpd_summary.py /tmp/bp_trace > $OUT/${leg}_trace.txt 2>&1; head -8 $OUT/${leg}_trace.txt | cut -c1-200
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/bp_pmc
    timeout 400 rocprofv3 --pmc $c -d /tmp/bp_pmc -- $CMD > /dev/null 2> /tmp/bp_pmc.err
    echo "# rocprofv3 --pmc $c -- $CMD" > $OUT/${leg}_pmc_$c.txt
    python tools/rocpd_summary.py /tmp/bp_pmc >> $OUT/${leg}_pmc_$c.txt 2>&1
  done
  grep -E "k_go" $OUT/${leg}_pmc_FETCH_SIZE.txt $OUT/${leg}_pmc_WRITE_SIZE.txt | cut -c1-220
  echo "$leg profiles after $(( $(date +%s) - T0 )) s"
done
echo "done after $(( $(date +%s) - T0 )) s"
