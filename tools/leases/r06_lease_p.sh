#!/bin/bash
# Round 6, lease P: what the GRAPH fast kernel's waves do (SQ counters), 256 Mbp SNP graph, 1 M pairs; and its wave-level split with the drain launch on (prof build, 4.9 Mbp graph)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_p; mkdir -p $OUT
T0=$(date +%s)
CMD="python tools/queued_steps.py graph 256e6 1000000 8,128,0,0,512,32"
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/bp_pmc
  timeout 600 rocprofv3 --pmc $set -d /tmp/bp_pmc -- $CMD > $OUT/pmc_run_$tag.log 2> /tmp/bp_pmc.err
  echo "# rocprofv3 --pmc $set -- $CMD" > $OUT/graph_pmc_$tag.txt
  python tools/rocpd_summary.py /tmp/bp_pmc >> $OUT/graph_pmc_$tag.txt 2>&1
  grep -E "k_go_fast|k_go<" $OUT/graph_pmc_$tag.txt | cut -c1-200; tail -2 /tmp/bp_pmc.err | cut -c1-300
  echo "$tag after $(( $(date +%s) - T0 )) s"
done
H2G_LIB=$PWD/hisat2_amd/csrc/obj_prof/libh2g_prof.so timeout 600 python tools/fast_perf.py gpe 1000000 > $OUT/fast_prof_graph.log 2>&1; tail -16 $OUT/fast_prof_graph.log | cut -c1-1800
echo "done after $(( $(date +%s) - T0 )) s"
