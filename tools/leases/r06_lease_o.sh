#!/bin/bash
# Round 6, lease O: what the fast kernel's waves do (SQ counters: parked / issuing / VALU) and the time-resolved lane fill with the drain launch on (prof build)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_o; mkdir -p $OUT
T0=$(date +%s)
H2G_LIB=$PWD/hisat2_amd/csrc/obj_prof/libh2g_prof.so timeout 600 python tools/fast_perf.py pe 1000000 256e6 > $OUT/fast_prof_rnd256.log 2>&1; tail -14 $OUT/fast_prof_rnd256.log | cut -c1-1800
echo "prof after $(( $(date +%s) - T0 )) s"
CMD="python tools/fast_perf.py pe 1000000 256e6"
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/bp_pmc
  timeout 400 rocprofv3 --pmc $set -d /tmp/bp_pmc -- $CMD > $OUT/pmc_run_$tag.log 2> /tmp/bp_pmc.err
  echo "# rocprofv3 --pmc $set -- $CMD" > $OUT/rnd_pmc_$tag.txt
  python tools/rocpd_summary.py /tmp/bp_pmc >> $OUT/rnd_pmc_$tag.txt 2>&1
  grep -E "k_go_fast" $OUT/rnd_pmc_$tag.txt | cut -c1-200; tail -2 /tmp/bp_pmc.err | cut -c1-300
done
echo "done after $(( $(date +%s) - T0 )) s"
