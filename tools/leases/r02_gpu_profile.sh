#!/bin/bash
# One GPU-box visit: kernel-trace + PMC passes of the go() kernel on the 1 M-read E. coli-size SE workload (tools/align_perf.py).
# usage: tools/r02_gpu_profile.sh <tag> [nreads]   -> gpurun_out/r02_<tag>_{trace,pmc_*}.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
TAG=${1:-x}; N=${2:-1000000}; OUT=gpurun_out; mkdir -p $OUT
CMD="python tools/align_perf.py $N"
rocprofv3 --kernel-trace --stats -d /tmp/p_trace -- $CMD > $OUT/r02_${TAG}_run.log 2>&1
python tools/rocpd_summary.py /tmp/p_trace > $OUT/r02_${TAG}_trace.txt 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/p_pmc
  timeout 300 rocprofv3 --pmc $set -d /tmp/p_pmc -- $CMD > /tmp/p_pmc.log 2>&1
  echo "# --pmc $set" > $OUT/r02_${TAG}_pmc_$i.txt
  python tools/rocpd_summary.py /tmp/p_pmc 2>&1 | grep -E "^kernel|k_go|k_collect" >> $OUT/r02_${TAG}_pmc_$i.txt
done
