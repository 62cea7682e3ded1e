#!/bin/bash
# PMC passes over the fast go() pass (PE, 1 M pairs, E. coli-size index): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQC?_[A-Z0-9_]+)\b" | sort -u | tr '\n' ' ' > $OUT/r03_counters_sq.txt
CMD="python tools/fast_perf.py pe 1000000"
i=0
for set in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
  "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INST_CYCLES_SALU SQ_INSTS_FLAT SQ_INSTS_GDS" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" \
  "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/p_pmc
  timeout 200 rocprofv3 --pmc $set -d /tmp/p_pmc -- $CMD > /tmp/p_pmc.log 2>&1
  echo "# --pmc $set" > $OUT/r03_fastpmc_$i.txt
  python tools/rocpd_summary.py /tmp/p_pmc 2>&1 | grep -E "^kernel|k_go" >> $OUT/r03_fastpmc_$i.txt
  tail -3 /tmp/p_pmc.log >> $OUT/r03_fastpmc_$i.txt
done
cat $OUT/r03_fastpmc_*.txt
