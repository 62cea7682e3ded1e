cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
{
H2G_LIB=$PWD/hisat2_amd/libh2g_v1.so timeout 300 python tools/fast_perf.py pe 1000000 2>&1 | tail -16
CMD="python tools/fast_perf.py pe 1000000"
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  rm -rf /tmp/p_pmc
  timeout 200 rocprofv3 --pmc $set -d /tmp/p_pmc -- $CMD > /tmp/p_pmc.log 2>&1
  python tools/rocpd_summary.py /tmp/p_pmc 2>&1 | grep -E "k_go_fast"
done
} > $OUT/fast_perf4.log 2>&1
cat $OUT/fast_perf4.log
