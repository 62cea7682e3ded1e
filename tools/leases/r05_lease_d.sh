#!/bin/bash
# Round 5, lease D: (1) which pairs differ in test_gpu_fast_pass[case5] and how; (2) the graph primitives' per-lane scratch in private memory (obj_gp/libh2g_gp.so) against
# the shipped library on a 32 Mbp SNP graph; (3) trace + PMC of the spliced unit.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r05_d; mkdir -p $OUT
T0=$(date +%s)
timeout 400 python tools/r05_case5_debug.py > $OUT/case5_debug.log 2>&1; cut -c1-600 $OUT/case5_debug.log | tail -60
echo "case5 after $(( $(date +%s) - T0 )) s"
timeout 400 python tools/r05_mstreams.py graph 32e6 1000000 "8,128,0,0" > $OUT/graph32_shipped.jsonl 2> $OUT/graph32_shipped.err; cut -c1-330 $OUT/graph32_shipped.jsonl; tail -2 $OUT/graph32_shipped.err
H2G_LIB=$PWD/hisat2_amd/csrc/obj_gp/libh2g_gp.so timeout 400 python tools/r05_mstreams.py graph 32e6 1000000 "8,128,0,0" > $OUT/graph32_gp.jsonl 2> $OUT/graph32_gp.err; cut -c1-330 $OUT/graph32_gp.jsonl; tail -2 $OUT/graph32_gp.err
H2G_LIB=$PWD/hisat2_amd/csrc/obj_gp/libh2g_gp.so timeout 300 python -m pytest tests/test_gpu_fast_pass.py -q -k "case2 or case3" > $OUT/gp_fastpass.log 2>&1; tail -2 $OUT/gp_fastpass.log
echo "gp after $(( $(date +%s) - T0 )) s"
CMD="python bench.py --only-legs spliced_pe --no-cpu-baseline"
rm -rf /tmp/bp_trace
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/bp_trace -- $CMD > $OUT/spliced_traced.json 2> /tmp/bp_trace.err
python tools/rocpd_summary.py /tmp/bp_trace > $OUT/spliced_trace.txt 2>&1; head -6 $OUT/spliced_trace.txt | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/bp_pmc
  timeout 300 rocprofv3 --pmc $c -d /tmp/bp_pmc -- $CMD > /dev/null 2> /tmp/bp_pmc.err
  echo "# rocprofv3 --pmc $c -- $CMD" > $OUT/spliced_pmc_$c.txt
  python tools/rocpd_summary.py /tmp/bp_pmc >> $OUT/spliced_pmc_$c.txt 2>&1
done
grep -E "k_go" $OUT/spliced_pmc_FETCH_SIZE.txt $OUT/spliced_pmc_WRITE_SIZE.txt | grep SIZE | cut -c1-220
head -c 1500 $OUT/spliced_traced.json; echo
echo "done after $(( $(date +%s) - T0 )) s"
