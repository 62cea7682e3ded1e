#!/bin/bash
# Round 6, lease A: the fast pass with reads that stay in their lane (sticky lanes) and the graph primitives' scratch in private memory — on == off digests, then
# A/B steps against the round-5 library (hisat2_amd/variants/libh2g_r5base.so) on one box: random 256 Mbp linear, SNP graph 256 Mbp; FETCH/WRITE of the graph kernel.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_a; mkdir -p $OUT
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_fast_pass.py tests/test_gpu_fast_stress.py -m gpu -x -q > $OUT/gputests_fast.log 2>&1; tail -4 $OUT/gputests_fast.log | cut -c1-300
echo "tests after $(( $(date +%s) - T0 )) s"
for lib in r5base gwspriv new; do
  if [ $lib = new ]; then unset H2G_LIB; else export H2G_LIB=$PWD/hisat2_amd/variants/libh2g_$lib.so; fi
  timeout 600 python tools/queued_steps.py graph 256e6 1000000 "8,128,0,0" > $OUT/graph_$lib.jsonl 2> $OUT/graph_$lib.err; echo "graph $lib: $(tail -1 $OUT/graph_$lib.jsonl | cut -c1-420)"
done
echo "graph after $(( $(date +%s) - T0 )) s"
for lib in r5base new; do
  if [ $lib = new ]; then unset H2G_LIB; else export H2G_LIB=$PWD/hisat2_amd/variants/libh2g_$lib.so; fi
  timeout 600 python tools/queued_steps.py rnd 256e6 1000000 "8,128,0,0" > $OUT/rnd_$lib.jsonl 2> $OUT/rnd_$lib.err; echo "rnd $lib: $(tail -1 $OUT/rnd_$lib.jsonl | cut -c1-420)"
done
unset H2G_LIB
echo "rnd after $(( $(date +%s) - T0 )) s"
for leg in graph rnd; do
  CMD="python tools/queued_steps.py $leg 256e6 1000000 8,128,0,0"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/bp_pmc
    timeout 400 rocprofv3 --pmc $c -d /tmp/bp_pmc -- $CMD > $OUT/${leg}_pmc_run.jsonl 2> /tmp/bp_pmc.err
    echo "# rocprofv3 --pmc $c -- $CMD" > $OUT/${leg}_pmc_$c.txt
    python tools/rocpd_summary.py /tmp/bp_pmc >> $OUT/${leg}_pmc_$c.txt 2>&1
  done
  grep -E "k_go_fast" $OUT/${leg}_pmc_FETCH_SIZE.txt $OUT/${leg}_pmc_WRITE_SIZE.txt | grep SIZE | cut -c1-220
done
echo "done after $(( $(date +%s) - T0 )) s"
