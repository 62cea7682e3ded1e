#!/bin/bash
# Round 6, lease Y: the Occ-rank micro-kernels with the synthetic query's row drawn by one multiply-high instead of a 64-bit modulo (the generator was 100 of a query's 340 instructions)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_y; mkdir -p $OUT
T0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "rank" > $OUT/gputests_rank.log 2>&1; tail -3 $OUT/gputests_rank.log | cut -c1-300
timeout 600 python tools/rank_variants.py 28 > $OUT/rank_variants.json 2> $OUT/rank_variants.err; cat $OUT/rank_variants.json | tr -d '\n' | cut -c1-1800; echo; tail -2 $OUT/rank_variants.err
timeout 300 python tools/chain_bench.py $((1<<21)) 64 none compact 2>/dev/null | tail -1 | cut -c1-600
echo "done after $(( $(date +%s) - T0 )) s"
