#!/bin/bash
# (a) runtime knobs tail / align_mate: stress + timing at E. coli size; (b) FETCH_SIZE / WRITE_SIZE of the rank micro-kernels (calibration)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests
OUT=gpurun_out/r04_g9; mkdir -p $OUT
python - <<'PY' > $OUT/case.txt
import sys, tempfile
sys.path.insert(0, "tests")
import test_gpu_fast_stress as T
print(*T.hard_case(tempfile.mkdtemp(prefix="h2fs")))
PY
read BASE NPZ < $OUT/case.txt
for cfg in "0 0" "16 0" "0 1" "16 1"; do
  set -- $cfg; export H2G_FAST_TAIL=$1 H2G_FAST_AM=$2
  timeout 600 python tests/fast_stress.py $BASE $NPZ 4 all,0.37 > $OUT/stress_t$1_am$2.json 2> $OUT/stress_t$1_am$2.err; echo "stress tail=$1 am=$2 rc=$?"
  timeout 600 python tools/fast_perf.py pe 1000000 > $OUT/pe_t$1_am$2.log 2>&1; tail -1 $OUT/pe_t$1_am$2.log | cut -c1-330
done
unset H2G_FAST_TAIL H2G_FAST_AM
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rk_pmc
  rocprofv3 --pmc $c -d /tmp/rk_pmc -- python tools/rank_pmc.py 28 > $OUT/rank_pmc_$c.json 2> /tmp/rk_pmc.err
  echo "# rocprofv3 --pmc $c -- python tools/rank_pmc.py 28" > $OUT/rank_pmc_$c.txt
  python tools/rocpd_summary.py /tmp/rk_pmc >> $OUT/rank_pmc_$c.txt 2>&1
done
grep -E "k_rank" $OUT/rank_pmc_FETCH_SIZE.txt $OUT/rank_pmc_WRITE_SIZE.txt | cut -c1-200
cat $OUT/rank_pmc_FETCH_SIZE.json
