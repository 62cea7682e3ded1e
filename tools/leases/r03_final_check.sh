cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fast_pass.py tests/test_gpu_pairs.py tests/test_gpu_align.py tests/test_gpu_sam.py tests/test_gpu_chr22.py -x -q -m gpu > $OUT/r03_final_tests.log 2>&1
tail -3 $OUT/r03_final_tests.log
timeout 600 python tools/cli_perf.py 10000000 > $OUT/r03_cli_10M.log 2>&1
cat $OUT/r03_cli_10M.log
