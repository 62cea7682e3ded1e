cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_zy_spliced.py tests/test_gpu_zx_chr22_spliced.py tests/test_gpu_zz_graph_spliced.py tests/test_gpu_sam.py -x -q -m gpu > $OUT/r03_run21.log 2>&1
tail -8 $OUT/r03_run21.log
