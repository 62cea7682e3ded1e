set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in pe se; do for f in 0 1; do H2G_GO_FAST=$f timeout 300 python tools/fast_perf.py $m 1000000 2>&1 | tail -2; done; done > gpurun_out/fast_perf1.log 2>&1
cat gpurun_out/fast_perf1.log
timeout 600 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_align.py -x -q 2>&1 | tail -5
