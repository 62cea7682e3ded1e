#!/bin/bash
# graph fast pass: device equality (fast on == off) and timing on the bench's graph leg workload
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD:$PWD/tests
timeout 1200 python -m pytest tests/test_gpu_fast_pass.py -x -q > gpurun_out/r04_g3_tests.log 2>&1; tail -5 gpurun_out/r04_g3_tests.log
for f in 1 0; do H2G_GO_FAST=$f timeout 600 python tools/fast_perf.py gpe 500000 > gpurun_out/r04_g3_gpe_fast$f.log 2>&1; tail -2 gpurun_out/r04_g3_gpe_fast$f.log; done
