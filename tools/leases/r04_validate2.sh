#!/bin/bash
# the second pass on a stream of its own: device equality (incl. repeat-genome cases with second passes), stress, timing on repeat-structured sequence
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests
timeout 1500 python -m pytest tests/test_gpu_fast_pass.py tests/test_gpu_fast_stress.py tests/test_gpu_pairs.py tests/test_gpu_chr22.py -x -q > gpurun_out/r04_validate2_tests.log 2>&1; tail -3 gpurun_out/r04_validate2_tests.log
timeout 600 python tools/fast_perf.py rpe 1000000 40e6 2>&1 | tail -1 | cut -c1-360
python bench.py --only-legs repeat_pe > gpurun_out/r04_legs_repeat2.json 2> gpurun_out/r04_legs_repeat2.err; head -c 1500 gpurun_out/r04_legs_repeat2.json; echo
