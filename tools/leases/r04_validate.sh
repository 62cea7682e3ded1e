#!/bin/bash
# final-build validation on small indexes: device equality cases, the queued-run stress, the repeat-structured leg, the graph leg's workload
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests
timeout 1500 python -m pytest tests/test_gpu_fast_pass.py tests/test_gpu_fast_stress.py -x -q > gpurun_out/r04_validate_tests.log 2>&1; tail -4 gpurun_out/r04_validate_tests.log
python bench.py --only-legs repeat_pe > gpurun_out/r04_legs_repeat.json 2> gpurun_out/r04_legs_repeat.err; head -c 1800 gpurun_out/r04_legs_repeat.json; echo
timeout 600 python tools/fast_perf.py gpe 500000 2>&1 | tail -2 | cut -c1-330
timeout 600 python tools/fast_perf.py pe 1000000 2>&1 | tail -1 | cut -c1-330
