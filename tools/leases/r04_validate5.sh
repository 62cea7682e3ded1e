#!/bin/bash
# the machine pass's share on graph indexes chosen from the batch size (mach_div 0 = auto) next to fixed shares; graph equality cases of the fast pass
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests
timeout 600 python tools/graph_scale.py 8e6 500000 400,0,400 > gpurun_out/r04_graph_scale3.jsonl 2> gpurun_out/r04_graph_scale3.err
timeout 600 python tools/graph_scale.py 32e6 750000 400,0,1600 >> gpurun_out/r04_graph_scale3.jsonl 2>> gpurun_out/r04_graph_scale3.err
timeout 600 python tools/graph_scale.py 32e6 1000000 1600,0 >> gpurun_out/r04_graph_scale3.jsonl 2>> gpurun_out/r04_graph_scale3.err
cat gpurun_out/r04_graph_scale3.jsonl | cut -c1-1100
timeout 400 python -m pytest tests/test_gpu_fast_pass.py -x -q > gpurun_out/r04_validate5_tests.log 2>&1; tail -3 gpurun_out/r04_validate5_tests.log
