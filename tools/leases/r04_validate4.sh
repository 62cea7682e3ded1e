#!/bin/bash
# the 16-bit Smith-Waterman cells on the device (golden vectors of the reference's i16 path, the C oracle on random problems, go() against the
# live reference), then where the compact-state pass pays on a graph index (tools/graph_scale.py) and the quota-aware builder thread count
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests
timeout 400 python -m pytest tests/test_gpu_sw.py tests/test_gpu_align.py -x -q -k "sw or bowtie2" > gpurun_out/r04_validate4_sw.log 2>&1; tail -3 gpurun_out/r04_validate4_sw.log
python -c "
import sys; sys.path.insert(0,'tools'); import build_bench_index as BB, os; print('usable cpus', BB.usable_cpus(), 'of', os.cpu_count())"
timeout 900 python tools/graph_scale.py 32e6,128e6,256e6 1000000 200,400,1600 > gpurun_out/r04_graph_scale.jsonl 2> gpurun_out/r04_graph_scale.err; cat gpurun_out/r04_graph_scale.jsonl | cut -c1-900
