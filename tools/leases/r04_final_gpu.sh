#!/bin/bash
# the whole GPU suite on the final build, then bench.py end to end on a small genome (every extra leg; the big legs off)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests.log 2>&1; tail -5 gpurun_out/r04_gputests.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
H2G_BENCH_GENOME=40e6 H2G_BENCH_BIG_DEADLINE=0 timeout 1500 python bench.py --steps 5 --warmup 2 > gpurun_out/r04_bench_40Mbp.json 2> gpurun_out/r04_bench_40Mbp.err; tail -c 300 gpurun_out/r04_bench_40Mbp.err; head -c 2500 gpurun_out/r04_bench_40Mbp.json
