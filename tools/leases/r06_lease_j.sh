#!/bin/bash
# Round 6, lease J: a machine stream's first kernels run in the stream's first run now — per-run times from the first run on, and bench.py's 20 / 5 window on a fresh box
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_j; mkdir -p $OUT
timeout 900 python tools/batches_first.py 256e6 1000000 10 > $OUT/batches_first.jsonl 2> $OUT/batches_first.err; head -12 $OUT/batches_first.jsonl | cut -c1-200; tail -3 $OUT/batches_first.err
H2G_BENCH_GENOME=256e6 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_256.json 2> $OUT/bench_256.err
python -c "
import json; d = json.loads(open('$OUT/bench_256.json').read().strip().splitlines()[-1]); print('256 Mbp 20/5:', d['ms_per_step'], d['roofline']['kernel_ms'])"
timeout 600 python -m pytest tests/test_gpu_fast_pass.py tests/test_gpu_fast_stress.py tests/test_gpu_batches.py -m gpu -x -q 2>&1 | tail -3
