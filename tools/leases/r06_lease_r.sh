#!/bin/bash
# Round 6, lease R: the whole GPU suite on the drain launch + parked alignMate pairs (defaults: large batches only; the fast-pass file's cases force them on small ones),
# smoke(), bench.py 20 / 5 at 256 Mbp on the new defaults
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_r; mkdir -p $OUT
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; tail -4 $OUT/gputests.log | cut -c1-300
echo "tests after $(( $(date +%s) - T0 )) s"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
H2G_BENCH_GENOME=256e6 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_256.json 2> $OUT/bench_256.err
python -c "
import json; d = json.loads(open('$OUT/bench_256.json').read().strip().splitlines()[-1]); r = d['roofline']; print('256 Mbp 20/5:', d['ms_per_step'], d['value'], r['kernel_ms'], r['frac'], r.get('drain_launch'), r['pairs_handed_on'])"
tail -3 $OUT/bench_256.err
echo "done after $(( $(date +%s) - T0 )) s"
