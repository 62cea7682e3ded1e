#!/bin/bash
# the round's last build: the whole GPU suite, then a bench.py pass at 40 Mbp (same code path as the driver's run; the extras take their own indexes)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 330 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests_final2.log 2>&1; tail -3 gpurun_out/r04_gputests_final2.log
H2G_BENCH_GENOME=40e6 timeout 200 python bench.py --steps 10 --warmup 3 > gpurun_out/r04_bench_40Mbp_final2.json 2> gpurun_out/r04_bench_40Mbp_final2.err; echo bench rc $?
python - <<'P'
import json
try:
    d = json.loads(open('gpurun_out/r04_bench_40Mbp_final2.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline'].get('traffic'), d['roofline'].get('traffic_note', '')[:80])
    print('graph leg', {k: d['graph_index_pe'][k] for k in ('ms_per_step', 'fast_kernel_ms', 'pairs_handed_on')} if 'graph_index_pe' in d else d.get('extras_skipped'))
    print('keys', sorted(d.keys()))
except Exception as e:
    print('no bench line', e)
P
