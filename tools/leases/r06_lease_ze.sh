#!/bin/bash
# Round 6, lease ZE: the drain launch's tail hand-off (its last reads per workgroup go to the machine: 64 workgroups x 16 = a third of what the machine still gets on the random genome):
# 16 (the default) / 4 / 0, one process each (the knob is read when a stream is created), 256 Mbp random genome, then the repeat-structured one
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_ze; mkdir -p $OUT
T0=$(date +%s)
for leg in rnd rep; do
for t in 16 4 0 16 4 0; do
  H2G_FAST_TAIL=$t timeout 600 python tools/queued_steps.py $leg 256e6 1000000 "8,0,0,0,-1,64,-1" 2> $OUT/${leg}_$t.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$leg tail $t:', d['ms_per_step'], 'ms, handed on', d['handed_on'], 'drain', d['drain_ms_solo'], 'fast', d['fast_kernel_ms_solo'], 'crc', d['records_crc'])" | tee -a $OUT/tail.log
done
done
echo "done after $(( $(date +%s) - T0 )) s"
