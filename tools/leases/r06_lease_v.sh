#!/bin/bash
# Round 6, lease V: the repeat-structured leg is bounded by the latency of a machine pass times the eight in flight (lease U) — the passes' share of the chip once more, with the hand-ons
# at 21 000 instead of 27 000: mach_total 0 (the policy: 192 here) / 128 / 256 / 320 / 384; and the rest of the GPU suite with the drain launch forced on small batches
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_v; mkdir -p $OUT
T0=$(date +%s)
S="8,0,0,0;8,128,0,0;8,256,0,0;8,320,0,0;8,384,0,0;8,0,0,0"
timeout 900 python tools/queued_steps.py rep 256e6 1000000 "$S" > $OUT/rep.jsonl 2> $OUT/rep.err; cut -c1-330 $OUT/rep.jsonl; tail -3 $OUT/rep.err
echo "rep after $(( $(date +%s) - T0 )) s"
H2G_FAST_ORPHAN=64 H2G_DRAIN_GRID=8 timeout 1800 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fast_stress.py::test_queued_runs_equal_the_machine_read_by_read > $OUT/gputests_forced.log 2>&1; tail -4 $OUT/gputests_forced.log | cut -c1-400
echo "forced suite after $(( $(date +%s) - T0 )) s"
