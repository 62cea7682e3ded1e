#!/bin/bash
# Round 5, lease K: the machine's share of the chip following the hand-on rate (mach_total 0 = the default policy) against the fixed shares, on the repeat-structured leg and the E. coli-size leg.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r05_k; mkdir -p $OUT
timeout 500 python tools/r05_mstreams.py rep 256e6 1000000 "8,0,0,0;8,128,0,0;8,192,0,0;8,0,0,0" > $OUT/mstreams_rep_auto.jsonl 2> $OUT/rep.err; cut -c1-300 $OUT/mstreams_rep_auto.jsonl
timeout 200 python tools/r05_mstreams.py rnd 4.9e6 1000000 "8,0,0,0;8,128,0,0;8,192,0,0" > $OUT/mstreams_rnd_auto.jsonl 2> $OUT/rnd.err; cut -c1-300 $OUT/mstreams_rnd_auto.jsonl
timeout 300 python -m pytest tests/test_gpu_fast_pass.py tests/test_gpu_fast_stress.py -q 2>&1 | tail -2
