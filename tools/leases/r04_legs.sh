#!/bin/bash
# the two legs with an index build of their own (repeat-structured 256 Mbp genome; 256 Mbp SNP graph), outside the headline run; then the graph leg's workload
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests
mkdir -p gpurun_out
python bench.py --only-legs ${LEGS:-repeat_pe,graph256_pe} > gpurun_out/r04_legs.json 2> gpurun_out/r04_legs.err
tail -c 300 gpurun_out/r04_legs.err; head -c 400 gpurun_out/r04_legs.json; echo
timeout 600 python tools/fast_perf.py gpe 500000 2>&1 | tail -2 | cut -c1-330
