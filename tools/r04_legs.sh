#!/bin/bash
# the two legs with an index build of their own (repeat-structured 256 Mbp genome; 256 Mbp SNP graph), outside the headline run
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py --only-legs ${LEGS:-repeat_pe,graph256_pe} > gpurun_out/r04_legs.json 2> gpurun_out/r04_legs.err
tail -c 400 gpurun_out/r04_legs.err; head -c 6000 gpurun_out/r04_legs.json
