#!/bin/bash
# the mmap / threaded index loader through the command line and the ABI; the 256 Mbp graph leg with the fast pass OFF (the general machine alone)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests
timeout 1500 python -m pytest tests/test_gpu_sam.py tests/test_gpu_zy_spliced.py tests/test_gpu_parity.py tests/test_gpu_ext_search.py -x -q > gpurun_out/r04_validate3_tests.log 2>&1; tail -3 gpurun_out/r04_validate3_tests.log
timeout 600 python tools/cli_perf.py 4000000 2>&1 | tail -2 | cut -c1-420
H2G_GO_FAST=0 python bench.py --only-legs graph256_pe > gpurun_out/r04_legs_graph256_machine_only.json 2> gpurun_out/r04_legs_graph256_machine_only.err
python -c "
import json; r=json.load(open('gpurun_out/r04_legs_graph256_machine_only.json'))['graph256_pe']; print('graph256 machine only:', r['ms_per_step'], r['reads_per_s'], r['pairs_completed_by_the_fast_pass'], r['parity']['sam_lines_differing'])"
