#!/bin/bash
# The open inequality of round 3 (profiles/r04_NOTES.md): the tail hand-off variant of the fast kernel (make -C hisat2_amd/csrc variant:
# obj/libh2g_tail.so, -DFG_TAIL=16) and the shipped library through tests/fast_stress.py on the hard-read case, read by read.
set -u
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD:$PWD/tests
mkdir -p gpurun_out/r04_stress
python - <<'PY'
import os, sys, tempfile
sys.path.insert(0, "tests")
import test_gpu_fast_stress as T
tmp = tempfile.mkdtemp(prefix="h2fs")
base, npz = T.hard_case(tmp)
open("gpurun_out/r04_stress/case.txt", "w").write(base + " " + npz + "\n")
PY
read BASE NPZ < gpurun_out/r04_stress/case.txt
for lib in shipped tail; do
  for runs in 1 3 8; do
    if [ $lib = shipped ]; then unset H2G_LIB; else export H2G_LIB=$PWD/hisat2_amd/csrc/obj/libh2g_$lib.so; fi
    timeout 600 python tests/fast_stress.py $BASE $NPZ $runs all,0.63,0.2,999 > gpurun_out/r04_stress/${lib}_runs$runs.json 2> gpurun_out/r04_stress/${lib}_runs$runs.err
    echo "$lib runs=$runs rc=$?"; python -c "
import json,sys
r=json.load(open('gpurun_out/r04_stress/${lib}_runs$runs.json'))
for c in r['cases']: print('  ', c['kind'], c['n'], 'fast', c['fast'], 'handed_on', c['handed_on'], 'differing', c['differing'], c['first'][:2])
"
  done
done
