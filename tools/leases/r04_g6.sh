#!/bin/bash
# the graph fast kernel in other builds: does the device-only divergence follow the compiler?
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD:$PWD/tests
for v in ${VARIANTS:-gwa}; do
  echo "== $v"
  H2G_LIB=$PWD/hisat2_amd/csrc/obj/libh2g_$v.so bash tools/r04_g4.sh 2>&1 | grep -E "^(pairs|reads)" 
done
