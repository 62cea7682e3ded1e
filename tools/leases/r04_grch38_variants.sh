#!/bin/bash
# GRCh38-size (3.1 Gbp staged index): the fast kernel's variants measured where every index line is an HBM miss
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD:$PWD/tests
mkdir -p gpurun_out/r04_grch38
for v in ${VARIANTS:-shipped tail am tailam prof}; do
  if [ $v = shipped ]; then unset H2G_LIB; else export H2G_LIB=$PWD/hisat2_amd/csrc/obj/libh2g_$v.so; fi
  timeout 900 python tools/fast_perf.py pe 1000000 3.1e9 > gpurun_out/r04_grch38/pe_$v.log 2>&1; echo "== $v"; tail -${TAILN:-3} gpurun_out/r04_grch38/pe_$v.log | cut -c1-400
done
