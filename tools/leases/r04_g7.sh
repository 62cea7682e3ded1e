#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD:$PWD/tests
for v in ${VARIANTS:-shipped gprof}; do
  if [ $v = shipped ]; then unset H2G_LIB; else export H2G_LIB=$PWD/hisat2_amd/csrc/obj/libh2g_$v.so; fi
  timeout 600 python tools/fast_perf.py gpe 500000 > gpurun_out/r04_g7_gpe_$v.log 2>&1; echo "== $v"; tail -22 gpurun_out/r04_g7_gpe_$v.log | cut -c1-300
done
unset H2G_LIB
bash tools/r04_g4.sh 2>&1 | grep -E "^(pairs|reads)"
