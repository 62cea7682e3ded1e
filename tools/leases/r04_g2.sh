sed -i 's/for runs in 1 3 8/for runs in 3 8/' tools/r04_stress.sh
bash tools/r04_stress.sh > gpurun_out/r04_stress2.log 2>&1
for lib in shipped tail; do
  if [ $lib = shipped ]; then unset H2G_LIB; else export H2G_LIB=$PWD/hisat2_amd/csrc/obj/libh2g_$lib.so; fi
  python tools/fast_perf.py pe 1000000 > gpurun_out/r04_perf_pe_$lib.log 2>&1
  python tools/fast_perf.py se 1000000 > gpurun_out/r04_perf_se_$lib.log 2>&1
done
tail -2 gpurun_out/r04_perf_*.log
