cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for lib in libh2g_v1.so libh2g.so; do echo "== $lib"; H2G_LIB=$PWD/hisat2_amd/$lib timeout 300 python tools/fast_perf.py pe 1000000 2>&1 | tail -14; done
H2G_LIB=$PWD/hisat2_amd/libh2g.so timeout 300 python tools/fast_perf.py se 1000000 2>&1 | tail -2
} > gpurun_out/fast_perf3.log 2>&1
cat gpurun_out/fast_perf3.log
