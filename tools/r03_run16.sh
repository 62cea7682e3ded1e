cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
L=$OUT/r03_run16.log; : > $L
H2G_DUMP=/tmp/d_off.npy H2G_GO_FAST=0 timeout 300 python tools/fast_perf.py se 1000000 >> $L 2>&1
H2G_DUMP=/tmp/d_main.npy timeout 300 python tools/fast_perf.py se 1000000 >> $L 2>&1
H2G_DUMP=/tmp/d_k65.npy H2G_LIB=$PWD/hisat2_amd/libh2g_k65.so timeout 300 python tools/fast_perf.py se 1000000 >> $L 2>&1
echo "off vs main" >> $L; python tools/r03_diff.py /tmp/d_off.npy /tmp/d_main.npy 40 >> $L 2>&1
echo "off vs k65" >> $L; python tools/r03_diff.py /tmp/d_off.npy /tmp/d_k65.npy 40 >> $L 2>&1
grep -v "^index ready\|bails:" $L
