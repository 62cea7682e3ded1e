cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for lib in libh2g_v1.so libh2g_v2.so libh2g.so; do echo "== $lib"; H2G_LIB=$PWD/hisat2_amd/$lib H2G_DUMP=/tmp/d_$lib.npy timeout 300 python tools/fast_perf.py pe 1000000 2>&1 | tail -14; done
H2G_GO_FAST=0 H2G_DUMP=/tmp/d_off.npy timeout 300 python tools/fast_perf.py pe 1000000 2>&1 | tail -1
python - <<'PY'
import numpy as np
a=np.load('/tmp/d_off.npy'); b=np.load('/tmp/d_libh2g.so.npy')
n=len(a)//104
a=a.reshape(n,104); b=b.reshape(n,104)
bad=np.nonzero((a!=b).any(axis=1))[0]
print("differing pairs", len(bad), bad[:10])
for i in bad[:5]:
    print(i, a[i].view(np.uint32)[:10], bytes(a[i][40:48]), bytes(a[i][72:80])); print(i, b[i].view(np.uint32)[:10], bytes(b[i][40:48]), bytes(b[i][72:80]))
PY
} > gpurun_out/fast_perf2.log 2>&1
cat gpurun_out/fast_perf2.log
