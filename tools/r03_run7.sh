cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
{
timeout 300 python tools/fast_perf.py pe 1000000 2>&1 | tail -2
timeout 300 python tools/fast_perf.py se 1000000 2>&1 | tail -2
python - <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import numpy as np, bench
from hisat2_amd import api, synth
base, contigs = bench.build_index(".bench_cache", 4_900_000)
n = 1_000_000
m1, m2 = synth.make_pairs(contigs, n, 101, bench.SEED + 7, sub_rate=0.005)
c1, o1 = synth.flatten_reads(m1); c2, o2 = synth.flatten_reads(m2)
names = [str(i) for i in range(n)]
ix = api.Index(base); st = api.Stream(ix, max_reads=n, max_bases=c1.size)
st.set_reads(c1, o1); st.set_read_names(names); st.set_mates(c2, o2, names)
for _ in range(3): st.align_pairs_run()
st.sync()
for K in (1, 5, 20):
    t0 = time.perf_counter()
    for _ in range(K): st.align_pairs_run()
    st.sync()
    dt = time.perf_counter() - t0
    print("K=%d steps back to back: %.2f ms per step (wall)" % (K, 1e3 * dt / K))
c = st.counters()
print("last step: fast %.2f ms machine %.2f ms; fast done %d bailed %d" % (c.ms_fast_kernel, c.ms_align_kernel, c.n_fast, c.n_fast_bail))
PY
timeout 900 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_align.py tests/test_gpu_sam.py tests/test_gpu_chr22.py -x -q 2>&1 | tail -3
} > $OUT/fast_perf7.log 2>&1
cat $OUT/fast_perf7.log
