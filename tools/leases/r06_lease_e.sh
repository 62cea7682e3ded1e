#!/bin/bash
# Round 6, lease E: one batch K times against 10 distinct batches on one stream (A/B in one process, first thing on a fresh box); the tests that changed; alignMate in the pass
# under the sticky loop (random + repeat-structured 256 Mbp); the rank micro-benchmark without its output stream.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$PWD:$PWD/tests:$PWD/tools
OUT=gpurun_out/r06_e; mkdir -p $OUT
T0=$(date +%s)
timeout 900 python tools/batches_ab.py 256e6 1000000 10 3 > $OUT/batches_ab.jsonl 2> $OUT/batches_ab.err; cat $OUT/batches_ab.jsonl | cut -c1-300; tail -3 $OUT/batches_ab.err
echo "batches after $(( $(date +%s) - T0 )) s"
timeout 900 python -m pytest tests/test_gpu_batches.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/gputests_new.log 2>&1; tail -4 $OUT/gputests_new.log | cut -c1-300
echo "tests after $(( $(date +%s) - T0 )) s"
for leg in rnd rep; do for am in 0 1; do
  H2G_FAST_AM=$am timeout 600 python tools/queued_steps.py $leg 256e6 1000000 "8,128,0,0" > $OUT/${leg}_am$am.jsonl 2> $OUT/${leg}_am$am.err; echo "$leg am=$am: $(tail -1 $OUT/${leg}_am$am.jsonl | cut -c1-420)"
done; done
echo "am after $(( $(date +%s) - T0 )) s"
python - <<'PY' > gpurun_out/r06_e/rank_variants.json 2> gpurun_out/r06_e/rank_variants.err
import json, sys
from hisat2_amd import api
n = 1 << 28
out = {}
ix = api.Index(synth_sides=15_300_000, seed=20260925 + 38, device=0)
st = api.Stream(ix)
for rep in range(2):
    for v in (0, 10, 6, 0, 10):
        st.rank_synth(n, 7, variant=v, repeats=1)
        ms, ck = st.rank_synth(n, 7, variant=v, repeats=3)
        out.setdefault(str(v), []).append({"ms": round(ms, 3), "GB/s": round(n * 64 / (ms * 1e-3) / 1e9, 1), "ck": int(ck)})
print(json.dumps(out))
PY
cat $OUT/rank_variants.json | cut -c1-900; tail -2 $OUT/rank_variants.err
echo "done after $(( $(date +%s) - T0 )) s"
