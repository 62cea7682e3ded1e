cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python tools/cli_perf.py 4000000 hisat2_amd/hisat2-align-amd-old hisat2_amd/hisat2-align-amd > $OUT/r03_run25.log 2>&1
cat $OUT/r03_run25.log


