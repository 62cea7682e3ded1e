#!/bin/bash
# first measurement of the next round (~2 GPU-minutes): chains of dependent rank queries per lane at the compact-state pass's occupancy (tools/chain_bench.py)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
# (a 128 Mbp SNP graph: 308 MB of index, beyond the 256 MB MALL; ~40 s to build on the box)
H2G_CHAIN_GRAPH_GENOME=128e6 timeout 400 python tools/chain_bench.py > gpurun_out/r05_chain_bench.json 2> gpurun_out/r05_chain_bench.err; echo rc $?
python - <<'P'
import json
d = json.load(open('gpurun_out/r05_chain_bench.json'))
for k in ('linear_64B', 'graph_128B'):
    print(k, 'checksums equal:', d[k]['checksums_equal'])
    for label, r in d[k].items():
        if isinstance(r, dict) and 'ms' in r:
            print('  %-20s %8.3f ms  %7.1f GB/s  %.3f of 8 TB/s' % (label, r['ms'], r['GB/s'], r['frac_of_8TBs']))
g = d['graph_lf_walks']
print('graph LF walks, checksums equal:', g['checksums_equal'])
for label, r in g.items():
    if isinstance(r, dict) and 'ms' in r:
        print('  %-44s %8.3f ms  %.2f G LF steps/s' % (label, r['ms'], r['lf_steps_per_s'] / 1e9))
P
