#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD:$PWD/tests
export H2G_LIB=$PWD/hisat2_amd/csrc/obj/libh2g_gdbg.so
python - <<'PY'
import os, sys, tempfile, subprocess, json, ctypes as C
sys.path.insert(0, "tests")
import numpy as np
from hisat2_amd import synth, api
import fast_stress as FS
tmp = tempfile.mkdtemp(prefix="h2fg")
contigs = synth.make_genome([1500000, 400000, 100000], 73, n_gaps=3, gap_len=300, repeats=80, repeat_len=600)
var = synth.make_snps(contigs, 82, every=250)
fa = os.path.join(tmp, "g.fa"); synth.write_fasta(fa, contigs); synth.write_snps(os.path.join(tmp, "g.snp"), var)
base = os.path.join(tmp, "g")
subprocess.run(["oracle/_ref/hisat2-build-s", "-q", "-p", "16", "--snp", os.path.join(tmp, "g.snp"), fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
alt = synth.apply_snps(contigs, var)
reads, _ = synth.make_reads(alt, 100000, 101, 75, sub_rate=0.005)
ix = api.Index(base, device=0)
for n, ids in ((3000, (2099,)),):
    rc, ro = synth.flatten_reads(np.asarray(reads)[:n])
    st = api.Stream(ix, max_reads=n, max_bases=rc.size + 64)
    st.set_reads(rc, ro); st.set_read_names([str(i) for i in range(n)])
    p = st.align_params(); p.no_spliced_alignment = 1
    for rid in ids:
        FS.tune(st, "dbg_read", rid)
        st.align_run(p)
        buf = (C.c_uint32 * 4096)()
        f = api.lib().h2g_go_debug_trace; f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        assert f(st.h, buf, 4096) == 0
        nw = buf[0]
        res, aln, offs = st.align_fetch_dense(0, n)
        print("batch", n, "read", rid, "words", nw, "result", res[rid])
        for k in range(0, min(nw, 4000), 12):
            d = buf[1 + k: 1 + k + 12]
            print("  op %u from pc %u/%u a %u %u %u %u %u %u -> pc %u op %u sp %u nrank %u nsteps %u co0.joff %u a4 %u" % (d[0], d[1] & 255, d[1] >> 8, d[2], d[3], d[4], d[5], d[6], d[7], d[8] & 255, (d[8] >> 8) & 255, d[8] >> 16, d[9] & 0xffff, d[9] >> 16, d[10], d[11]))
    st.close()
PY
