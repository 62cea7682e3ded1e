#!/usr/bin/env python3
"""debug: one fuzz case, one read: GPU with / without the second pass vs the host instantiation"""
import os, sys, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sam_util as SU
from hisat2_amd import synth
import test_gpu_align as T
from h2gemu_align import emu_align
seed, nreads, rdlen = 203, 10000, 101
focus = [int(x) for x in sys.argv[1:]] or [7798]
contigs = synth.make_genome([120000], seed, n_gaps=0, gap_len=300, repeats=300, repeat_len=500)
tmp = tempfile.mkdtemp(prefix="h2dbg")
fa = os.path.join(tmp, "g.fa"); synth.write_fasta(fa, contigs); base = os.path.join(tmp, "g")
subprocess.run([os.path.join(ROOT, "oracle", "_ref", "hisat2-build-s"), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
reads, _ = synth.make_reads(contigs, nreads, rdlen, seed + 1, sub_rate=0.003, indel_rate=0.0, n_rate=0.0)
rl = [reads[i] for i in range(nreads)]
qn = [str(i) for i in range(nreads)]
refnames = ["chr1"]
def show(tag, res, got):
    for f in focus:
        print(tag, "read", f, "overflow", int(res[f]["overflow"]) if hasattr(res[f], "dtype") else res[f].overflow, got[str(f)], flush=True)
for nosec in ("1", "0"):
    os.environ["H2G_GO_NO_SECOND_PASS"] = nosec
    res, aln, c = T.gpu_align(base, rl, qn)
    got = SU.render_selected(res, aln, refnames, rl, qn)
    show("gpu no_second=%s" % nosec, res, got)
    print("   flagged:", [i for i in range(nreads) if res[i]["overflow"]][:20], "second-pass", c.n_second_pass, "n_overflow", c.n_overflow)
outs, recs = emu_align(base, rl, qn)
got = SU.render(outs, recs, refnames, rl, qn)
show("emu big", outs, got)
# op trace of the focus read on the GPU (first pass only) vs the host instantiation
import ctypes as C
from hisat2_amd import api
os.environ["H2G_GO_NO_SECOND_PASS"] = "1"
os.environ["H2G_GO_DBG_READ"] = str(focus[0])
codes = np.concatenate(rl).astype(np.uint8)
offs = np.concatenate([[0], np.cumsum([len(r) for r in rl])]).astype(np.uint32)
ix = api.Index(base, device=0)
st = api.Stream(ix, max_reads=nreads, max_bases=codes.size)
st.set_reads(codes, offs, None); st.set_read_names(qn)
st.align_run(st.align_params()); st.sync()
buf = (C.c_uint32 * 65536)()
L = api.lib()
L.h2g_go_debug_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
print("trace rc", L.h2g_go_debug_trace(st.h, buf, 65536), "words", buf[0])
with open(os.path.join(ROOT, "gpurun_out", "r02_trace_gpu.txt"), "w") as f:
    for k in range(0, min(buf[0], 65000), 8):
        f.write("T " + " ".join(str(buf[1 + k + j]) for j in range(8)) + "\n")
