"""ctypes binding of tests/emul/libh2gemu.so — host instantiation of the device item functions (tests only)."""
import ctypes as C
import os
import subprocess

import numpy as np

from hisat2_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))


class Emu:
    def __init__(self, base, variant=""):
        """variant "" = the shipped kernel's configuration; "am" = alignMate in the fast path (tests/emul/Makefile)"""
        subprocess.run(["make", "-s", "-j2", "-C", os.path.join(HERE, "emul")], check=True)
        self.L = C.CDLL(os.environ.get("H2GEMU_LIB") or os.path.join(HERE, "emul", "libh2gemu%s.so" % ("_" + variant if variant else "")))
        vp = C.c_void_p
        self.L.h2gemu_load.argtypes = [C.c_char_p, C.POINTER(vp)]
        self.L.h2gemu_set_reads.argtypes = [vp, vp, vp, vp, C.c_size_t]
        self.L.h2gemu_rank.argtypes = [vp, vp, vp, C.c_size_t, vp]
        self.L.h2gemu_fm_search.argtypes = [vp, vp, C.c_size_t, C.c_uint32, vp]
        self.L.h2gemu_sa_resolve.argtypes = [vp, vp, C.c_size_t, C.c_uint32, vp, vp]
        self.L.h2gemu_sw_align.argtypes = [vp, vp, C.c_size_t, vp]
        self.L.h2gemu_adjust_with_alt.argtypes = [vp, vp, C.c_size_t, C.c_uint32, vp, vp]
        self.L.h2gemu_sa_resolve_graph.argtypes = [vp, vp, vp, C.c_size_t, C.c_uint32, vp, vp]
        self.L.h2gemu_graph_lf.argtypes = [vp, vp, C.c_size_t, C.c_uint32, vp, vp]
        self.L.h2gemu_fm_search_graph.argtypes = [vp, vp, C.c_size_t, C.c_uint32, C.c_uint32, vp, vp]
        self.L.h2gemu_extend.argtypes = [vp, vp, vp, C.c_size_t, vp]
        self.L.h2gemu_seed_extend.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
        self.h = vp()
        rc = self.L.h2gemu_load(base.encode(), C.byref(self.h))
        assert rc == 0, rc
        self.n_reads = 0

    def set_reads(self, codes, offs, quals=None):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint32)
        if quals is not None:
            quals = np.ascontiguousarray(np.frombuffer(quals, dtype=np.uint8) if isinstance(quals, (bytes, bytearray)) else quals, dtype=np.uint8)
            assert quals.size == codes.size
        self._keep = (codes, offs, quals)
        self.n_reads = len(offs) - 1
        self.L.h2gemu_set_reads(self.h, codes.ctypes.data, offs.ctypes.data, quals.ctypes.data if quals is not None else None, self.n_reads)

    def rank(self, rows, cs):
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        cs = np.ascontiguousarray(cs, dtype=np.uint8)
        out = np.empty(len(rows), dtype=np.uint32)
        self.L.h2gemu_rank(self.h, rows.ctypes.data, cs.ctypes.data, len(rows), out.ctypes.data)
        return out

    def fm_search(self, queries, khits=5):
        n = len(queries)
        q = (api.FmQuery * n)(*queries)
        out = (api.FmHit * n)()
        self.L.h2gemu_fm_search(self.h, q, n, khits, out)
        return out

    def sw_align(self, queries, repeats=1):
        n = len(queries)
        q = (api.SwQuery * n)(*queries)
        out = (api.SwResult * n)()
        self.L.h2gemu_sw_align(self.h, q, n, out)
        return out, 0.0

    def adjust_with_alt(self, queries, cap=8):
        n = len(queries)
        q = (api.AdjustQuery * n)(*queries)
        hits = (api.GHit * (n * cap))()
        nh = (C.c_uint32 * n)()
        self.L.h2gemu_adjust_with_alt(self.h, q, n, cap, hits, nh)
        return hits, nh

    def local_graph_lf(self, q6, k=10):
        """q6: (n, 6) uint32 rows {single, tidx, toff, top, bot, c} -> mapGLF / mapGLF1 on the covering local graph index"""
        q6 = np.ascontiguousarray(q6, dtype=np.uint32)
        n = len(q6)
        res = (api.GlfResult * n)()
        ie = (api.IEdges * n)()
        self.L.h2gemu_local_graph_lf.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p]
        self.L.h2gemu_local_graph_lf(self.h, q6.ctypes.data, n, k, res, ie)
        return res, ie

    def sa_resolve_graph(self, queries, iedges, cap=24):
        n = len(queries)
        q = (api.GsaQuery * n)(*queries)
        ie = (api.IEdges * n)(*iedges)
        co = (api.Coord * (n * cap))()
        res = (api.SaResult * n)()
        self.L.h2gemu_sa_resolve_graph(self.h, q, ie, n, cap, co, res)
        return co, res

    def graph_lf(self, queries, k=10):
        n = len(queries)
        q = (api.GlfQuery * n)(*queries)
        res = (api.GlfResult * n)()
        ie = (api.IEdges * n)()
        self.L.h2gemu_graph_lf(self.h, q, n, k, res, ie)
        return res, ie

    def fm_search_graph(self, queries, khits=10, kseeds=20):
        n = len(queries)
        q = (api.FmQuery * n)(*queries)
        out = (api.FmHit * n)()
        ie = (api.IEdges * n)()
        self.L.h2gemu_fm_search_graph(self.h, q, n, khits, kseeds, out, ie)
        return out, ie

    def sa_resolve(self, queries, cap=16):
        n = len(queries)
        q = (api.SaQuery * n)(*queries)
        co = (api.Coord * (n * cap))()
        res = (api.SaResult * n)()
        self.L.h2gemu_sa_resolve(self.h, q, n, cap, co, res)
        return co, res

    def extend(self, hits, args):
        n = len(hits)
        h = (api.GHit * n)(*hits)
        a = (api.ExtArgs * n)(*args)
        res = (api.ExtResult * n)()
        self.L.h2gemu_extend(self.h, h, a, n, res)
        return h, res

    def seed_extend(self, pseudogeneStop=0, khits=5):
        out = np.zeros(self.n_reads * 2, dtype=api.SEED_RESULT_DTYPE)
        self.L.h2gemu_seed_extend(self.h, pseudogeneStop, khits, out.ctypes.data)
        return out
