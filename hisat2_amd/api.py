"""ctypes mirror of include/h2g.h (libh2g.so) — the host-side view used by tests and bench.py.

There is no fallback: if the HIP library is missing or no GPU is usable, loading / the first call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_TAIL, DEFAULT_ALIGN_MATE = 16, 0     # H2G_DEFAULT_TAIL / H2G_DEFAULT_ALIGN_MATE of csrc/h2g_kernels.hip (what a stream does without H2G_FAST_TAIL / H2G_FAST_AM)
# one h2g stream = 1 + up to 8 HIP streams that must run side by side; ROCclr's default of 4 hardware queues would serialise them (h2g_kernels.hip).
# Read at HIP runtime initialisation: set before anything touches the device.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
LIB_PATH = os.environ.get("H2G_LIBPATH") or os.path.join(_HERE, "libh2g.so")   # H2G_LIBPATH: development builds (tools/)
MAX = 0xFFFFFFFF
MAX_EDITS = 32
SEED_CAP = 5

u32, u8, i32, u64, i64 = C.c_uint32, C.c_uint8, C.c_int32, C.c_uint64, C.c_int64


class LoadOpts(C.Structure):
    _fields_ = [("device", i32), ("load_local", i32)]


class IndexInfo(C.Structure):
    _fields_ = [("len", u32), ("gbwtLen", u32), ("numNodes", u32), ("lineRate", i32), ("offRate", i32),
                ("ftabChars", i32), ("eftabLen", u32), ("linear", u32), ("sideSz", u32), ("sideGbwtSz", u32),
                ("sideGbwtLen", u32), ("numSides", u32), ("offsLen", u32), ("ftabLen", u32), ("nPat", u32),
                ("nFrag", u32), ("nZ", u32), ("minK", u32), ("nLocal", u32), ("nRefRecs", u32), ("device_bytes", u64)]


class FmQuery(C.Structure):
    _fields_ = [("read", u32), ("offset", u32), ("fw", u8), ("mode", u8), ("pseudogeneStop", u8), ("anchorStop", u8)]


FM_HIT_FIELDS = ("top", "bot", "node_top", "node_bot", "bwoff", "len", "hit_type", "cur", "done", "numPartialSearch",
                 "numUniqueSearch", "pseudogeneStop", "anchorStop", "nrank", "nside")


class FmHit(C.Structure):
    _fields_ = [(n, u32) for n in FM_HIT_FIELDS]


H2G_IEDGE_CAP = 24
H2G_ERR_ARG = -4          # include/h2g.h: bad argument / capacity exceeded


class IEdges(C.Structure):           # h2g_iedges
    _fields_ = [("n", u32), ("e", (u32 * 2) * H2G_IEDGE_CAP)]

    def pairs(self):
        return [(self.e[i][0], self.e[i][1]) for i in range(min(self.n, H2G_IEDGE_CAP))]


class GlfQuery(C.Structure):         # h2g_glf_query
    _fields_ = [("top", u32), ("bot", u32), ("c", C.c_uint8), ("single", C.c_uint8), ("pad", C.c_uint8 * 2)]


class GlfResult(C.Structure):        # h2g_glf_result
    _fields_ = [("ok", u32), ("top", u32), ("bot", u32), ("node_top", u32), ("node_bot", u32)]


class AdjustQuery(C.Structure):      # h2g_adjust_query
    _fields_ = [(n, u32) for n in ("read", "fw", "rdoff", "len", "tidx", "toff", "joinedOff")]


class GsaQuery(C.Structure):         # h2g_gsa_query
    _fields_ = [(n, u32) for n in ("top", "bot", "node_top", "node_bot", "maxelt", "len", "rejectStraddle")]


class SaQuery(C.Structure):
    _fields_ = [("top", u32), ("bot", u32), ("maxelt", u32), ("len", u32), ("rejectStraddle", u32)]


class Coord(C.Structure):
    _fields_ = [("tidx", u32), ("toff", u32), ("joinedOff", u32)]


class SaResult(C.Structure):
    _fields_ = [("ok", u32), ("ncoords", u32), ("straddled", u32), ("nsteps", u32)]


class Edit(C.Structure):
    _fields_ = [("pos", u32), ("chr", u8), ("qchr", u8), ("type", u8), ("pad", u8), ("snp", u32)]


class GHit(C.Structure):
    _fields_ = [("read", u32), ("fw", u32), ("rdoff", u32), ("len", u32), ("trim5", u32), ("trim3", u32),
                ("tidx", u32), ("toff", u32), ("joinedOff", u32), ("score", i64), ("nedits", u32), ("overflow", u32),
                ("edits", Edit * MAX_EDITS)]


class ExtSearchQuery(C.Structure):   # h2g_ext_search_query
    _fields_ = [("read", u32), ("rdoff", u32), ("lidx", u32), ("maxHitLen", u32), ("fw", u8), ("uniqueStop", u8), ("pad", u8 * 2)]


class ExtSearchHit(C.Structure):     # h2g_ext_search_hit
    _fields_ = [("nelt", u32), ("hitlen", u32), ("top", u32), ("bot", u32), ("uniqueStop", u32), ("nrank", u32), ("nside", u32), ("staged", u32)]


class ExtSearchStats(C.Structure):   # h2g_ext_search_stats
    _fields_ = [("n_local", u64), ("n_staged", u64), ("n_buckets", u64), ("n_buckets_staged", u64), ("lds_bytes_staged", u64),
                ("ms_staged", C.c_float), ("ms_hbm", C.c_float)]


class SwQuery(C.Structure):          # h2g_sw_query
    _fields_ = [("read", u32), ("fw", u32), ("tidx", u32), ("refoff", u32), ("minsc", C.c_int32), ("rnd", u32)]


class SwResult(C.Structure):         # h2g_sw_result
    _fields_ = [("found_align", C.c_int32), ("found", C.c_int32), ("best", C.c_int32), ("score", C.c_int32),
                ("off", C.c_int64), ("nedits", u32), ("gaps", u32), ("overflow", u32), ("rnd", u32),
                ("refl", C.c_int64), ("refr", C.c_int64), ("edits", Edit * MAX_EDITS)]


class ExtArgs(C.Structure):
    _fields_ = [("mm", u32), ("max_leftext", u32), ("max_rightext", u32)]


class ExtResult(C.Structure):
    _fields_ = [("extended", u32), ("leftext", u32), ("rightext", u32)]


class SeedExt(C.Structure):
    _fields_ = [("tidx", u32), ("toff", u32), ("joinedOff", u32), ("rdoff", u32), ("len", u32), ("score", i32)]


class SeedResult(C.Structure):
    _fields_ = [("hit", FmHit), ("ncoords", u32), ("straddled", u32), ("nsteps", u32), ("pad", u32),
                ("ext", SeedExt * SEED_CAP)]


class SeedParams(C.Structure):
    _fields_ = [("pseudogeneStop", u32), ("anchorStop", u32), ("khits", u32), ("search_variant", u32)]


class Counters(C.Structure):
    _fields_ = [("n_rank", u64), ("n_side", u64), ("n_sa_steps", u64), ("n_ext", u64), ("n_ref_bytes", u64),
                ("n_queries", u64), ("n_aligned", u64), ("n_overflow", u64), ("ms_search", C.c_float),
                ("ms_resolve_extend", C.c_float), ("ms_rank", C.c_float), ("ms_align", C.c_float), ("ms_align_kernel", C.c_float),
                ("n_second_pass", u64), ("n_fast", u64), ("n_fast_bail", u64), ("ms_fast_kernel", C.c_float), ("pad_", C.c_float), ("n_fast_side", u64), ("n_fast_sa_steps", u64),
                ("ms_drain_kernel", C.c_float), ("pad2_", C.c_float), ("n_drain_side", u64), ("n_drain_sa_steps", u64), ("n_adopted", u64)]


ALN_CAP = 10


class AlnRes(C.Structure):
    _fields_ = [("fw", u32), ("tidx", u32), ("toff", u32), ("len", u32), ("trim5", u32), ("trim3", u32), ("nedits", u32),
                ("pad", u32), ("score", i64), ("edits", Edit * MAX_EDITS)]


class SpliceSite(C.Structure):
    _fields_ = [("tidx", u32), ("left", u32), ("right", u32), ("readid", u32), ("dir", u8), ("fromfile", u8), ("known", u8), ("editdist", u8)]


def splice_site_array(sites, known=True):
    """[(tidx, left, right, '+'|'-'), ...] -> (SpliceSite * n) as --known-splicesite-infile / --novel-splicesite-infile load them"""
    a = (SpliceSite * max(1, len(sites)))()
    for i, (t, l, r, d) in enumerate(sites):
        a[i].tidx, a[i].left, a[i].right, a[i].readid = t, l, r, 0
        a[i].dir, a[i].fromfile, a[i].known = (2 if d == "+" else 3), 1, (1 if known else 0)
    return a


def read_splice_site_file(path, refnames):
    """the reference's splice-site file (SpliceSiteDB::read splice_site.cpp:727): name, left, right, strand per line"""
    idx = {n: i for i, n in reversed(list(enumerate(refnames)))}
    out = []
    toks = open(path).read().split()
    for k in range(0, len(toks) - 3, 4):
        if toks[k] in idx:
            out.append((idx[toks[k]], int(toks[k + 1]), int(toks[k + 2]), toks[k + 3][0]))
    return out


class ReadResult(C.Structure):
    _fields_ = [("nres", u32), ("nselect", u32), ("overflow", u32), ("nrank", u32), ("nsteps", u32), ("depth", u32),
                ("best", C.c_int32), ("secbest", C.c_int32), ("best_h2", u32), ("secbest_h2", u32)]


class AlignParams(C.Structure):
    _fields_ = [("khits", u32), ("kseeds", u32), ("no_spliced_alignment", u32), ("secondary", u32), ("bowtie2_dp", u32),
                ("mm_max", C.c_int32), ("mm_min", C.c_int32), ("n_pen", C.c_int32), ("rdg_const", C.c_int32), ("rdg_linear", C.c_int32),
                ("rfg_const", C.c_int32), ("rfg_linear", C.c_int32), ("sc_max", C.c_int32), ("sc_min", C.c_int32), ("score_min_type", u32),
                ("score_min_const", C.c_double), ("score_min_coeff", C.c_double), ("no_temp_splicesite", u32),
                ("min_intronlen", u32), ("max_intronlen", u32), ("pen_cansplice", C.c_int32), ("pen_noncansplice", C.c_int32),
                ("pen_canintronlen_type", u32), ("pen_noncanintronlen_type", u32), ("first_read_id", u32),
                ("pen_canintronlen_const", C.c_double), ("pen_canintronlen_coeff", C.c_double),
                ("pen_noncanintronlen_const", C.c_double), ("pen_noncanintronlen_coeff", C.c_double),
                ("min_anchor_len", u32), ("min_anchor_len_noncan", u32), ("xs_only", u32), ("use_haplotype", u32), ("max_alts_tried", u32), ("max_frag_len", u32), ("min_frag_len", u32), ("pe_orientation", u32), ("nofw", u32), ("norc", u32)]

    def apply_options(self, opts, linear=None):
        """apply a list of reference command-line options (['-k', '3', '--mp', '4,2', ...]) to this block; returns leftovers.
        Presets are resolved AFTER all options were read, as hisat2.cpp:1882-1909 / 3174-3176 / 3903-3906 does (the same
        rules as h2g_align_params_presets): --sensitive alone keeps -k 5 on a linear index, and a preset's --score-min wins.
        `linear` defaults to what the block's default -k says about the index (5 = linear, 10 = graph)."""
        if linear is None:
            linear = self.khits == 5
        rest, i = [], 0
        saw_k, k_arg, max_seeds, sensitive, very = False, 0, 0, False, False
        ignore_quals = False
        dta = False
        while i < len(opts):
            o = opts[i]
            v = opts[i + 1] if i + 1 < len(opts) else None
            if o == "-k":
                k_arg = int(v); saw_k = True; i += 2
            elif o == "--max-seeds":
                max_seeds = int(v); i += 2
            elif o == "--secondary":
                self.secondary = 1; i += 1
            elif o == "--no-temp-splicesite":
                self.no_temp_splicesite = 1; i += 1
            elif o == "--no-spliced-alignment":
                self.no_spliced_alignment = 1; i += 1
            elif o == "--spliced":          # test shorthand: the reference's default mode
                self.no_spliced_alignment = 0; i += 1
            elif o == "--sensitive":
                sensitive = True; i += 1
            elif o == "--very-sensitive":
                very = True; i += 1
            elif o == "--bowtie2-dp":
                self.bowtie2_dp = int(v); i += 2
            elif o == "--mp":
                a = v.split(","); self.mm_max = int(a[0]); self.mm_min = int(a[1]) if len(a) > 1 else self.mm_min; i += 2
            elif o == "--sp":
                a = v.split(",")   # the reference reads BOTH max and min from the first number (aligner_seed_policy.cpp:438-441)
                self.sc_max = self.sc_min = int(a[0]); i += 2
            elif o == "--no-softclip":
                self.sc_max = self.sc_min = 2 ** 31 - 1; i += 1
            elif o == "--np":
                self.n_pen = int(v); i += 2
            elif o == "--rdg":
                a = v.split(","); self.rdg_const = int(a[0]); self.rdg_linear = int(a[1]) if len(a) > 1 else self.rdg_linear; i += 2
            elif o == "--rfg":
                a = v.split(","); self.rfg_const = int(a[0]); self.rfg_linear = int(a[1]) if len(a) > 1 else self.rfg_linear; i += 2
            elif o in ("--dta", "--downstream-transcriptome-assembly", "--dta-cufflinks"):
                dta = True
                if o == "--dta-cufflinks":
                    self.xs_only = 1
                i += 1
            elif o in ("--rna-strandness", "--novel-splicesite-outfile"):   # output only: h2g_sam_set_rna_strandness / h2g_sam_novel_splice_sites_text
                i += 2
            elif o in ("-I", "--minins"):
                self.min_frag_len = int(v); i += 2
            elif o in ("--fr", "--rf", "--ff"):
                self.pe_orientation = {"--fr": 0, "--rf": 1, "--ff": 2}[o]; i += 1
            elif o == "--nofw":
                self.nofw = 1; i += 1
            elif o == "--norc":
                self.norc = 1; i += 1
            elif o in ("-X", "--maxins"):
                self.max_frag_len = int(v); i += 2
            elif o == "--max-altstried":
                self.max_alts_tried = int(v); i += 2
            elif o == "--haplotype":
                self.use_haplotype = 1; i += 1
            elif o in ("--rg-id", "--rg"):                                   # output only: h2g_sam_add_read_group
                i += 2
            elif o == "--ignore-quals":
                ignore_quals = True; i += 1
            elif o == "--summary-file":
                i += 2
            elif o in ("--no-sq", "--omit-sec-seq", "--new-summary", "--add-chrname", "--remove-chrname"):                         # output only: h2g_sam_set_header_options
                i += 1
            elif o in ("--no-mixed", "--no-discordant"):                      # output only: h2g_sam_set_report_policy
                i += 1
            elif o == "--no-templatelen-adjustment":                       # output only (TLEN): h2g_sam_set_templatelen_adjustment
                i += 1
            elif o == "--min-intronlen":
                self.min_intronlen = int(v); i += 2
            elif o == "--max-intronlen":
                self.max_intronlen = int(v); i += 2
            elif o == "--pen-cansplice":
                self.pen_cansplice = int(v); i += 2
            elif o == "--pen-noncansplice":
                self.pen_noncansplice = int(v); i += 2
            elif o in ("--pen-canintronlen", "--pen-intronlen", "--pen-noncanintronlen"):
                a = v.split(",")            # PARSE_FUNC aligner_seed_policy.cpp:47: only the fields given are changed
                w = "pen_noncanintronlen" if o == "--pen-noncanintronlen" else "pen_canintronlen"
                setattr(self, w + "_type", {"C": 1, "L": 2, "S": 3, "G": 4}[a[0]])
                if len(a) > 1:
                    setattr(self, w + "_const", float(a[1]))
                if len(a) > 2:
                    setattr(self, w + "_coeff", float(a[2]))
                i += 2
            elif o == "--score-min":
                a = v.split(",")
                self.score_min_type = {"C": 1, "L": 2, "S": 3, "G": 4}[a[0]]
                self.score_min_const = float(a[1]) if len(a) > 1 else 0.0
                self.score_min_coeff = float(a[2]) if len(a) > 2 else 0.0
                i += 2
            else:
                rest.append(o); i += 1
        if dta:                              # hisat2.cpp:3920, 4078: applied after every option was read
            self.min_anchor_len, self.min_anchor_len_noncan = 15, 20
            self.pen_noncanintronlen_type, self.pen_noncanintronlen_const, self.pen_noncanintronlen_coeff = 4, -8.0, 2.0
        if ignore_quals and "--mp" not in opts:   # COST_MODEL_CONSTANT: every mismatch costs the maximum (aligner_seed_policy.cpp:279); --mp sets the quality model again (:418)
            self.mm_min = self.mm_max
        self.presets(linear, saw_k, k_arg, max_seeds, sensitive, very)
        return rest

    def presets(self, linear, saw_k, k_arg, max_seeds, sensitive, very_sensitive):
        """Python mirror of h2g_align_params_presets (tests/test_abi.py checks the two against each other)"""
        import numpy as np
        khits = k_arg if saw_k else 10
        if sensitive:
            if self.bowtie2_dp == 0:
                self.bowtie2_dp = 1
            if khits < 10:
                khits, saw_k = 10, True
            self.score_min_type, self.score_min_const, self.score_min_coeff = 2, 0.0, float(np.float32(-0.5))
        elif very_sensitive:
            self.bowtie2_dp = 2
            if khits < 30:
                khits, saw_k = 30, True
            self.score_min_type, self.score_min_const, self.score_min_coeff = 2, 0.0, float(np.float32(-1.0))
        if not saw_k:
            khits = 5 if linear else 10
        self.khits = khits
        self.kseeds = max_seeds if max_seeds else max(5, 2 * khits)


PAIR_RES_CAP = 16
PAIR_CAP = 32


class PairResult(C.Structure):
    _fields_ = [("nres", u32 * 2), ("npairs", u32), ("overflow", u32), ("nrank", u32), ("nsteps", u32), ("depth", u32),
                ("nside", u32), ("rnd_state", u32), ("pad", u32), ("pair_i", u8 * PAIR_CAP), ("pair_j", u8 * PAIR_CAP)]


READ_RESULT_DTYPE = np.dtype([(n, np.uint32) for n in ("nres", "nselect", "overflow", "nrank", "nsteps", "depth")] +
                             [("best", np.int32), ("secbest", np.int32), ("best_h2", np.uint32), ("secbest_h2", np.uint32)])


# numpy views of the result structs (same memory layout)
FM_HIT_DTYPE = np.dtype([(n, np.uint32) for n in FM_HIT_FIELDS])
SEED_EXT_DTYPE = np.dtype([("tidx", np.uint32), ("toff", np.uint32), ("joinedOff", np.uint32), ("rdoff", np.uint32),
                           ("len", np.uint32), ("score", np.int32)])
SEED_RESULT_DTYPE = np.dtype([("hit", FM_HIT_DTYPE), ("ncoords", np.uint32), ("straddled", np.uint32),
                              ("nsteps", np.uint32), ("pad", np.uint32), ("ext", SEED_EXT_DTYPE, (SEED_CAP,))])
assert SEED_RESULT_DTYPE.itemsize == C.sizeof(SeedResult)

EXPORTS = [
    "h2g_load_opts_init", "h2g_index_load", "h2g_index_get_info", "h2g_index_synth_sides", "h2g_index_free", "h2g_index_set_splice_sites", "h2g_index_add_splice_sites",
    "h2g_last_error", "h2g_stream_create", "h2g_stream_free", "h2g_stream_hip", "h2g_stream_sync", "h2g_stream_select_batch", "h2g_set_reads",
    "h2g_rank_bench", "h2g_rank_bench_synth", "h2g_rank_bench_synth_sample", "h2g_fm_search", "h2g_sa_resolve", "h2g_extend",
    "h2g_seed_params_init", "h2g_seed_extend_run", "h2g_seed_extend_fetch", "h2g_get_counters",
    "h2g_device_count", "h2g_ext_search", "h2g_local_index_of", "h2g_align_params_init", "h2g_align_params_presets", "h2g_set_read_names", "h2g_align_run", "h2g_align_fetch",
    "h2g_set_mates", "h2g_combine_with", "h2g_align_pairs_run", "h2g_align_pairs_fetch", "h2g_align_fetch_dense", "h2g_align_pairs_fetch_dense", "h2g_align_fetch_long_edits",
    "h2g_align_fetch_compact", "h2g_align_pairs_fetch_compact", "h2g_host_alloc", "h2g_host_free",
    "h2g_graph_lf", "h2g_fm_search_graph", "h2g_index_synth_graph_sides", "h2g_sw_align", "h2g_sa_resolve_graph", "h2g_adjust_with_alt",
]


class H2GError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libh2g.so (built by hisat2_amd/csrc/Makefile).  Raises if it is not there: no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    global LIB_PATH
    if os.environ.get("H2G_LIB"):          # development: a variant build of the library
        LIB_PATH = os.environ["H2G_LIB"]
    if not os.path.exists(LIB_PATH):
        raise H2GError(f"{LIB_PATH} not built (run __graft_entry__.build()); hisat2_amd has no CPU path")
    L = C.CDLL(LIB_PATH)
    P, vp = C.POINTER, C.c_void_p
    L.h2g_last_error.restype = C.c_char_p
    L.h2g_index_load.argtypes = [C.c_char_p, P(LoadOpts), P(vp)]
    L.h2g_index_get_info.argtypes = [vp, P(IndexInfo)]
    L.h2g_index_synth_sides.argtypes = [u64, u64, C.c_int, P(vp)]
    L.h2g_index_free.argtypes = [vp]
    L.h2g_index_free.restype = None
    L.h2g_stream_create.argtypes = [vp, C.c_size_t, C.c_size_t, P(vp)]
    L.h2g_stream_free.argtypes = [vp]
    L.h2g_stream_free.restype = None
    L.h2g_stream_hip.argtypes = [vp]
    L.h2g_stream_hip.restype = vp
    L.h2g_stream_sync.argtypes = [vp]
    L.h2g_stream_select_batch.argtypes = [vp, C.c_uint]
    L.h2g_set_reads.argtypes = [vp, vp, vp, vp, C.c_size_t]
    L.h2g_rank_bench.argtypes = [vp, vp, vp, C.c_size_t, vp, C.c_int, C.c_int, C.c_int, P(C.c_float)]
    L.h2g_rank_bench_synth.argtypes = [vp, C.c_size_t, u64, C.c_int, C.c_int, P(C.c_float), P(u64)]
    L.h2g_rank_bench_synth_sample.argtypes = [vp, C.c_size_t, C.c_size_t, vp]
    L.h2g_fm_search.argtypes = [vp, vp, C.c_size_t, u32, vp]
    L.h2g_sa_resolve.argtypes = [vp, vp, C.c_size_t, u32, vp, vp]
    L.h2g_sw_align.argtypes = [vp, vp, C.c_size_t, vp, C.c_int, P(C.c_float)]
    L.h2g_adjust_with_alt.argtypes = [vp, vp, C.c_size_t, u32, vp, vp]
    L.h2g_sa_resolve_graph.argtypes = [vp, vp, vp, C.c_size_t, u32, vp, vp]
    L.h2g_graph_lf.argtypes = [vp, vp, C.c_size_t, u32, vp, vp]
    L.h2g_fm_search_graph.argtypes = [vp, vp, C.c_size_t, u32, u32, vp, vp]
    L.h2g_index_synth_graph_sides.argtypes = [u64, u64, C.c_int, P(vp)]
    L.h2g_extend.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.h2g_seed_params_init.argtypes = [P(SeedParams), vp, C.c_int]
    L.h2g_seed_params_init.restype = None
    L.h2g_seed_extend_run.argtypes = [vp, P(SeedParams)]
    L.h2g_seed_extend_fetch.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
    L.h2g_get_counters.argtypes = [vp, P(Counters)]
    L.h2g_align_params_init.argtypes = [P(AlignParams), vp]
    L.h2g_align_params_init.restype = None
    L.h2g_align_params_presets.argtypes = [P(AlignParams), vp, C.c_int, u32, u32, C.c_int, C.c_int]
    L.h2g_align_params_presets.restype = None
    L.h2g_ext_search.argtypes = [vp, vp, C.c_size_t, u32, vp, vp]
    L.h2g_local_index_of.argtypes = [vp, u32, u32]
    L.h2g_local_index_of.restype = u32
    L.h2g_set_read_names.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    L.h2g_align_run.argtypes = [vp, P(AlignParams)]
    L.h2g_align_fetch.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t]
    L.h2g_set_mates.argtypes = [vp, vp, vp, vp, C.c_char_p, vp, C.c_size_t]
    L.h2g_align_pairs_run.argtypes = [vp, P(AlignParams)]
    L.h2g_align_pairs_fetch.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_size_t]
    L.h2g_align_fetch_dense.argtypes = [vp, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t]
    L.h2g_align_pairs_fetch_dense.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t]
    _lib = L
    return L


def _chk(rc, what):
    if rc != 0:
        msg = lib().h2g_last_error()
        raise H2GError(f"{what} failed: status {rc} ({msg.decode() if msg else ''})")


class Index:
    def __init__(self, base=None, device=0, synth_sides=None, seed=20260925, graph=False):
        L = lib()
        self.h = C.c_void_p()
        if synth_sides is not None and graph:
            _chk(L.h2g_index_synth_graph_sides(int(synth_sides), int(seed), int(device), C.byref(self.h)),
                 "h2g_index_synth_graph_sides")
        elif synth_sides is not None:
            _chk(L.h2g_index_synth_sides(int(synth_sides), int(seed), int(device), C.byref(self.h)), "h2g_index_synth_sides")
        else:
            o = LoadOpts(device, 1)
            _chk(L.h2g_index_load(base.encode(), C.byref(o), C.byref(self.h)), "h2g_index_load")
        self.info = IndexInfo()
        _chk(L.h2g_index_get_info(self.h, C.byref(self.info)), "h2g_index_get_info")

    def close(self):
        if self.h:
            lib().h2g_index_free(self.h)
            self.h = C.c_void_p()


class PinnedPool:
    """page-locked host buffers by key (h2g_host_alloc), grown on demand, handed out as numpy views: what a caller moving results over PCIe allocates once"""
    def __init__(self):
        L = lib()
        L.h2g_host_alloc.restype = C.c_void_p
        L.h2g_host_alloc.argtypes = [C.c_size_t]
        L.h2g_host_free.argtypes = [C.c_void_p]
        self._b = {}

    def array(self, key, nbytes, dtype=np.uint8):
        p, cap = self._b.get(key, (None, 0))
        if cap < nbytes:
            if p:
                lib().h2g_host_free(p)
            cap = nbytes + nbytes // 4 + 4096
            p = lib().h2g_host_alloc(cap)
            if not p:
                raise H2GError("h2g_host_alloc(%d) failed" % cap)
            self._b[key] = (p, cap)
        raw = (C.c_uint8 * cap).from_address(p)
        return np.frombuffer(raw, dtype=dtype, count=nbytes // np.dtype(dtype).itemsize)

    def close(self):
        for p, _ in self._b.values():
            lib().h2g_host_free(p)
        self._b = {}


class Stream:
    def __init__(self, index: Index, max_reads=0, max_bases=0):
        self.ix = index
        self.h = C.c_void_p()
        _chk(lib().h2g_stream_create(index.h, max_reads, max_bases, C.byref(self.h)), "h2g_stream_create")
        self.n_reads = 0
        self._batch = 0
        self._batch_n = {}

    def select_batch(self, k: int):
        """resident batch k (h2g_stream_select_batch): the set_* calls, runs and fetches that follow are its own; runs over other batches stay in flight"""
        self._batch_n[self._batch] = self.n_reads
        _chk(lib().h2g_stream_select_batch(self.h, k), "h2g_stream_select_batch")
        self._batch = k
        self.n_reads = self._batch_n.get(k, 0)

    def close(self):
        if self.h:
            lib().h2g_stream_free(self.h)
            self.h = C.c_void_p()

    def set_reads(self, codes: np.ndarray, offs: np.ndarray, quals=None):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint32)
        n = len(offs) - 1
        q = None
        if quals is not None:
            q = np.ascontiguousarray(quals, dtype=np.uint8).ctypes.data
        _chk(lib().h2g_set_reads(self.h, codes.ctypes.data, offs.ctypes.data, q, n), "h2g_set_reads")
        self.n_reads = n

    def rank(self, rows, cs, variant=0, repeats=1):
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        cs = np.ascontiguousarray(cs, dtype=np.uint8)
        out = np.empty(len(rows), dtype=np.uint32)
        ms = C.c_float(0)
        _chk(lib().h2g_rank_bench(self.h, rows.ctypes.data, cs.ctypes.data, len(rows), out.ctypes.data, variant, 0,
                                  repeats, C.byref(ms)), "h2g_rank_bench")
        return out, ms.value

    def rank_synth(self, n, seed, variant=0, repeats=1):
        ms, ck = C.c_float(0), u64(0)
        _chk(lib().h2g_rank_bench_synth(self.h, n, seed, variant, repeats, C.byref(ms), C.byref(ck)), "h2g_rank_bench_synth")
        return ms.value, ck.value

    def rank_synth_sample(self, stride, nsample):
        """results j * stride (j < nsample) of the last rank_synth run"""
        out = np.empty(nsample, dtype=np.uint32)
        _chk(lib().h2g_rank_bench_synth_sample(self.h, stride, nsample, out.ctypes.data), "h2g_rank_bench_synth_sample")
        return out

    def fm_search(self, queries, khits=5):
        n = len(queries)
        q = (FmQuery * n)(*queries)
        out = (FmHit * n)()
        _chk(lib().h2g_fm_search(self.h, q, n, khits, out), "h2g_fm_search")
        return out

    def ext_search(self, queries, stage_min=8):
        """h2g_ext_search: (hits, stats)"""
        n = len(queries)
        q = (ExtSearchQuery * n)(*queries) if not isinstance(queries, C.Array) else queries
        out = (ExtSearchHit * n)()
        st = ExtSearchStats()
        _chk(lib().h2g_ext_search(self.h, q, n, stage_min, out, C.byref(st)), "h2g_ext_search")
        return out, st

    def sw_align(self, queries, repeats=1):
        """SwAligner call site of hybridSearch (frame + end-to-end DP, 8-bit cells or 16-bit for minsc < -254, + gather + first backtrace) -> (results, kernel ms)"""
        n = len(queries)
        q = (SwQuery * n)(*queries)
        out = (SwResult * n)()
        ms = C.c_float(0)
        _chk(lib().h2g_sw_align(self.h, q, n, out, repeats, C.byref(ms)), "h2g_sw_align")
        return out, ms.value

    def adjust_with_alt(self, queries, cap=8):
        n = len(queries)
        q = (AdjustQuery * n)(*queries)
        hits = (GHit * (n * cap))()
        nh = (u32 * n)()
        _chk(lib().h2g_adjust_with_alt(self.h, q, n, cap, hits, nh), "h2g_adjust_with_alt")
        return hits, nh

    def sa_resolve_graph(self, queries, iedges, cap=24):
        n = len(queries)
        q = (GsaQuery * n)(*queries)
        ie = (IEdges * n)(*iedges)
        co = (Coord * (n * cap))()
        res = (SaResult * n)()
        _chk(lib().h2g_sa_resolve_graph(self.h, q, ie, n, cap, co, res), "h2g_sa_resolve_graph")
        return co, res

    def graph_lf(self, queries, k=10):
        """GFM::mapGLF / mapGLF1 on a graph index -> (results, in-edge lists)"""
        n = len(queries)
        q = (GlfQuery * n)(*queries)
        res = (GlfResult * n)()
        ie = (IEdges * n)()
        _chk(lib().h2g_graph_lf(self.h, q, n, k, res, ie), "h2g_graph_lf")
        return res, ie

    def fm_search_graph(self, queries, khits=10, kseeds=20):
        n = len(queries)
        q = (FmQuery * n)(*queries)
        out = (FmHit * n)()
        ie = (IEdges * n)()
        _chk(lib().h2g_fm_search_graph(self.h, q, n, khits, kseeds, out, ie), "h2g_fm_search_graph")
        return out, ie

    def sa_resolve(self, queries, cap=16):
        n = len(queries)
        q = (SaQuery * n)(*queries)
        co = (Coord * (n * cap))()
        res = (SaResult * n)()
        _chk(lib().h2g_sa_resolve(self.h, q, n, cap, co, res), "h2g_sa_resolve")
        return co, res

    def extend(self, hits, args):
        n = len(hits)
        h = (GHit * n)(*hits)
        a = (ExtArgs * n)(*args)
        res = (ExtResult * n)()
        _chk(lib().h2g_extend(self.h, h, a, n, res), "h2g_extend")
        return h, res

    def seed_params(self, no_spliced=True):
        p = SeedParams()
        lib().h2g_seed_params_init(C.byref(p), self.ix.h, 1 if no_spliced else 0)
        return p

    def seed_extend_run(self, params):
        _chk(lib().h2g_seed_extend_run(self.h, C.byref(params)), "h2g_seed_extend_run")

    def sync(self):
        _chk(lib().h2g_stream_sync(self.h), "h2g_stream_sync")

    def seed_extend_fetch(self, first=0, n=None):
        n = self.n_reads - first if n is None else n
        out = np.zeros(n * 2, dtype=SEED_RESULT_DTYPE)
        _chk(lib().h2g_seed_extend_fetch(self.h, out.ctypes.data, first, n), "h2g_seed_extend_fetch")
        return out

    def counters(self):
        c = Counters()
        _chk(lib().h2g_get_counters(self.h, C.byref(c)), "h2g_get_counters")
        return c

    @staticmethod
    def pack_names(qnames):
        """-> (bytes, uint32 offsets [n + 1]): the form the C ABI takes (a caller that uploads a batch more than once packs once)"""
        return "".join(qnames).encode(), np.concatenate([[0], np.cumsum([len(q) for q in qnames])]).astype(np.uint32)

    def set_read_names(self, qnames):
        nb, offs = qnames if isinstance(qnames, tuple) else self.pack_names(qnames)
        _chk(lib().h2g_set_read_names(self.h, nb, offs.ctypes.data, len(offs) - 1), "h2g_set_read_names")

    def align_params(self):
        p = AlignParams()
        lib().h2g_align_params_init(C.byref(p), self.ix.h)
        return p

    def align_run(self, params=None):
        params = params or self.align_params()
        _chk(lib().h2g_align_run(self.h, C.byref(params)), "h2g_align_run")

    def align_fetch(self, first=0, n=None, with_alignments=True):
        n = self.n_reads - first if n is None else n
        res = np.zeros(n, dtype=READ_RESULT_DTYPE)
        aln = (AlnRes * (n * ALN_CAP))() if with_alignments else None
        _chk(lib().h2g_align_fetch(self.h, res.ctypes.data, aln, first, n), "h2g_align_fetch")
        return res, aln

    def set_mates(self, codes2, offs2, qnames2, quals2=None):
        codes2 = np.ascontiguousarray(codes2, dtype=np.uint8)
        offs2 = np.ascontiguousarray(offs2, dtype=np.uint32)
        nb, noffs = qnames2 if isinstance(qnames2, tuple) else self.pack_names(qnames2)
        q = None if quals2 is None else np.ascontiguousarray(quals2, dtype=np.uint8).ctypes.data
        _chk(lib().h2g_set_mates(self.h, codes2.ctypes.data, offs2.ctypes.data, q, nb, noffs.ctypes.data, len(offs2) - 1), "h2g_set_mates")

    def align_fetch_dense(self, first=0, n=None):
        """-> (res, aln, offs): only the printed alignments, read i's records are aln[offs[i]:offs[i+1]]"""
        n = self.n_reads - first if n is None else n
        res = np.zeros(n, dtype=READ_RESULT_DTYPE)
        offs = np.zeros(n + 1, dtype=np.uint64)
        cap = n * 2 + 16
        while True:
            aln = (AlnRes * cap)()
            rc = lib().h2g_align_fetch_dense(self.h, res.ctypes.data, aln, cap, offs.ctypes.data, first, n)
            if rc == 0:
                return res, aln, offs
            if int(offs[n]) <= cap:
                _chk(rc, "h2g_align_fetch_dense")
            cap = int(offs[n])

    def align_pairs_fetch_dense(self, first=0, n=None):
        n = self.n_reads - first if n is None else n
        res = (PairResult * n)()
        o1 = np.zeros(n + 1, dtype=np.uint64)
        o2 = np.zeros(n + 1, dtype=np.uint64)
        c1 = c2 = n * 2 + 16
        while True:
            a1 = (AlnRes * c1)()
            a2 = (AlnRes * c2)()
            rc = lib().h2g_align_pairs_fetch_dense(self.h, res, a1, c1, o1.ctypes.data, a2, c2, o2.ctypes.data, first, n)
            if rc == 0:
                return res, a1, o1, a2, o2
            if int(o1[n]) <= c1 and int(o2[n]) <= c2:
                _chk(rc, "h2g_align_pairs_fetch_dense")
            c1, c2 = max(c1, int(o1[n])), max(c2, int(o2[n]))

    def tune(self, key, value):
        """development / measurement knobs of go_run by name (h2g_stream_tune: "mstreams", "mach_total", "fast_reserve", "pair_slots", "fast", ...)"""
        f = lib().h2g_stream_tune
        f.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
        _chk(f(self.h, key.encode(), int(value)), "h2g_stream_tune " + key)

    def align_pairs_fetch_compact(self, first=0, n=None, pinned=None):
        """-> (res, rec1, boffs1, rec2, boffs2): compact records (40 bytes + 12 per edit held, 8-aligned) as uint8 arrays, byte offsets [n + 1] per mate.
        pinned: a PinnedPool whose page-locked buffers the results land in (valid until its next use)"""
        n = self.n_reads - first if n is None else n
        f = lib().h2g_align_pairs_fetch_compact
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t]
        alloc = pinned.array if pinned is not None else (lambda key, nbytes, dtype=np.uint8: np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype))
        res = alloc("res", n * C.sizeof(PairResult))
        o1 = alloc("o1", (n + 1) * 8, np.uint64); o2 = alloc("o2", (n + 1) * 8, np.uint64)
        c1 = c2 = n * 64 + 4096
        for attempt in range(2):           # at most one retry, and only on "buffer too small" (H2G_ERR_ARG with the needed size in boffs[n]: buffers may be uninitialised memory)
            r1 = alloc("r1", c1); r2 = alloc("r2", c2)
            o1[n] = 0; o2[n] = 0
            rc = f(self.h, res.ctypes.data, r1.ctypes.data, r1.size, o1.ctypes.data, r2.ctypes.data, r2.size, o2.ctypes.data, first, n)
            if rc == 0:
                return res, r1, o1, r2, o2
            if attempt or rc != H2G_ERR_ARG or (int(o1[n]) <= r1.size and int(o2[n]) <= r2.size):
                _chk(rc, "h2g_align_pairs_fetch_compact")
            c1, c2 = max(c1, int(o1[n]) + 8), max(c2, int(o2[n]) + 8)

    def align_fetch_compact(self, first=0, n=None):
        n = self.n_reads - first if n is None else n
        f = lib().h2g_align_fetch_compact
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t]
        res = np.zeros(n, dtype=READ_RESULT_DTYPE)
        offs = np.zeros(n + 1, dtype=np.uint64)
        cap = n * 64 + 4096
        for attempt in range(2):           # (as align_pairs_fetch_compact: one retry, on "buffer too small" only)
            rec = np.empty(cap, dtype=np.uint8)
            offs[n] = 0
            rc = f(self.h, res.ctypes.data, rec.ctypes.data, cap, offs.ctypes.data, first, n)
            if rc == 0:
                return res, rec, offs
            if attempt or rc != H2G_ERR_ARG or int(offs[n]) <= cap:
                _chk(rc, "h2g_align_fetch_compact")
            cap = int(offs[n]) + 8

    def align_fetch_long_edits(self):
        """the used prefix of the stream's long-edit area (records with nedits > MAX_EDITS: edits[0].pos is their offset in it) -> (Edit array or None, n)"""
        n = C.c_size_t(0)
        f = lib().h2g_align_fetch_long_edits
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        f(self.h, None, 0, C.byref(n))
        if n.value == 0:
            return None, 0
        a = (Edit * n.value)()
        _chk(f(self.h, a, n.value, C.byref(n)), "h2g_align_fetch_long_edits")
        return a, n.value

    def align_pairs_run(self, params=None):
        params = params or self.align_params()
        _chk(lib().h2g_align_pairs_run(self.h, C.byref(params)), "h2g_align_pairs_run")

    def align_pairs_fetch(self, first=0, n=None, with_alignments=True):
        n = self.n_reads - first if n is None else n
        res = (PairResult * n)()
        a1 = (AlnRes * (n * PAIR_RES_CAP))() if with_alignments else None
        a2 = (AlnRes * (n * PAIR_RES_CAP))() if with_alignments else None
        _chk(lib().h2g_align_pairs_fetch(self.h, res, a1, a2, first, n), "h2g_align_pairs_fetch")
        return res, a1, a2

    def hip_stream(self):
        return lib().h2g_stream_hip(self.h)
