"""ctypes binding of the CPU oracle (oracle/libh2o.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import gzip
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAX = 0xFFFFFFFF


class BwtHit(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "top", "bot", "node_top", "node_bot", "bwoff", "len", "hit_type", "cur", "done",
        "numPartialSearch", "numUniqueSearch", "pseudogeneStop", "anchorStop", "nrank", "nside")]


class Coord(C.Structure):
    _fields_ = [("tidx", C.c_uint32), ("toff", C.c_uint32), ("joinedOff", C.c_uint32)]


class Edit(C.Structure):
    _fields_ = [("pos", C.c_uint32), ("chr", C.c_uint8), ("qchr", C.c_uint8), ("type", C.c_uint8), ("pad", C.c_uint8),
                ("snp", C.c_uint32)]


class GHit(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("fw", "rdoff", "len", "trim5", "trim3", "tidx", "toff", "joinedOff")] + [
        ("score", C.c_int64), ("nedits", C.c_uint32), ("edits", Edit * 64)]


class Scoring(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("mmpMax", "mmpMin", "nPen", "rdGapConst", "rdGapLinear", "rfGapConst",
                                        "rfGapLinear", "scMax", "scMin", "matchBonus")]


class Params(C.Structure):
    _fields_ = [("len", C.c_uint32), ("gbwtLen", C.c_uint32), ("numNodes", C.c_uint32), ("lineRate", C.c_int32),
                ("offRate", C.c_int32), ("ftabChars", C.c_int32), ("eftabLen", C.c_uint32), ("linear", C.c_int),
                ("offMask", C.c_uint32), ("ftabLen", C.c_uint32), ("offsLen", C.c_uint32), ("sideSz", C.c_uint32),
                ("sideGbwtSz", C.c_uint32), ("sideGbwtLen", C.c_uint32), ("numSides", C.c_uint32),
                ("gbwtTotLen", C.c_uint32), ("wsz", C.c_int)]


class Gfm(C.Structure):
    _fields_ = [("p", Params), ("nPat", C.c_uint32), ("nFrag", C.c_uint32), ("plen", C.POINTER(C.c_uint32)),
                ("rstarts", C.POINTER(C.c_uint32)), ("gfm", C.POINTER(C.c_uint8)), ("nZ", C.c_uint32),
                ("zOffs", C.POINTER(C.c_uint32)), ("fchr", C.c_uint32 * 5), ("ftab", C.POINTER(C.c_uint32)),
                ("eftab", C.POINTER(C.c_uint32)), ("offs", C.POINTER(C.c_uint32)), ("tidx", C.c_uint32),
                ("localOffset", C.c_uint32), ("joinedOffset", C.c_uint32)]


class Ref(C.Structure):
    _fields_ = [("nrecs", C.c_uint32), ("nrefs", C.c_uint32), ("rec_off", C.POINTER(C.c_uint32)),
                ("rec_len", C.POINTER(C.c_uint32)), ("rec_first", C.POINTER(C.c_uint8)),
                ("refRecOffs", C.POINTER(C.c_uint32)), ("refOffs", C.POINTER(C.c_uint32)),
                ("refLens", C.POINTER(C.c_uint32)), ("buf", C.POINTER(C.c_uint8)), ("bufSz", C.c_uint64)]


class Index(C.Structure):
    _fields_ = [("g", Gfm), ("r", Ref), ("nlocal", C.c_uint32), ("local", C.POINTER(Gfm)),
                ("local_first", C.POINTER(C.c_uint32)), ("minK", C.c_uint32), ("names", C.POINTER(C.c_char_p)),
                ("nalts", C.c_uint32), ("alts", C.c_void_p)]


class SwResult(C.Structure):       # h2o_sw_result
    _fields_ = [("refl", C.c_int64), ("refr", C.c_int64), ("refl_pretrim", C.c_int64), ("refr_pretrim", C.c_int64),
                ("corel", C.c_int64), ("corer", C.c_int64), ("found_align", C.c_int32), ("best", C.c_int64),
                ("found", C.c_int32), ("score", C.c_int64), ("off", C.c_int64), ("nedits", C.c_uint32),
                ("gaps", C.c_uint32), ("overflow", C.c_uint32), ("edits", Edit * 64)]


def lcg_next(last):
    """RandomSource::nextU32 (random_source.h:52-61) -> (value, new state)"""
    last = (1664525 * last + 1013904223) & 0xFFFFFFFF
    r = last >> 16
    last = (1664525 * last + 1013904223) & 0xFFFFFFFF
    return r ^ last, last


def load():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libh2o.so"))
    P = C.POINTER
    lib.h2o_index_load.argtypes = [C.c_char_p, P(P(Index))]
    lib.h2o_index_load.restype = C.c_int
    lib.h2o_rank.argtypes = [P(Gfm), C.c_uint32, C.c_int]
    lib.h2o_rank.restype = C.c_uint32
    lib.h2o_rowL.argtypes = [P(Gfm), C.c_uint32]
    lib.h2o_rowL.restype = C.c_int
    lib.h2o_ftab_lohi.argtypes = [P(Gfm), C.c_void_p, C.c_uint32, P(C.c_uint32), P(C.c_uint32)]
    lib.h2o_ftab_lohi.restype = C.c_int
    lib.h2o_get_offset.argtypes = [P(Gfm), C.c_uint32, P(C.c_uint32)]
    lib.h2o_get_offset.restype = C.c_uint32
    lib.h2o_joined_to_text.argtypes = [P(Gfm), C.c_uint32, C.c_uint32, P(C.c_uint32), P(C.c_uint32), P(C.c_uint32),
                                       C.c_int, P(C.c_int)]
    lib.h2o_joined_to_text.restype = C.c_int
    lib.h2o_get_stretch.argtypes = [P(Ref), C.c_uint32, C.c_int64, C.c_uint32, C.c_void_p]
    lib.h2o_get_stretch.restype = None
    lib.h2o_partial_search.argtypes = [P(Index), C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint32,
                                       P(BwtHit)]
    lib.h2o_partial_search.restype = None
    u32 = C.c_uint32
    lib.h2o_partial_search_graph.argtypes = [P(Index), C.c_void_p, u32, u32, C.c_int, C.c_int, u32, u32, P(BwtHit),
                                             P(u32), u32, P(u32)]
    lib.h2o_partial_search_graph.restype = None
    lib.h2o_rank_M.argtypes = [P(Gfm), u32]
    lib.h2o_rank_M.restype = u32
    lib.h2o_select_F.argtypes = [P(Gfm), u32, u32]
    lib.h2o_select_F.restype = u32
    lib.h2o_map_glf.argtypes = [P(Gfm), u32, u32, C.c_int, u32] + [P(u32)] * 4 + [P(u32), u32, P(u32)]
    lib.h2o_map_glf.restype = C.c_int
    lib.h2o_map_glf1.argtypes = [P(Gfm), u32, C.c_int] + [P(u32)] * 4
    lib.h2o_map_glf1.restype = C.c_int
    lib.h2o_genome_coords_graph.argtypes = [P(Index), u32, u32, u32, u32, P(u32), u32, u32, u32, C.c_int, P(Coord), P(u32),
                                            P(C.c_int), P(u32)]
    lib.h2o_genome_coords_graph.restype = C.c_int
    lib.h2o_adjust_with_alt.argtypes = [P(Index), C.c_void_p, C.c_int, u32, u32, u32, u32, u32, P(GHit), P(u32), u32]
    lib.h2o_adjust_with_alt.restype = C.c_int
    lib.h2o_genome_coords.argtypes = [P(Index), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, P(Coord),
                                      P(C.c_uint32), P(C.c_int), P(C.c_uint32)]
    lib.h2o_genome_coords.restype = C.c_int
    lib.h2o_extend.argtypes = [P(Index), P(Scoring), C.c_void_p, C.c_char_p, C.c_uint32, P(GHit), P(C.c_uint32),
                               P(C.c_uint32), C.c_uint32]
    lib.h2o_extend.restype = C.c_int
    lib.h2o_scoring_default.argtypes = [P(Scoring)]
    lib.h2o_sw_align.argtypes = [P(Index), P(Scoring), C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.c_int64, C.c_int, C.c_int, P(C.c_uint32), P(SwResult)]
    lib.h2o_sw_align.restype = C.c_int
    lib.h2o_seed_extend_batch.argtypes = [P(Index), C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32,
                                          P(C.c_uint64)]
    lib.h2o_seed_extend_batch.restype = C.c_uint64
    return lib


def load_index(lib, base):
    ix = C.POINTER(Index)()
    rc = lib.h2o_index_load(base.encode(), C.byref(ix))
    assert rc == 0, rc
    return ix


CODE = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("N"): 4}
_LUT = np.full(256, 4, dtype=np.uint8)
for k, v in CODE.items():
    _LUT[k] = v
COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


def encode(s: bytes):
    return _LUT[np.frombuffer(s, dtype=np.uint8)]


def read_fasta_reads(path):
    op = gzip.open if path.endswith(".gz") else open
    names, seqs = [], []
    with op(path, "rb") as f:
        for line in f:
            line = line.strip()
            if line.startswith(b">"):
                names.append(line[1:].decode())
            elif line:
                seqs.append(encode(line))
    return names, seqs


def revcomp(codes):
    return COMP[codes[::-1]].copy()


def glines(golden_dir, name):
    with gzip.open(os.path.join(golden_dir, name), "rt") as f:
        return [l.rstrip("\n") for l in f if l.strip()]
