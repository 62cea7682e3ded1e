"""Driver for the host instantiation of the go() state machine (tests only)."""
import ctypes as C

import numpy as np

import sam_util as SU
from h2gemu_py import Emu


def set_options(e, bowtie2_dp, options):
    """reference command-line options -> the emulator's h2g_align_params block"""
    from hisat2_amd import api
    p = api.AlignParams()
    e.L.h2gemu_default_params.argtypes = [C.c_void_p, C.c_void_p]
    e.L.h2gemu_set_params.argtypes = [C.c_void_p, C.c_void_p]
    e.L.h2gemu_default_params(e.h, C.byref(p))
    p.bowtie2_dp = bowtie2_dp
    rest = p.apply_options(list(options))
    assert not rest, rest
    e.L.h2gemu_set_params(e.h, C.byref(p))
    return p


def set_splice_sites(e, sites, known=True, window=0, rdid_base=0):
    """sites: [(tidx, left, right, '+'|'-'), ...] (a splice-site file) or a ready (api.SpliceSite * n) array with its length"""
    from hisat2_amd import api
    if isinstance(sites, tuple) and len(sites) == 2 and not isinstance(sites[0], tuple):
        a, n = sites
    else:
        a, n = api.splice_site_array(sites, known), len(sites)
    e.L.h2gemu_set_splice_sites.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]
    e.L.h2gemu_set_splice_sites(e.h, a, n, window)
    e.L.h2gemu_set_rdid_base.argtypes = [C.c_void_p, C.c_uint32]
    e.L.h2gemu_set_rdid_base(e.h, rdid_base)


def emu_align(base, reads_list, qnames, no_spliced=1, bowtie2_dp=0, quals=None, options=(), splice_sites=None, window=0, rdid_base=0):
    e = Emu(base)
    if splice_sites:
        set_splice_sites(e, splice_sites, window=window, rdid_base=rdid_base)
    if options:
        set_options(e, bowtie2_dp, options)
    e.L.h2gemu_set_bowtie2_dp.argtypes = [C.c_void_p, C.c_uint32]
    e.L.h2gemu_set_bowtie2_dp(e.h, bowtie2_dp)
    codes = np.concatenate(reads_list).astype(np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads_list])]).astype(np.uint32)
    e.set_reads(codes, offs, quals)   # quals: flat phred+33 bytes, same offsets (FASTQ input); None = FASTA ('I')
    nb = "".join(qnames).encode()
    noffs = np.concatenate([[0], np.cumsum([len(q) for q in qnames])]).astype(np.uint32)
    n = len(reads_list)
    outs = (SU.ReadOut * n)()
    recs = (SU.AlnRec * (n * SU.AL_MAX_RESULTS))()
    e.L.h2gemu_align.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]
    e.L.h2gemu_align(e.h, no_spliced, nb, noffs.ctypes.data, outs, recs)
    return outs, recs
