"""The splice-site database kept up to date wave by wave (merge_splice_db, h2g_splice_db_host.h: what h2g_index_add_splice_sites and
h2g_sam_add_splice_sites do for the temporary-splice-site waves) against the database built from scratch over the same sites
(build_splice_db, pinned to SpliceSiteDB by the spliced SAM fixtures): identical arrays after every wave."""
import ctypes as C
import os

import numpy as np
import pytest

from hisat2_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("seed,n,npat,nwaves", [(1, 4000, 5, 17), (2, 300, 1, 300), (3, 20000, 24, 8), (4, 50, 3, 1)])
def test_incremental_database_equals_the_rebuilt_one(seed, n, npat, nwaves):
    L = C.CDLL(os.path.join(HERE, "emul", "libh2gemu.so"))
    L.h2gemu_splice_db_merge_check.restype = C.c_uint64
    L.h2gemu_splice_db_merge_check.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]
    rng = np.random.default_rng(seed)
    sites = (api.SpliceSite * n)()
    for i in range(n):
        s = sites[i]
        s.tidx = int(rng.integers(0, npat + 1))                # one text id past the end: such sites are dropped
        s.left = int(rng.integers(0, 400))                     # a small coordinate range: many equal sites, many updates
        s.right = s.left + int(rng.integers(2, 30))
        s.dir = int(rng.integers(1, 4))
        s.readid = int(rng.integers(0, 100000))
        s.fromfile = 1 if i < n // 10 and rng.integers(0, 2) else 0          # file sites come first and keep their ids
        s.known = s.fromfile
        s.editdist = 0
    assert L.h2gemu_splice_db_merge_check(sites, n, npat, nwaves) == 0
