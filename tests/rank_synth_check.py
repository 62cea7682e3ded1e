"""The Occ-rank micro-benchmark's sampled check ("verify 2^20 sampled outputs against the CPU GFM::mapLF").  The benchmark runs on a synthetic
side array the library generates on the device (k_synth_fill: payload word k of side s = splitmix64(seed + 16 s + k), Occ = the exclusive prefix of the per-side
symbol counts); this module rebuilds that array on the host, hands it to the ORACLE's own mapLF (oracle/h2o.c h2o_rank: SideLocus::initFromRow + countBt2Side +
fchr) and compares the sampled device outputs.  Test infrastructure: used by tests/ and by bench.py as the checker of its micro-benchmark line, never by the product."""
import ctypes as C

import numpy as np

import h2o_py as H

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
        return x ^ (x >> np.uint64(31))


def synth_linear_sides(nsides, seed, chunk=1 << 20):
    """(sides uint8[nsides * 64], fchr uint32[5]) exactly as h2g_index_synth_sides lays them out"""
    sides = np.zeros((nsides, 16), dtype=np.uint32)
    run = np.zeros(4, dtype=np.uint64)
    for a in range(0, nsides, chunk):
        b = min(nsides, a + chunk)
        s = np.arange(a, b, dtype=np.uint64)
        with np.errstate(over="ignore"):
            w = splitmix64((np.uint64(seed) + s[:, None] * np.uint64(16) + np.arange(6, dtype=np.uint64)[None, :]) & M64)       # (b - a, 6)
        sides[a:b, :12] = w.view(np.uint32).reshape(b - a, 12)
        cnt = np.empty((b - a, 4), dtype=np.uint64)
        for c in range(4):
            x = w ^ np.uint64((0x5555555555555555 * c) & 0xFFFFFFFFFFFFFFFF)          # symbol == c  <=>  both bits of the pair zero
            t = ~(x | (x >> np.uint64(1))) & np.uint64(0x5555555555555555)
            cnt[:, c] = np.bitwise_count(t).sum(axis=1)
        ex = np.cumsum(cnt, axis=0) - cnt + run[None, :]
        sides[a:b, 12:16] = ex.astype(np.uint32)
        run = run + cnt.sum(axis=0)
    fchr = np.zeros(5, dtype=np.uint32)
    fchr[1:] = np.cumsum(run).astype(np.uint32)
    return sides.reshape(-1).view(np.uint8), fchr


def queries(seed, idx, gbwt_len):
    """(row, c) of query i as the rank kernels draw it: h = splitmix64(seed + i), row = ((h >> 32) * gbwtLen) >> 32 (synth_row of h2g_kernels.hip), c = (h >> 40) & 3"""
    with np.errstate(over="ignore"):
        h = splitmix64((np.uint64(seed) + idx.astype(np.uint64)) & M64)
    return (((h >> np.uint64(32)) * np.uint64(gbwt_len)) >> np.uint64(32)).astype(np.uint32), ((h >> np.uint64(40)) & np.uint64(3)).astype(np.int32)


def oracle_map_lf(olib, sides, fchr, nsides, rows, cs):
    g = H.Gfm()
    p = g.p
    p.len = nsides * 192 - 1; p.gbwtLen = nsides * 192; p.numNodes = nsides * 192; p.lineRate = 6; p.offRate = 4; p.ftabChars = 10
    p.linear = 1; p.sideSz = 64; p.sideGbwtSz = 48; p.sideGbwtLen = 192; p.numSides = nsides; p.wsz = 4
    g.gfm = sides.ctypes.data_as(C.POINTER(C.c_uint8))
    g.nZ = 0
    for k in range(5):
        g.fchr[k] = int(fchr[k])
    olib.h2o_rank.restype = C.c_uint32
    olib.h2o_rank.argtypes = [C.POINTER(H.Gfm), C.c_uint32, C.c_int]
    f, gp = olib.h2o_rank, C.byref(g)
    return np.fromiter((f(gp, int(r), int(c)) for r, c in zip(rows, cs)), dtype=np.uint32, count=len(rows))


def sampled_expect(olib, nsides, seed, n, stride, nsample):
    """the oracle's mapLF for queries 0, stride, 2 stride, ... of an n-query run over h2g_index_synth_sides(nsides, seed)"""
    sides, fchr = synth_linear_sides(nsides, seed)
    idx = np.arange(nsample, dtype=np.uint64) * np.uint64(stride)
    assert int(idx[-1]) < n
    rows, cs = queries(seed, idx, nsides * 192)
    return oracle_map_lf(olib, sides, fchr, nsides, rows, cs)


def sampled_check(olib, device_samples, nsides, seed, n, stride):
    """device_samples[j] = output of query j * stride; returns (number compared, number differing)"""
    want = sampled_expect(olib, nsides, seed, n, stride, len(device_samples))
    return len(want), int((want != np.asarray(device_samples, dtype=np.uint32)).sum())
