#!/usr/bin/env python3
"""Occ-rank memory-system experiments (MI355X): table size x kernel organisation.  Prints one line per cell.
variants: 0 lane/side  1 4-lanes/side  2 8-lanes/side  3 lane/side + partner half of the 128 B line
          4 lane/side nontemporal  5 lane/side, 2 queries in flight"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hisat2_amd import api  # noqa: E402

nq = 1 << 26
for nsides in (1_000_000, 4_000_000, 15_300_000, 22_000_000):
    ix = api.Index(synth_sides=nsides, seed=1)
    st = api.Stream(ix)
    ck0 = None
    for v in (0, 1, 2, 3, 4, 5):
        st.rank_synth(nq, 42, variant=v, repeats=1)
        ms, ck = st.rank_synth(nq, 42, variant=v, repeats=3)
        ck0 = ck if ck0 is None else ck0
        print(f"sides {nsides:>9d} ({nsides * 64 / 1e6:8.1f} MB) variant {v}: {ms:8.3f} ms  {nq * 64 / ms / 1e6:8.1f} GB/s algorithmic  {'ok' if ck == ck0 else 'CHECKSUM MISMATCH'}")
    st.close()
    ix.close()
