#!/usr/bin/env python3
"""Occ-rank vs memory type of the side array (H2G_SIDES_MTYPE env): does a non-L2-allocating type give 64 B fetches?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hisat2_amd import api
nq = 1 << 26
ix = api.Index(synth_sides=15_300_000, seed=1)
st = api.Stream(ix)
for v in (0, 1, 2, 3):
    st.rank_synth(nq, 42, variant=v, repeats=1)
    ms, ck = st.rank_synth(nq, 42, variant=v, repeats=3)
    print(f"mtype {os.environ.get('H2G_SIDES_MTYPE','default'):12s} variant {v}: {ms:8.3f} ms {nq*64/ms/1e6:8.1f} GB/s ck {ck}")
