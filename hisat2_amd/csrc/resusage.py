#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks from build.log."""
import re, sys, os
txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "build.log")).read()
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    def g(k):
        m = re.search(re.escape(k) + r": (\d+)", b)
        return m.group(1) if m else "?"
    print("%-44s SGPR %4s VGPR %4s scratch %5s occ %2s LDS %s" % (
        name[:44], g("TotalSGPRs"), g("VGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"),
        g("LDS Size [bytes/block]")))
