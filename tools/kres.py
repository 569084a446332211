#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: python tools/kres.py build.log"""
import re, subprocess, sys

txt = open(sys.argv[1]).read()
blocks = re.split(r"Function Name: ", txt)[1:]
for b in blocks:
    name = b.split(" ")[0]
    name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("mtgv::", "")
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    print("%-70s VGPR %4s AGPR %4s spill %3s scratch %4s occ %s lds %s" % (
        name[:70], g("    VGPRs"), g("AGPRs"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
