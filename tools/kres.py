#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr file)."""
import re, subprocess, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0].strip(" []")
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("void sefd::", "").replace("sefd::", "")
    if flt not in dem:
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    sc, lds, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{dem[:120]:120s} VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} scratch {sc:>4} LDS {lds:>6} occ {occ}")
