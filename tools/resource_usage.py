"""Summarise -Rpass-analysis=kernel-resource-usage output (python -m gaussianvi_amd.build -v 2> file): one line per kernel."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: Function Name: ", txt)[1:]
rows = []
for b in blocks:
    name = b.split()[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    rows.append((dem, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
for r in rows:
    if pat in r[0]:
        print(f"{r[0][:110]:110s} VGPR {r[1]:>3} AGPR {r[2]:>3} SGPR {r[3]:>3} scratch {r[4]:>4} occ {r[5]} lds {r[6]}")
