#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage from a hipcc build log made with -Rpass-analysis=kernel-resource-usage.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c backend_bls12381.hip 2> log; python tools/kernel_resources.py log"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", t)[1:]
print("%-78s %5s %5s %8s %5s %7s" % ("kernel", "vgpr", "agpr", "scratch", "occ", "lds"))
for b in blocks:
    name = b.split()[0]

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = re.sub(r"\(.*", "", name).replace("void apk::", "")
    print("%-78s %5d %5d %8d %5d %7d" % (name[:78], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
