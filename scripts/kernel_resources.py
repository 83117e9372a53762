#!/usr/bin/env python3
"""Prints registers, occupancy, scratch and LDS of every render_kernel variant from the compiler's
-Rpass-analysis=kernel-resource-usage remarks (the file `make -C raytracer_amd/csrc asm` writes)."""
import re
import sys

text = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/rt_api.resource.txt").read()
for block in re.split(r"Function Name: ", text)[1:]:
    m = re.match(r"_ZN2rt13render_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)", block.split()[0])
    if not m:
        continue
    get = lambda key: re.search(key + r": (\d+)", block).group(1)
    print("IMODE %s RMODE %s PRIMS %s STAGE %s:" % m.groups(), "VGPRs", get("VGPRs"), "SGPRs", get("TotalSGPRs"),
          "waves/SIMD", get(r"Occupancy \[waves/SIMD\]"), "scratch", get(r"ScratchSize \[bytes/lane\]"))
