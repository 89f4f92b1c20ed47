#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS use from `make -C blackbird_amd/csrc resource-usage` (stderr on stdin or a file).
usage: make -C blackbird_amd/csrc resource-usage 2>&1 | python tools/model/resource_usage.py [name filter ...]"""
import re
import subprocess
import sys

txt = sys.stdin.read()
filt = sys.argv[1:] or ["k_selfplay_queue", "k_dc_selfplay_fused", "k_net_x3", "k_gnet_conv_x3"]
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split(" ")[0]
    if not any(k in name for k in filt):
        continue
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    except OSError:
        pass

    def f(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    print("%-70s VGPR %3s AGPR %3s scratch %4s occ %s vspill %3s sspill %3s LDS %6s" % (
        name[-70:], f("VGPRs"), f("AGPRs"), f(r"ScratchSize \[bytes/lane\]"), f(r"Occupancy \[waves/SIMD\]"), f("VGPRs Spill"),
        f("SGPRs Spill"), f(r"LDS Size \[bytes/block\]")))
