#!/usr/bin/env python3
"""Registers / scratch / occupancy of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage, device only).  usage: kernel_resources.py csrc/mbwq.hip [filter]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "--offload-device-only", "-c", src,
                        "-o", os.path.join(d, "x.o"), "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:], capture_output=True, text=True)
blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
names = [b.split("\n")[0].split(" [")[0] for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
g = lambda pat, b: (re.search(pat, b) or [None, "?"])[1]
PATS = {"VGPR": r" VGPRs: (\d+)", "SGPR": r"TotalSGPRs: (\d+)", "scratch": r"ScratchSize \[bytes/lane\]: (\d+)", "occ": r"Occupancy \[waves/SIMD\]: (\d+)",
        "vspill": r"VGPRs Spill: (\d+)", "sspill": r"SGPRs Spill: (\d+)"}
for b, dn in zip(blocks, dem):
    if flt in dn:
        print(dn[:64].replace("bie::", "").replace("void ", "").ljust(64), "  ".join(k + " " + g(p_, b).rjust(3) for k, p_ in PATS.items()))
