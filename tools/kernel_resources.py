#!/usr/bin/env python3
"""Registers, spills, scratch and occupancy of every kernel of rt_kernels.hip, from hipcc's own remarks
(-Rpass-analysis=kernel-resource-usage).   python tools/kernel_resources.py [filter] [extra hipcc flags...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as d:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-c",
           "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(d, "k.o"), os.path.join(ROOT, "cuda-raytracing_amd", "csrc", "rt_kernels.hip")] + sys.argv[2:]
    err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
cur = None
rows = {}
for ln in err.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip().replace("(anonymous namespace)::", "").replace("(rt::RenderParams)", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:(?: \S+:\d+:\d+:)?\s+(.+?): (\S+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    if flt in k:
        print("%-52s VGPR %3s spill %3s | SGPR %3s spill %3s | scratch %5s B | occ %s | LDS %s" % (k[:52], v.get("VGPRs"), v.get("VGPRs Spill"), v.get("TotalSGPRs"),
              v.get("SGPRs Spill"), v.get("ScratchSize [bytes/lane]"), v.get("Occupancy [waves/SIMD]"), v.get("LDS Size [bytes/block]")))
