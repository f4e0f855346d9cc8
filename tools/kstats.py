#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of the HIP sources as hipcc compiles them for gfx950
(-Rpass-analysis=kernel-resource-usage).  Usage: python tools/kstats.py [file.hip ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cuda-raytracing_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-c",
         "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null"]


def stats(src):
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src], capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: [^ ]+ (?:Function )?Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z ]*?)(?: \[[^\]]*\])?: (\d+) \[-R", line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    return rows


if __name__ == "__main__":
    files = sys.argv[1:] or [os.path.join(CSRC, n) for n in sorted(os.listdir(CSRC)) if n.endswith(".hip")]
    for f in files:
        for r in stats(f):
            name = re.sub(r"\(anonymous namespace\)::", "", r["name"]).split("(")[0]
            if "rocprim" in name:
                continue
            print("%-36s vgpr %3d (spilled %2d)  sgpr %3d (spilled %2d)  scratch %4d B/lane  occupancy %d" % (
                name[:36], r.get("VGPRs", -1), r.get("VGPRs Spill", 0), r.get("TotalSGPRs", -1), r.get("SGPRs Spill", 0),
                r.get("ScratchSize", -1), r.get("Occupancy", -1)))
