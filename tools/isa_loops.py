#!/usr/bin/env python3
"""Loops of one kernel in hipcc's gfx950 assembly, with their instruction, scratch (spill) and memory-load counts: shows
whether register spills sit inside a hot loop.   python tools/isa_loops.py <kernel-name-substring> [file.hip]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "cuda-raytracing_amd", "csrc", "rt_kernels.hip")
asm = "/tmp/isa_loops.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                "--cuda-device-only", "-S", "-o", asm, src], check=True, stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(name), l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i, m.group(1)))
isn = lambda x: re.match(r"^\s+[a-z]", x) is not None
print("kernel %s: %d instructions, %d scratch ops" % (lines[start][:60], sum(map(isn, body)), sum("scratch_" in x for x in body)))
for a, b, t in sorted(loops):
    seg = body[a:b + 1]
    print("  loop %-10s lines %5d..%5d  instrs %4d  valu %4d  scratch %3d  vmem loads %3d  lds %3d  smem %3d" % (
        t, a, b, sum(map(isn, seg)), sum(re.match(r"^\s+v_", x) is not None for x in seg), sum("scratch_" in x for x in seg),
        sum(re.match(r"^\s+(global_load|flat_load|buffer_load)", x) is not None for x in seg),
        sum(re.match(r"^\s+ds_", x) is not None for x in seg), sum(re.match(r"^\s+s_load", x) is not None for x in seg)))
