#!/usr/bin/env python3
"""Prints the launch sequence of the last frame of a rocprofv3 --kernel-trace run (CSV): name, grid, start, duration.
   python tools/trace_sequence.py <dir with *_kernel_trace.csv> [kernel-name filter] [launches]"""
import csv, glob, os, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
n = int(sys.argv[3]) if len(sys.argv) > 3 else 45
fs = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)
assert fs, "no kernel trace under " + src
rows = [r for r in csv.DictReader(open(fs[0])) if flt in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    print("%-58s wg %8d  start %10.1f us  gap %7.1f  dur %10.1f us" % (r["Kernel_Name"][:58], grid // 256, (s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3))
    prev_end = e
