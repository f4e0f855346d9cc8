#!/usr/bin/env python3
"""Per-level kernel durations of one rt_bvh_build from a rocprofv3 --kernel-trace CSV (the last build of the last mesh in
the trace): which levels the time goes to.   python tools/bvh_level_times.py <..._kernel_trace.csv> [launches per build]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["bins_kernel", "decide_kernel", "flags_scan_kernel", "scatter_kernel"]
# the last build: walk back from the end to the last prep_kernel
last = max(i for i, r in enumerate(rows) if "prep_kernel" in r["Kernel_Name"])
build = rows[last:]
per = {n: [] for n in names}
for r in build:
    for n in names:
        if n + "(" in r["Kernel_Name"]:
            per[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
span = (int(build[-1]["End_Timestamp"]) - int(build[0]["Start_Timestamp"])) / 1e3
busy = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in build)
print("last build: %d launches, first start to last end %.1f us, kernels busy %.1f us" % (len(build), span, busy))
print("level " + " ".join("%18s" % n for n in names))
for l in range(len(per["bins_kernel"])):
    print("%5d " % l + " ".join("%18.1f" % (per[n][l] if l < len(per[n]) else 0.0) for n in names))
print("sum   " + " ".join("%18.1f" % sum(per[n]) for n in names))
other = {}
for r in build:
    nm = r["Kernel_Name"].split("(")[1] if r["Kernel_Name"].startswith("(anonymous") else r["Kernel_Name"][:40]
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:48]
    if any(n == nm for n in names):
        continue
    other.setdefault(nm, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for nm, v in other.items():
    print("%-48s x%d  %.1f us" % (nm, len(v), sum(v)))
