#!/usr/bin/env python3
"""Turns a tools/profile_bench.sh output directory into the committed summaries under profiles/.
   python tools/summarize_profile.py <profdir> <tag> <workload-key>"""
import collections, csv, glob, json, os, shutil, sys
src, tag, key = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles")
summ, meta = {}, {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    fs = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "render_kernel<false, false>" in r["Kernel_Name"] and int(r["Grid_Size"]) > 256 * 1000:
            acc[r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
            meta = {k: r[k] for k in ("Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size")}
    for k, v in acc.items():
        big = max(g for g, _ in v)                      # the batched launches (largest grid), not the single-frame probes
        vals = [x for g, x in v if g == big]
        summ[k] = {"mean": sum(vals) / len(vals), "min": min(vals), "max": max(vals), "dispatches": len(vals), "grid": big}
ks = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(dst, "%s_kernel_stats.csv" % tag))
bench_line = None
for line in open(os.path.join(src, "stats.log"), errors="ignore"):
    if line.startswith("{\"metric\""):
        bench_line = json.loads(line)
tp = os.path.join(dst, "r01_traffic.json")
out = json.load(open(tp)) if os.path.exists(tp) else {}
fetch, write = summ.get("FETCH_SIZE", {}).get("mean"), summ.get("WRITE_SIZE", {}).get("mean")
entry = {"tag": tag, "dispatch": meta, "counters": summ, "bench_line_under_rocprof": bench_line}
if fetch is not None and write is not None:
    entry.update({"hbm_bytes_per_launch": int((2 * fetch + write) * 1024), "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write})
out[key] = entry
json.dump(out, open(tp, "w"), indent=1)
print(json.dumps({k: (v["mean"] if isinstance(v, dict) else v) for k, v in summ.items()}, indent=0))
print("hbm_bytes_per_launch", entry.get("hbm_bytes_per_launch"))
