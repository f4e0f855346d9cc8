#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats of the driver's exact command (`python3 bench.py --gpus 1 --steps 20 --warmup 5`, run by
tools/profile_campaign.sh <out> aux) -> profiles/<tag>_driver_command_kernel_stats.csv and _kernel_trace.json: the durations of
the command's 20-frame launches beside the kernel time the line itself reports.   python tools/driver_command_trace.py <out>/driver r06"""
import collections, csv, glob, json, os, shutil, statistics, sys
src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
line = json.loads([l for l in open(os.path.join(src, "driver_line.json")) if l.startswith("{")][-1])
kt = glob.glob(os.path.join(src, "stats", "*", "*_kernel_trace.csv"))[0]
ks = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
groups = collections.defaultdict(list)
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"]))
# the latency block starts with the first single-frame heavy-first launch (ORDERED = the third template argument); the 20-frame
# launches before it are the warm-up, the nine timed repeats and the hipEvent probe, the ones after it the PCIe-inclusive section
latency_start = min([int(r["Start_Timestamp"]) for r in rows if "render_kernel<false, false, true" in r["Kernel_Name"]] or [1 << 62])
main_section = []
for r in rows:
    if "render_kernel<false, false, false" in r["Kernel_Name"] and not r["Kernel_Name"].rstrip(")").split("(")[0].rstrip().endswith("true>"):
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        groups[grid].append(ms)
        if int(r["Start_Timestamp"]) < latency_start:
            main_section.append((grid, ms))
per_frame = ((line["config"]["width"] + 7) // 8) * ((line["config"]["height"] + 7) // 8) * 64          # (round 5: one 64-thread workgroup per 8x8-pixel tile)
out = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5",
       "line": {k: line.get(k) for k in ("value", "ms_per_step", "ms_per_step_min", "ms_per_step_max", "repeats")},
       "line_kernel_ms_per_launch": line["roofline"]["kernel_ms"], "line_frames_per_launch": line["roofline"]["frames_per_launch"],
       "code_hash": line["roofline"]["code_hash"], "launch_shapes": []}
for grid, v in sorted(groups.items(), key=lambda kv: -len(kv[1])):
    out["launch_shapes"].append({"frames_per_launch": round(grid / per_frame, 3), "launches": len(v), "avg_ms": round(sum(v) / len(v), 4),
                                 "median_ms": round(statistics.median(v), 4), "min_ms": round(min(v), 4), "max_ms": round(max(v), 4)})
big = max(groups, key=lambda g: g) if groups else 0
v = [ms for g, ms in main_section if g == big]
if v:
    out["timed_section"] = {"what": "the %g-frame launches before the latency block: warm-up, the nine timed repeats, the hipEvent probe the line's kernel_ms averages" % round(big / per_frame, 3),
                            "launches": len(v), "avg_ms": round(sum(v) / len(v), 4), "median_ms": round(statistics.median(v), 4), "min_ms": round(min(v), 4), "max_ms": round(max(v), 4)}
out["note"] = ("the command's 20-frame launches are: the warm-up, the nine timed repeats, the hipEvent probe (what the line's kernel_ms averages) and the "
               "PCIe-inclusive measurement, whose launches run beside a device-to-host copy of the previous batch (the slow ones); the median is the figure to compare with the line")
shutil.copy(ks, os.path.join(ROOT, "profiles", "%s_driver_command_kernel_stats.csv" % tag))
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_driver_command_kernel_trace.json" % tag), "w"), indent=1)
# (the line printed UNDER the profiler: its own file -- profiles/<tag>_final_bench_driver.json is the line of tools/final_lines.sh, taken without one)
json.dump(line, open(os.path.join(ROOT, "profiles", "%s_driver_command_bench_line_under_rocprof.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
