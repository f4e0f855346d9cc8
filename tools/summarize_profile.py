#!/usr/bin/env python3
"""Turns a tools/profile_bench.sh output directory into the committed summaries under profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command
  profiles/<tag>_bench_line.json    the line bench.py printed under the profiler
  profiles/<round>_counters.json    [workload key] -> PMC counters of the dominant kernel, normalised PER FRAME, with the
                                    hash of the kernel source + build flags they were taken from (bench.py checks it)
A frame = width x height x spp primary rays.  A dispatch of G work-items renders G / (work-items per frame) frames (one 64-thread
workgroup per 8x8 tile per frame of the batch; the extension kernel: 256 threads per 16x16 tile per sample), so counters are summed over every dispatch of
the kernel and divided by the frames those dispatches rendered: the result does not depend on --steps or frames per launch.
   python tools/summarize_profile.py <profdir> <tag>"""
import collections, csv, glob, json, os, shutil, sys

import importlib
src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_b = importlib.import_module("cuda-raytracing_amd._build")
# the hash COMPILED INTO the library that travelled to the GPU box and ran (rt_build_info); it must be the build of this tree
code_hash = _b.library_code_hash()
assert code_hash == _b.kernel_code_hash(), "librt_hip.so (%s) is not the build of the sources in the tree (%s)" % (code_hash, _b.kernel_code_hash())
dst = os.path.join(ROOT, "profiles")
import bench
bench_counters = bench.COUNTERS_JSON
bench_line = None
for line in open(os.path.join(src, "stats.log"), errors="ignore"):
    if line.startswith("{\"metric\""):
        bench_line = json.loads(line)
assert bench_line, "no bench line in stats.log"
cfg = bench_line["config"]
W, H, key = cfg["width"], cfg["height"], cfg["key"]
kernel = bench_line["roofline"]["kernel"].replace(",", ", ")
if kernel == "render_kernel<false, false>":                 # (lines printed before the ORDERED template parameter existed)
    kernel = "render_kernel<false, false, false>"
if kernel.startswith("render_kernel<") and kernel.count(",") == 2:      # (lines printed before the SPILL parameter existed: either form)
    kernel = kernel[:-1] + ", "
if kernel.startswith("render_kernel<") and kernel.endswith(">"):        # (round 6: a sixth template parameter, STATS, follows the five the line names)
    kernel = kernel[:-1] + ","
# (round 5: the primary kernels' workgroup is one wave = one 8x8-pixel tile; the extension kernel's tile is still 16x16 pixels,
# rendered by four one-wave workgroups)
tiles = ((W + 15) // 16) * ((H + 15) // 16)
per_frame_items = tiles * 256 * cfg["spp"] if "render_ex" in kernel else ((W + 7) // 8) * ((H + 7) // 8) * 64
# render_ex_kernel<.., PX> (4 and more samples per pixel): a launch covers up to 64 samples of every pixel, a frame is
# ceil(spp / 64) launches of ceil(W / 2pw) x ceil(H / 2ph) workgroups -- count frames by dispatches, not by work-items
ex_launches_per_frame = (cfg["spp"] + 63) // 64 if ("render_ex" in kernel and cfg["spp"] >= 4) else 0
tot, meta, frames_by_pass = collections.defaultdict(float), {}, {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    fs = glob.glob(os.path.join(d, "*", "*_counter_collection.csv")) if os.path.isdir(d) else []
    if not fs:
        continue
    seen = {}
    for r in csv.DictReader(open(fs[0])):
        if kernel in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            seen[r["Dispatch_Id"]] = int(r["Grid_Size"])
            meta = {k: r[k] for k in ("Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size") if k in r}
    frames = len(seen) / ex_launches_per_frame if ex_launches_per_frame else sum(seen.values()) / per_frame_items
    for r in csv.DictReader(open(fs[0])):
        if kernel in r["Kernel_Name"]:
            frames_by_pass[r["Counter_Name"]] = frames
pf = {k: v / frames_by_pass[k] for k, v in tot.items() if frames_by_pass.get(k)}
entry = {"tag": tag, "code_hash": code_hash, "kernel": (kernel[:-1] + ", false>") if kernel.endswith(",") else kernel, "dispatch": meta, "frames_profiled": {k: round(v, 2) for k, v in frames_by_pass.items()},
         "per_frame": pf, "bench_line_under_rocprof": {k: bench_line[k] for k in ("value", "ms_per_step", "steps")}}
if "SQ_INSTS_VALU" in pf:
    entry["valu_insts_per_frame"] = pf["SQ_INSTS_VALU"]
if "SQ_THREAD_CYCLES_VALU" in pf and "SQ_ACTIVE_INST_VALU" in pf:
    entry["lanes_active_per_valu"] = round(pf["SQ_THREAD_CYCLES_VALU"] / pf["SQ_ACTIVE_INST_VALU"], 2)
if "TCP_TOTAL_CACHE_ACCESSES_sum" in pf:
    entry["tcp_accesses_per_frame"] = pf["TCP_TOTAL_CACHE_ACCESSES_sum"]
if "FETCH_SIZE" in pf and "WRITE_SIZE" in pf:
    # KiB units.  No x2 on FETCH_SIZE: the render kernels read 64-B records at unrelated addresses (four 16-B loads per lane),
    # and for that pattern tools/fetch_calibration.sh measures 0.99 bytes reported per byte read (profiles/r04_fetch_calibration.json, round 4);
    # the x2 of MI355X_MICROARCH.md "HBM" is for 16 B-per-lane streaming reads (0.50 measured with the same tool).  Both
    # counters sit on the fabric side of the L2s: Infinity Cache hits are included.
    entry.update({"hbm_bytes_per_frame": (pf["FETCH_SIZE"] + pf["WRITE_SIZE"]) * 1024, "fetch_bytes_per_frame": pf["FETCH_SIZE"] * 1024,
                  "write_bytes_per_frame": pf["WRITE_SIZE"] * 1024, "fetch_size_correction": 1.0})
if "GRBM_GUI_ACTIVE" in pf and "SQ_INSTS_VALU" in pf:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; a wave64 VALU instruction holds a SIMD-32 for 2 cycles
    entry["valu_issue_frac_under_profiler"] = round(pf["SQ_INSTS_VALU"] * 2 / (1024 * pf["GRBM_GUI_ACTIVE"] / 8), 4)
# kernel-trace pass: durations of the dominant kernel per launch shape (a run also contains a few single-frame launches of
# the same kernel -- the frame checks -- which the plain --stats average mixes in)
kt = glob.glob(os.path.join(src, "stats", "*", "*_kernel_trace.csv"))
if kt:
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        if kernel in r["Kernel_Name"]:
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            groups[grid].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    entry["kernel_trace"] = [{"grid_work_items": g, "frames_per_launch": round(1.0 / ex_launches_per_frame if ex_launches_per_frame else g / per_frame_items, 3), "launches": len(v),
                              "avg_ms": round(sum(v) / len(v), 4), "min_ms": round(min(v), 4), "max_ms": round(max(v), 4)}
                             for g, v in sorted(groups.items(), key=lambda kv: -len(kv[1]))]
ks = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(dst, "%s_kernel_stats.csv" % tag))
tp = os.path.join(dst, "%s_counters.json" % tag.split("_")[0])       # (the round is the tag's prefix: r06_c2_mid -> r06_counters.json)
assert os.path.abspath(tp) == os.path.abspath(bench_counters), "bench.py prices its lines with %s: the tag's round must match" % bench_counters
out = json.load(open(tp)) if os.path.exists(tp) else {}
out[key] = entry
json.dump(out, open(tp, "w"), indent=1, sort_keys=True)
# The line was printed before these counters existed: price its roofline block now, with bench.py's own function, from the
# kernel time the line measured under the profiler and the counters of the same profiling run.
import bench
old = bench_line["roofline"]
alg = old.get("hbm_algorithmic", {}).get("bytes_per_launch")
share = 1.0 / bench_line.get("n_gpus", 1)
roof = bench.roofline(old["kernel"], key, old["kernel_ms"], old["frames_per_launch"], share,
                      None if alg is None else alg / (old["frames_per_launch"] * share))
roof["filled_by"] = "tools/summarize_profile.py from the PMC passes of the same profiling run"
bench_line["roofline"] = roof
json.dump(bench_line, open(os.path.join(dst, "%s_bench_line.json" % tag), "w"), indent=1)
print(key, json.dumps({k: entry.get(k) for k in ("valu_insts_per_frame", "lanes_active_per_valu", "tcp_accesses_per_frame", "hbm_bytes_per_frame",
                                                  "valu_issue_frac_under_profiler", "frames_profiled")}, indent=0))
