#!/bin/bash
# Per-lane extension kernel against its wavefront form (RT_EX_WAVEFRONT=1: one cast per launch, ballot / popc compaction of the live
# paths in between) on one workload, with the counters that say why: time, VALU instructions, active lanes per VALU instruction,
# fabric-side bytes.  On the GPU box:   bash tools/ex_wavefront_ab.sh <outdir> [bench args...]
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
cd $GRAFT_REPO_ROOT
ARGS="$*"
for form in perlane wavefront; do
    [ $form = wavefront ] && export RT_EX_WAVEFRONT=1 || unset RT_EX_WAVEFRONT
    echo "== $form: bench.py $ARGS" | tee -a $out/progress.log
    timeout -k 10 500 python3 bench.py $ARGS --steps 5 --warmup 2 --no-cpu-baseline > $out/${form}_line.json 2> $out/${form}.err || { tail -5 $out/${form}.err; exit 1; }
    pmc() { name=$1; shift; echo "== $form $name" | tee -a $out/progress.log
            timeout -k 10 500 rocprofv3 --pmc "$@" --output-format csv -d $out/${form}_$name -- python3 bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline > $out/${form}_$name.log 2> $out/${form}_$name.err; }
    pmc sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE || exit 1
    pmc fetch FETCH_SIZE || exit 1
    pmc write WRITE_SIZE || exit 1
done
unset RT_EX_WAVEFRONT
python3 - $out <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for form in ("perlane", "wavefront"):
    line = json.load(open("%s/%s_line.json" % (out, form)))
    e = {"ms_per_frame": line["ms_per_step"], "kernel_ms": line["roofline"]["kernel_ms"], "frame_matches": line.get("frame_matches_single_gpu_render"),
         "key": line["config"]["key"], "code_hash": line["roofline"]["code_hash"], "counters_per_frame": {}, "kernels": {}}
    for name in ("sq", "fetch", "write"):
        fs = glob.glob("%s/%s_%s/*/*_counter_collection.csv" % (out, form, name))
        if not fs:
            continue
        tot, per_kernel = {}, {}
        for r in csv.DictReader(open(fs[0])):
            k = r["Kernel_Name"]
            if not any(t in k for t in ("render_ex_kernel", "ex_wave_kernel", "resolve_ex_kernel")):
                continue
            short = k.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            per_kernel.setdefault(short, {}).setdefault(r["Counter_Name"], 0.0)
            per_kernel[short][r["Counter_Name"]] += float(r["Counter_Value"])
        frames = 3 + 1        # the PMC runs render warm-up 1 + steps 2 + the kernel-time probe (3 more: max(3, steps)) + the whole-frame check ... counted below
        e["counters_per_frame"].update(tot)
        for k, v in per_kernel.items():
            e["kernels"].setdefault(k, {}).update(v)
    res[form] = e
# frames rendered by a PMC run: warm-up 1 + timed 2 + kernel-time probe max(3, 2) + 1 whole-frame check = 7 extension frames
for form, e in res.items():
    e["frames_in_pmc_run"] = 7
    c = {k: v / 7.0 for k, v in e.pop("counters_per_frame").items()}
    e["per_frame"] = c
    if c.get("SQ_ACTIVE_INST_VALU"):
        e["lanes_active_per_valu"] = round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"], 2)
    if c.get("SQ_INSTS_VALU"):
        e["valu_issue_frac"] = round(c["SQ_INSTS_VALU"] / (e["kernel_ms"] * 1e-3) / 1e9 / 1228.8, 4)
    if "FETCH_SIZE" in c:
        e["fetch_GB_per_frame"] = round(c["FETCH_SIZE"] * 1024 / 1e9, 3)
    if "WRITE_SIZE" in c:
        e["write_GB_per_frame"] = round(c["WRITE_SIZE"] * 1024 / 1e9, 3)
json.dump(res, open(out + "/ex_wavefront_ab.json", "w"), indent=1)
for form, e in res.items():
    print(form, {k: e.get(k) for k in ("ms_per_frame", "kernel_ms", "lanes_active_per_valu", "valu_issue_frac", "fetch_GB_per_frame", "write_GB_per_frame", "frame_matches")},
          "VALU G/frame", round(e["per_frame"].get("SQ_INSTS_VALU", 0) / 1e9, 3))
PY
