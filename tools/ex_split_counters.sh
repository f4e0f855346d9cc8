#!/bin/bash
# PMC passes of c3 in the two-launch bounce form (RT_EX_SPLIT=1), per kernel, beside the one-kernel form's entry of the round's
# counters file: profiles/r06_experiments/ex_split_two_launches.md.   bash tools/ex_split_counters.sh <outdir>
out=$1; export TMPDIR=/tmp; mkdir -p $out; cd $GRAFT_REPO_ROOT
export RT_EX_SPLIT=1
pmc() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $out/$name.log 2> $out/$name.err; echo "$name rc=$?"; }
pmc pmc_fetch FETCH_SIZE && pmc pmc_write WRITE_SIZE && pmc pmc_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY \
 && pmc pmc_sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH
python3 - <<PY
import csv, glob, collections, json
out = "$out"
res = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(out + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "render_ex_kernel" not in k: continue
        phase = "phase 1 (camera ray)" if ", 1>" in k else ("phase 2 (rest of the path)" if ", 2>" in k else "one kernel")
        res[phase][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(phase, r["Counter_Name"])].add(r["Dispatch_Id"])
summary = {ph: {c: v / max(len(disp[(ph, c)]), 1) for c, v in cs.items()} for ph, cs in res.items()}
json.dump(summary, open(out + "/ex_split_counters_per_frame.json", "w"), indent=1, sort_keys=True)
for ph, cs in summary.items():
    print(ph, {c: ("%.3g" % v) for c, v in sorted(cs.items())})
PY
