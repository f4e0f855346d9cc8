#!/bin/bash
# PMC counters of the bounce kernel (c3) for each variant library under cuda-raytracing_amd/_variants: bash tools/ab_ex_counters.sh <outdir>
out=$1; export TMPDIR=/tmp; mkdir -p $out; cd $GRAFT_REPO_ROOT
cp cuda-raytracing_amd/librt_hip.so $out/librt_hip_saved.so
export RT_ALLOW_VARIANT_LIB=1
trap 'cp $out/librt_hip_saved.so cuda-raytracing_amd/librt_hip.so; touch cuda-raytracing_amd/librt_hip.so cuda-raytracing_amd/librt_host.so; rm -f $out/librt_hip_saved.so' EXIT
for lib in cuda-raytracing_amd/_variants/librt_hip_*.so; do
  name=$(basename $lib .so); name=${name#librt_hip_}
  cp $lib cuda-raytracing_amd/librt_hip.so; touch cuda-raytracing_amd/librt_hip.so cuda-raytracing_amd/librt_host.so
  pmc() { g=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $out/$name/$g -- python3 bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $out/$name.$g.log 2> $out/$name.$g.err; echo "$name $g rc=$?"; }
  pmc write WRITE_SIZE && pmc fetch FETCH_SIZE && pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY \
   && pmc sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH && pmc sq3 SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_WAIT_INST_LDS
done
python3 - <<PY
import csv, glob, collections, json, os
out = "$out"
res = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(out + "/*/*/*/*_counter_collection.csv"):
    name = f[len(out) + 1:].split("/")[0]
    for r in csv.DictReader(open(f)):
        if "render_ex_kernel" not in r["Kernel_Name"]: continue
        res[name][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(name, r["Counter_Name"])].add(r["Dispatch_Id"])
summary = {n: {c: v / max(len(disp[(n, c)]), 1) for c, v in cs.items()} for n, cs in res.items()}
json.dump(summary, open(out + "/counters_per_frame.json", "w"), indent=1, sort_keys=True)
names = sorted(summary)
for c in sorted({c for n in names for c in summary[n]}):
    print("%-24s" % c, "  ".join("%s %.4g" % (n, summary[n].get(c, float("nan"))) for n in names))
PY
