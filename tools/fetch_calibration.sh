#!/bin/bash
# FETCH_SIZE calibration for the traversal's access pattern (see tools/fetch_calibration.hip).  On the GPU box:
#   bash tools/fetch_calibration.sh <outdir>      -> <outdir>/fetch_calibration.json
out=$1; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $out/fetch_calibration tools/fetch_calibration.hip || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $out/fetch_calibration > $out/sizes.json 2> $out/run.err || exit 1
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/pmc_rdreq -- $out/fetch_calibration > /dev/null 2>> $out/run.err
python3 - $out <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
sizes = json.loads([l for l in open(out + "/sizes.json") if l.startswith("{")][-1])
def counters(d):
    rows = {}
    fs = glob.glob(out + "/" + d + "/*/*_counter_collection.csv")
    for r in (csv.DictReader(open(fs[0])) if fs else []):
        if "stream_kernel" in r["Kernel_Name"] or "gather_kernel" in r["Kernel_Name"]:
            rows.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [v for k, v in sorted(rows.items())]
f, q = counters("pmc_fetch"), counters("pmc_rdreq")
names = ["stream", "gather_cold", "gather_warm"]
known = [sizes["stream_bytes"], sizes["gather_cold_record_bytes"] + sizes["gather_cold_index_bytes"], sizes["gather_warm_record_bytes_algorithmic"] + sizes["gather_warm_index_bytes"]]
res = {"sizes": sizes, "launches": {}}
for i, n in enumerate(names):
    e = {"known_bytes": known[i]}
    if i < len(f):
        e["FETCH_SIZE_KiB"] = f[i].get("FETCH_SIZE")
        e["FETCH_SIZE_bytes_over_known"] = round(f[i].get("FETCH_SIZE", 0) * 1024 / known[i], 4)
    if i < len(q):
        e.update({k: v for k, v in q[i].items()})
    res["launches"][n] = e
json.dump(res, open(out + "/fetch_calibration.json", "w"), indent=1)
print(json.dumps(res["launches"], indent=1))
PY
