#!/usr/bin/env python3
"""Per-dispatch PMC counters of the LAST frame's launches from rocprofv3 --pmc CSV output directories.
   python tools/pmc_sequence.py <dir> [<dir> ...] -- prints one row per dispatch (kernel, grid) with every counter found."""
import csv, glob, os, sys, collections
rows = collections.OrderedDict()
names = []
for d in sys.argv[1:]:
    fs = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    if not fs:
        continue
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        k = int(r["Dispatch_Id"])
        e = disp.setdefault(k, {"kernel": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "c": collections.defaultdict(float)})
        e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    seq = [e for e in disp.values() if "ex_wave" in e["kernel"] or "render_ex" in e["kernel"] or "resolve" in e["kernel"]]
    # the last frame = after the last-but-one resolve ... keep it simple: index launches from the end
    for i, e in enumerate(reversed(seq)):
        row = rows.setdefault(i, {"kernel": e["kernel"], "grid": e["grid"]})
        for n, v in e["c"].items():
            row[n] = v
            if n not in names:
                names.append(n)
keys = sorted(rows.keys(), reverse=True)[-int(os.environ.get("LAST", "40")):]
print("%-28s %9s " % ("kernel", "wgs") + " ".join("%14s" % n[:14] for n in names))
for k in keys:
    r = rows[k]
    kn = r["kernel"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:28]
    print("%-28s %9d " % (kn, r["grid"] // 256) + " ".join("%14.4g" % r.get(n, float("nan")) for n in names))
