#!/bin/bash
# Vector-L1 (TCP) access-rate calibration for the traversal's fetch (see tools/l1_calibration.hip).  On the GPU box:
#   bash tools/l1_calibration.sh <outdir>      -> <outdir>/l1_calibration.json
# One plain run for the times, then PMC passes of the same binary (counters only, never combined with a trace).
out=$1; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $out/l1_calibration tools/l1_calibration.hip || exit 1
timeout -k 10 200 $out/l1_calibration > $out/times.json 2> $out/run.err || exit 1
timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD --output-format csv -d $out/pmc_tcp -- $out/l1_calibration > /dev/null 2>> $out/run.err || exit 1
timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_tcp2 -- $out/l1_calibration > /dev/null 2>> $out/run.err
python3 - $out <<'PY'
import csv, glob, json, re, sys
out = sys.argv[1]
runs = json.loads(open(out + "/times.json").read())["launches"]
info = open(out + "/run.err").read()
m = re.search(r"cus=(\d+) clock_khz=(\d+) wall_clock_khz=(\d+)", info)
cus, clock_khz, wall_khz = (int(v) for v in m.groups())
def counters(d):
    fs = glob.glob(out + "/" + d + "/*/*_counter_collection.csv")
    rows = {}
    for r in (csv.DictReader(open(fs[0])) if fs else []):
        k = re.search(r"gather_l1_kernel<(\d+)>", r["Kernel_Name"])
        if k:
            rows.setdefault(int(k.group(1)), {}).setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return {D: v[max(v)] for D, v in rows.items()}            # per D: warm-up launch first, timed launch second
pm = counters("pmc_tcp")
pm2 = counters("pmc_tcp2")
res = {"device": {"cus": cus, "clock_khz": clock_khz, "wall_clock_khz": wall_khz},
       "what": "four global_load_dwordx4 per lane and record from an L1-resident 16 KiB table, D different 64-B records per wave-instruction, "
               "8 waves per SIMD on every CU (tools/l1_calibration.hip)", "launches": []}
for r in runs:
    D = r["distinct_records_per_wave_instruction"]
    sec = r["timed_launch_ms"] * 1e-3
    e = dict(r)
    c = dict(pm.get(D, {}), **pm2.get(D, {}))
    e["counters"] = c
    e["wave_load_instructions_per_clk_per_cu_at_%d_MHz" % (clock_khz // 1000)] = round(r["wave_load_instructions"] / sec / (clock_khz * 1e3) / cus, 4)
    # s_memtime: which clock it counts is read off the constant-rate counter beside it
    e["memtime_per_realtime_tick"] = round(r["mean_wave_loop_memtime"] / max(r["mean_wave_loop_realtime_ticks"], 1), 4)
    e["loop_seconds_by_realtime"] = r["mean_wave_loop_realtime_ticks"] / (wall_khz * 1e3)
    acc = c.get("TCP_TOTAL_CACHE_ACCESSES_sum")
    if acc:
        e["tcp_accesses_per_wave_load_instruction"] = round(acc / r["wave_load_instructions"], 4)
        e["tcp_accesses_per_second"] = acc / sec
        e["tcp_accesses_per_clk_per_cu_at_%d_MHz" % (clock_khz // 1000)] = round(acc / sec / (clock_khz * 1e3) / cus, 4)
        e["tcp_GBs_at_64B_per_access"] = round(acc * 64 / sec / 1e9, 1)
    res["launches"].append(e)
best = max((e for e in res["launches"] if "tcp_accesses_per_second" in e), key=lambda e: e["tcp_accesses_per_second"], default=None)
if best:
    res["ceiling"] = {"tcp_accesses_per_second": best["tcp_accesses_per_second"], "at_distinct_records": best["distinct_records_per_wave_instruction"],
                      "tcp_GBs_at_64B_per_access": best["tcp_GBs_at_64B_per_access"],
                      "assumed_until_round_4_GBs": round(cus * 64 * clock_khz * 1e3 / 1e9, 1)}
json.dump(res, open(out + "/l1_calibration.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "launches"}, indent=1))
for e in res["launches"]:
    print(e["distinct_records_per_wave_instruction"], e.get("tcp_accesses_per_wave_load_instruction"), e.get("tcp_GBs_at_64B_per_access"), e["timed_launch_ms"])
PY
