"""bench.py prices `roofline.frac` with instruction counts from a committed rocprofv3 PMC pass (profiles/r04_counters.json).
Those counts describe one build of the kernels: every entry carries the hash of the kernel source + build flags it was taken
from, and a line printed by other code says `profile_stale` instead of a fraction (CPU-only: no kernel runs here)."""
import glob
import importlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
build = importlib.import_module("cuda-raytracing_amd._build")


def test_one_character_kernel_edit_marks_profile_stale(tmp_path, monkeypatch):
    srcs = [os.path.join(build.CSRC, n) for n in ("rt_kernels.hip", "rt_math.h", "rt_device_types.h")]
    copies = [shutil.copy(p, tmp_path / os.path.basename(p)) for p in srcs]
    h0 = build.kernel_code_hash()
    assert build.kernel_code_hash(sources=copies) == h0                 # the hash is of content, not of paths or times
    text = open(copies[0]).read()
    at = text.index("constexpr int kLdsStack = 16;")
    open(copies[0], "w").write(text[:at] + text[at:].replace("16", "17", 1))   # one character
    h1 = build.kernel_code_hash(sources=copies)
    assert h1 != h0
    assert build.kernel_code_hash(flags=[f for f in build.HIP_FLAGS if f != "-fno-slp-vectorize"]) != h0    # flags count too

    import bench
    table = tmp_path / "counters.json"
    json.dump({"k": {"tag": "t", "code_hash": h0, "valu_insts_per_frame": 1.159e8, "lanes_active_per_valu": 46.7,
                     "tcp_accesses_per_frame": 5.44e7, "hbm_bytes_per_frame": 9.4e6}}, open(table, "w"))
    monkeypatch.setattr(bench, "COUNTERS_JSON", str(table))
    fresh = bench.roofline("render_kernel<false,false,false>", "k", 4.228, 32, 1.0, code_hash=h0)
    assert fresh["profile_stale"] is False and 0.5 < fresh["frac"] < 1.0 and fresh["achieved"] is not None and fresh["code_hash"] == h0
    stale = bench.roofline("render_kernel<false,false,false>", "k", 4.228, 32, 1.0, code_hash=h1)
    assert stale["profile_stale"] is True and stale["frac"] is None and stale["achieved"] is None and stale["traffic"] is None
    assert stale["kernel_ms"] == fresh["kernel_ms"] and stale["profile_code_hash"] == h0    # the live time is still reported
    missing = bench.roofline("render_kernel<false,false,false>", "other", 4.228, 32, 1.0, code_hash=h0)
    assert missing["frac"] is None and missing["profile_stale"] is False


def test_committed_counters_carry_their_code_hash():
    path = os.path.join(ROOT, "profiles", "r04_counters.json")
    table = json.load(open(path))
    assert table, "no committed PMC profile"
    for key, e in table.items():
        assert isinstance(e.get("code_hash"), str) and len(e["code_hash"]) == 16, key
        assert e.get("valu_insts_per_frame", 0) > 0, key
    # every bench line committed beside the counters was priced (tools/summarize_profile.py fills the block from the same run)
    for f in glob.glob(os.path.join(ROOT, "profiles", "r04_*_bench_line.json")):
        roof = json.load(open(f))["roofline"]
        assert roof["frac"] is not None and roof["profile_stale"] is False, f



def test_headline_profile_is_of_the_code_in_the_tree():
    """The driver runs bench.py on this tree: the committed PMC profile of the headline workload (c2, mid camera) must have been
    taken from the kernels as they are now, or the line would say `profile_stale` and carry no roofline fraction.  A kernel
    edit therefore fails here until tools/profile_bench.sh + tools/summarize_profile.py have been run on the new code."""
    table = json.load(open(os.path.join(ROOT, "profiles", "r04_counters.json")))
    assert table["c2_mid_1920x1080_1_0_0"]["code_hash"] == build.kernel_code_hash()
