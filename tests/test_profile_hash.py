"""bench.py prices `roofline.frac` with instruction counts from a committed rocprofv3 PMC pass (profiles/r06_counters.json).
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
    srcs = list(build.HIP_DEPS)                                          # every source of librt_hip.so (round 6: not the kernel unit alone)
    assert {os.path.basename(p) for p in srcs} >= {"rt_kernels.hip", "rt_bvh_build.hip", "rt_comm.hip", "rt_scene_internal.h", "rt_hip.h"}
    copies = [shutil.copy(p, tmp_path / os.path.basename(p)) for p in srcs]
    h0 = build.kernel_code_hash()
    assert build.kernel_code_hash(sources=copies) == h0                 # the hash is of content, not of paths or times
    text = open(copies[0]).read()
    at = text.index("constexpr int kLdsStack = 16;")
    open(copies[0], "w").write(text[:at] + text[at:].replace("16", "17", 1))   # one character
    h1 = build.kernel_code_hash(sources=copies)
    assert h1 != h0
    assert build.kernel_code_hash(flags=[f for f in build.HIP_FLAGS if f != "-fno-slp-vectorize"]) != h0    # flags count too
    # an edit to the scene layout the three translation units share changes it as well
    at = [i for i, p in enumerate(copies) if str(p).endswith("rt_scene_internal.h")][0]
    open(copies[at], "a").write("// edited\n")
    assert build.kernel_code_hash(sources=copies) not in (h0, h1)
    # sources that cannot be read: an error that says so (the loader turns it into RtError), not a bare FileNotFoundError
    import pytest
    with pytest.raises(build.BuildError):
        build.kernel_code_hash(sources=[str(tmp_path / "missing.hip")])

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
    path = os.path.join(ROOT, "profiles", "r06_counters.json")
    table = json.load(open(path))
    assert table, "no committed PMC profile"
    for key, e in table.items():
        assert isinstance(e.get("code_hash"), str) and len(e["code_hash"]) == 16, key
        assert e.get("valu_insts_per_frame", 0) > 0, key
    # every bench line committed beside the counters was priced (tools/summarize_profile.py fills the block from the same run)
    for f in glob.glob(os.path.join(ROOT, "profiles", "r06_*_bench_line.json")):
        roof = json.load(open(f))["roofline"]
        assert roof["frac"] is not None and roof["profile_stale"] is False, f



def test_headline_profile_is_of_the_code_in_the_tree():
    """The driver runs bench.py on this tree: the committed PMC profile of the headline workload (c2, mid camera) must have been
    taken from the kernels as they are now, or the line would say `profile_stale` and carry no roofline fraction.  A kernel
    edit therefore fails here until tools/profile_bench.sh + tools/summarize_profile.py have been run on the new code."""
    table = json.load(open(os.path.join(ROOT, "profiles", "r06_counters.json")))
    assert table["c2_mid_1920x1080_1_0_0"]["code_hash"] == build.kernel_code_hash()


def _variant_package(tmp_path):
    """A copy of the package whose librt_hip.so carries ANOTHER kernel code hash and is newer than every source -- what
    tools/ab_variants.sh leaves behind if it is interrupted, or a library built from an edited tree and copied in."""
    pkg = tmp_path / "cuda-raytracing_amd"
    shutil.copytree(os.path.join(ROOT, "cuda-raytracing_amd"), pkg, ignore=shutil.ignore_patterns("__pycache__", "_variants", ".build.lock"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")       # (include/rt_hip.h is one of the library's hashed sources)
    so = pkg / "librt_hip.so"
    data = so.read_bytes()
    real = build.library_code_hash()
    assert real == build.kernel_code_hash(), "the shipped library is not the build of this tree: run __graft_entry__.build()"
    fake = "0123456789abcdef"
    assert data.count(build.HASH_MARKER + real.encode()) == 1
    so.write_bytes(data.replace(build.HASH_MARKER + real.encode(), build.HASH_MARKER + fake.encode()))
    future = os.path.getmtime(os.path.join(build.CSRC, "rt_kernels.hip")) + 3600
    os.utime(so, (future, future))
    os.utime(pkg / "librt_host.so", (future + 1, future + 1))
    return pkg, fake, real


def test_variant_library_under_fresh_sources_is_noticed(tmp_path):
    """the guard hashed sources, and trusted the .so through file times."""
    import subprocess
    pkg, fake, real = _variant_package(tmp_path)
    assert build.library_code_hash(str(pkg / "librt_hip.so")) == fake
    why = build.library_mismatch(str(pkg / "librt_hip.so"))
    assert why is not None and fake in why and real in why
    prog = ("import importlib, sys; sys.path.insert(0, %r); rt = importlib.import_module('cuda-raytracing_amd'); "
            "print('HASH', rt.library_hash())" % str(tmp_path))
    # no compiler within reach: the loader refuses the library and says why
    env = dict(os.environ, ROCM_PATH=str(tmp_path / "no_rocm"))
    env.pop("RT_ALLOW_VARIANT_LIB", None)
    r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and fake in r.stderr and real in r.stderr and "HASH" not in r.stdout, r.stderr[-2000:]
    # the A/B switch loads it as it is, and the library reports ITS hash -- which is what bench.py prices the line with
    r = subprocess.run([sys.executable, "-c", prog], env=dict(env, RT_ALLOW_VARIANT_LIB="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and ("HASH " + fake) in r.stdout, r.stderr[-2000:]


def test_bench_prices_the_line_with_the_loaded_library(monkeypatch, tmp_path):
    import bench
    rt = importlib.import_module("cuda-raytracing_amd")
    h = rt.library_hash()
    assert h == build.library_code_hash() == build.kernel_code_hash()
    table = tmp_path / "counters.json"
    json.dump({"k": {"tag": "t", "code_hash": h, "valu_insts_per_frame": 1.0e8, "lanes_active_per_valu": 45.0,
                     "tcp_accesses_per_frame": 5.4e7, "hbm_bytes_per_frame": 9.0e6}}, open(table, "w"))
    monkeypatch.setattr(bench, "COUNTERS_JSON", str(table))
    line = bench.roofline("render_kernel", "k", 3.7, 32, 1.0)
    assert line["code_hash"] == h and line["sources_code_hash"] == h and line["profile_stale"] is False
    monkeypatch.setattr(rt, "library_hash", lambda hip=None: "0123456789abcdef")        # a variant library is running
    line = bench.roofline("render_kernel", "k", 3.7, 32, 1.0)
    assert line["code_hash"] == "0123456789abcdef" and line["profile_stale"] is True and line["frac"] is None
