"""The exchange step on a real GPU: the RCCL communicator of the C-ABI (rt_comm_* / rt_gather / rt_all_to_all /
rt_render_tiled) with the one rank a one-GPU box offers, and the double-buffered frame loop (tiling.StripePipeline) on its
real streams and events.  More than one RCCL peer needs more than one GPU: the N-rank bookkeeping is covered by the
virtual-rank tests (test_gpu_parity.py, test_gpu_full_size.py), by the gloo tests on CPU (test_tiling_gloo.py) and, for
the single-process form of the C-ABI (rt_render_tiled_all), by N ranks on the one GPU over an in-process mock of RCCL."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import scene_defs as sd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def comm1(rt):
    v = C.c_int32(0)
    rt.check(rt.libs()[0].rt_comm_available(C.byref(v)), "rt_comm_available")
    assert v.value > 20000                                     # RCCL reports NCCL-style version numbers (2.x.y -> 2xxyy)
    c = rt.Comm(rt.Comm.unique_id(), 0, 1)
    yield c
    c.close()


def test_comm_gather_and_all_to_all_single_rank(rt, comm1):
    h = rt.libs()[0]
    r, n, d = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
    rt.check(h.rt_comm_info(comm1.h, C.byref(r), C.byref(n), C.byref(d)))
    assert (r.value, n.value) == (0, 1) and d.value >= 0
    data = np.random.default_rng(5).integers(0, 256, 1 << 20, dtype=np.uint8)
    src, dst = rt.DeviceBuffer(nbytes=data.nbytes), rt.DeviceBuffer(nbytes=data.nbytes)
    rt.check(h.rt_memcpy_h2d(src.ptr, data.ctypes.data, data.nbytes, None))
    comm1.gather(src.ptr, data.nbytes, dst.ptr, root=0)
    rt.check(h.rt_device_synchronize())
    assert np.array_equal(dst.to_host(), data)
    dst2 = rt.DeviceBuffer(nbytes=data.nbytes)
    comm1.all_to_all(src.ptr, [4096], [512], dst2.ptr, [4096], [1024])
    rt.check(h.rt_device_synchronize())
    assert np.array_equal(dst2.to_host()[1024:1024 + 4096], data[512:512 + 4096])
    assert h.rt_gather(comm1.h, src.ptr, 16, None, 0, None) == -1          # the root needs a destination
    assert h.rt_gather(comm1.h, src.ptr, 16, dst.ptr, 1, None) == -1       # no such rank
    assert h.rt_gather(None, src.ptr, 16, dst.ptr, 0, None) == -1


@pytest.mark.parametrize("opts", [(1, 0, 0), (4, 2, 1)])
def test_render_scene_tiled_single_rank(rt, scenes, blob5k, comm1, opts):
    """Camera::render_scene_tiled -> rt_render_tiled with one rank renders the frame Camera::render_scene renders."""
    W, H = 322, 203
    sp = sd.shiny_scene(scenes, blob5k).build_product(rt)
    sp.upload_to_device()
    cam = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF)
    cam.set_pose(sd.SHINY_CAMERA["pose"])
    cam.set_options(*opts)
    want = rt.render(sp, cam)
    img = rt.DeviceBuffer(width_bytes=W * 3, height=H)
    cam.render_scene_tiled(sp, comm1, img.ptr, img.pitch, synchronize=True)
    assert np.array_equal(img.to_host().reshape(H, W, 3), want)
    # the single-process form (rt_render_tiled_all) with the same one communicator
    h = rt.libs()[0]
    p = cam.params()
    o = (C.c_int32 * 3)(*opts)
    scn, cm = (C.c_void_p * 1)(sp.device_handle), (C.c_void_p * 1)(comm1.h)
    img2 = rt.DeviceBuffer(width_bytes=W * 3, height=H)
    rt.check(h.rt_render_tiled_all(scn, cm, 1, C.byref(p), o, img2.ptr, img2.pitch, 16, 0, None, 1), "rt_render_tiled_all")
    assert np.array_equal(img2.to_host().reshape(H, W, 3), want)


def test_stripe_pipeline_on_streams_with_distinct_poses(rt, scenes, blob5k, comm1):
    """bench.py's N > 1 frame loop on its real streams (two compute streams, one comm stream, events between them) with a
    different camera for every frame of every group: if a render overwrote a stripe buffer that an exchange still reads,
    or an exchange a buffer the un-stripe pass still reads, some frame would carry another frame's rows."""
    import torch
    tiling = importlib.import_module("cuda-raytracing_amd.tiling")
    h = rt.libs()[0]
    W, H, F, ngroups, stripe = 480, 272, 4, 7, 16
    pitch = W * 3
    sp = sd.blob_scene(scenes, blob5k).build_product(rt)
    sp.upload_to_device()
    dev = torch.device("cuda", 0)
    cs = [torch.cuda.Stream(), torch.cuda.Stream()]
    cams = []
    for s in cs:
        c = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF)
        c.set_stream(s.cuda_stream)
        cams.append(c)
    poses = [[(0.03 * f - 0.01 * g, -1.5 - 0.1 * g - 0.02 * f, 0.2, 0.01 * g, -0.005 * f, 0.0) for f in range(F)] for g in range(ngroups)]
    max_rows = tiling.stripe_rows(H, stripe, 0, 1)
    local = [torch.zeros((F * max_rows, pitch), dtype=torch.uint8, device=dev) for _ in range(2)]
    gathered = [torch.zeros((1, F * max_rows, pitch), dtype=torch.uint8, device=dev) for _ in range(2)]
    frames = [torch.zeros((F, H, pitch), dtype=torch.uint8, device=dev) for _ in range(2)]
    results = torch.zeros((ngroups, F, H, pitch), dtype=torch.uint8, device=dev)
    ex = tiling.RcclExchange(comm1)
    group_of = [0, 0]

    def render_fn(b):
        cams[b].render_scene_stripes_batch(sp, poses[group_of[b]], tiling.batch_local_ptrs(local[b].data_ptr(), F, max_rows, pitch), pitch, stripe, 0, 1)

    def exchange_fn(b):
        ex.to_root(local[b], gathered[b], 0)

    def unstripe_fn(b):
        rt.check(h.rt_unstripe_batch(gathered[b].data_ptr(), pitch, F * max_rows * pitch, max_rows * pitch, frames[b].data_ptr(), pitch, H * pitch,
                                     F, W, H, stripe, 1, torch.cuda.current_stream().cuda_stream))
        results[group_of[b]].copy_(frames[b], non_blocking=True)            # (on the comm stream, behind the un-stripe pass)

    pipe = tiling.StripePipeline(render_fn, exchange_fn, unstripe_fn, compute_streams=cs, comm_stream=torch.cuda.Stream())
    for g in range(ngroups):
        group_of[g & 1] = g
        pipe.step(g)
    pipe.drain()
    torch.cuda.synchronize()
    got = results.cpu().numpy().reshape(ngroups, F, H, W, 3)
    cam = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF)
    for g in range(ngroups):
        for f in range(F):
            cam.set_pose(poses[g][f])
            assert np.array_equal(got[g, f], rt.render(sp, cam)), "group %d frame %d" % (g, f)
    assert not np.array_equal(got[0, 0], got[1, 0])


@pytest.mark.parametrize("extra", [["--workload", "c2", "--width", "640", "--height", "360", "--steps", "40", "--warmup", "8", "--rccl-max-channels", "4"],
                                   ["--workload", "c2", "--width", "640", "--height", "360", "--steps", "40", "--warmup", "8", "--gather", "root0"],
                                   ["--workload", "c3", "--width", "320", "--height", "180", "--spp", "4", "--bounces", "2", "--steps", "3", "--warmup", "1"],
                                   ["--workload", "c2", "--width", "640", "--height", "360", "--steps", "40", "--warmup", "8", "--exchange", "torch"],
                                   ["--workload", "c3", "--width", "320", "--height", "180", "--spp", "4", "--bounces", "2", "--steps", "3", "--warmup", "1",
                                    "--exchange", "torch"]])
def test_bench_forced_collective_path(extra):
    """bench.py --force-collective: the whole N > 1 code path (stripes, RCCL exchange through the C-ABI, un-stripe, frame
    check) with the one rank a one-GPU box has; --exchange torch = the announced fallback transport."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-collective", "--no-cpu-baseline", "--no-latency"] + extra,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["FORCED_COLLECTIVE_PATH"]
    assert line.get("frame_matches_debug_kernel", line.get("frame_matches_single_gpu_render")) is True
    assert ("EXCHANGE_FALLBACK" in line) == ("torch" in extra)
    if "c2" in extra:                                                   # the stream workloads run the pipeline: one rank's stage times
        assert len(line["per_rank"]) == 1 and line["per_rank"][0]["render_ms_per_group"] > 0 and line["stripe_share_imbalance"] == 1.0
        assert line["rccl"]["NCCL_MAX_NCHANNELS"] == ("4" if "--rccl-max-channels" in extra else None)


@pytest.mark.parametrize("workload", [["--workload", "c2", "--width", "480", "--height", "272", "--steps", "24", "--warmup", "4"],
                                      ["--workload", "c5", "--width", "480", "--height", "272", "--spp", "2", "--bounces", "1", "--steps", "2", "--warmup", "1"]])
def test_bench_self_launches_two_ranks(workload):
    """`python bench.py --gpus 2` started bare: it launches its two ranks itself (before touching a GPU) and relays ONE
    line.  On a one-GPU box the two ranks share the device, so the exchange goes through the gloo rehearsal backend."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--debug-backend", "gloo", "--no-cpu-baseline"] + workload,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and "REHEARSAL_NOT_A_MEASUREMENT" in line
    assert line.get("frame_matches_debug_kernel", line.get("frame_matches_single_gpu_render")) is True


@pytest.fixture(scope="module")
def mock_rccl(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("mock") / "librccl_mock.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-o", so, os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp"), "-lrt"],
                   check=True, timeout=600)
    return so


@pytest.mark.parametrize("extra", [["--gpus", "2", "--workload", "c2", "--width", "480", "--height", "272", "--steps", "40", "--warmup", "8"],
                                   ["--gpus", "4", "--workload", "c2", "--width", "322", "--height", "203", "--steps", "21", "--warmup", "5", "--frames-per-launch", "7"],
                                   ["--gpus", "3", "--workload", "c2", "--width", "480", "--height", "272", "--steps", "12", "--warmup", "3", "--gather", "root0"],
                                   # the driver's shape: ONE group per timed region, rendered as a pipeline of sub-groups (round 6)
                                   ["--gpus", "2", "--workload", "c2", "--width", "480", "--height", "272", "--steps", "20", "--warmup", "5"],
                                   ["--gpus", "4", "--workload", "c2", "--width", "322", "--height", "203", "--steps", "20", "--warmup", "5"],
                                   ["--gpus", "3", "--workload", "c2", "--width", "480", "--height", "272", "--steps", "7", "--warmup", "3"],
                                   ["--gpus", "2", "--workload", "c2", "--width", "480", "--height", "272", "--steps", "20", "--warmup", "5", "--subgroups", "1"],
                                   ["--gpus", "2", "--workload", "c5", "--width", "480", "--height", "272", "--spp", "6", "--bounces", "1", "--steps", "2", "--warmup", "1"],
                                   ["--gpus", "4", "--workload", "c3", "--width", "322", "--height", "203", "--spp", "3", "--bounces", "2", "--steps", "3", "--warmup", "1"],
                                   ["--gpus", "3", "--workload", "c3", "--width", "322", "--height", "203", "--spp", "70", "--bounces", "1", "--steps", "2", "--warmup", "1"]])
def test_bench_ranks_as_processes_over_mock_transport(mock_rccl, extra):
    """bench.py --gpus N with one process per rank, the ranks sharing the box's GPU: torch.distributed talks gloo, and the data
    path is the product's own N-rank code -- RtComm from a broadcast id, rt_all_to_all with the rotating plan or rt_gather,
    rt_unstripe_batch, rt_render_tiled, the double-buffered pipeline on its three streams -- over the tests' shared-memory
    stand-in for RCCL (real RCCL refuses two ranks on one device).  Every rank compares frames it assembled with the
    instrumented kernel's; the line must say it is a rehearsal."""
    env = dict(os.environ, RT_RCCL_LIBRARY=mock_rccl)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--debug-backend", "gloo", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    import json
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == int(extra[1]) and "RT_RCCL_LIBRARY" in line["REHEARSAL_NOT_A_MEASUREMENT"] and "EXCHANGE_FALLBACK" not in line
    assert line["ranks_seen"] == int(extra[1])                       # the size the (mock) communicator itself reports
    assert line.get("frame_matches_debug_kernel", line.get("frame_matches_single_gpu_render")) is True
    assert "rt_" in line["config"]["parallelism"]
    if "c2" in extra:
        _check_per_rank_report(line, int(extra[1]))
        steps, world = int(extra[extra.index("--steps") + 1]), int(extra[1])
        if steps <= 32 and "--gather" not in extra and "--frames-per-launch" not in extra:
            # one group per timed region: sub-groups unless switched off; every frame of the group in exactly one of them, none smaller
            # than the rank count; the ranks' report says how much of a sub-group's exchange ran inside the next one's render
            sub = line["config"]["sub_groups"]
            if "--subgroups" in extra:
                assert sub is None
            else:
                assert sub is not None and len(sub) >= 2 and sum(c for _, c in sub) == steps and all(c >= world for _, c in sub), sub
                assert [f for f, _ in sub] == [sum(c for _, c in sub[:k]) for k in range(len(sub))]
                assert all("exchange_inside_next_render_frac" in r for r in line["per_rank"]), line["per_rank"]
                assert line["config"]["frames_per_launch"] == sub[0][1] and line["config"]["frames_per_group"] == steps


def _check_per_rank_report(line, world):
    """An N > 1 stream line says where every rank's time went: per group of frames the wait for a free buffer
    set, the render, the exchange and the un-stripe pass, measured with hipEvents at the stage boundaries."""
    ranks = line["per_rank"]
    assert [r["rank"] for r in ranks] == list(range(world))
    for r in ranks:
        assert r["groups_timed"] >= 1
        assert r["render_ms_per_group"] > 0 and r["exchange_ms_per_group"] > 0 and r["unstripe_ms_per_group"] >= 0 and r["wait_for_buffer_ms"] >= 0
        parts = r["wait_for_buffer_ms"] + r["render_ms_per_group"] + r["exchange_ms_per_group"] + r["unstripe_ms_per_group"]
        assert parts <= r["group_span_ms"] * 1.001 + 1e-3, r          # the four stages of a group follow one another: they add up to its span
    assert any(r["unstripe_ms_per_group"] > 0 for r in ranks)          # somebody assembles frames
    assert line["stripe_share_imbalance"] >= 1.0
    assert set(line["rccl"]) >= {"version", "through", "NCCL_MAX_NCHANNELS"} and line["rccl"]["through"].startswith("rt_comm")
    # the ranks of one node meet at a shared-memory barrier around the timed region (bench.NodeBarrier), not at an all-reduce
    assert line["barrier"].startswith("shared-memory rendezvous of the node's %d ranks" % world), line["barrier"]
    assert "stripe owner rotating" in line["config"]["parallelism"]


def test_bench_rank_that_never_joins_ends_the_run_inside_the_deadline(mock_rccl):
    """A peer that never reaches the communicator: rank 0 is then stuck inside rt_comm_init_rank (the mock, like RCCL, returns
    when every rank has joined).  bench.py's phase watchdog must end the run -- non-zero exit, no result line, the phase and
    rt_comm_last_error() on stderr -- within its deadline instead of leaving it to whoever started the run to time out."""
    import time
    env = dict(os.environ, RT_RCCL_LIBRARY=mock_rccl)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--debug-backend", "gloo", "--no-cpu-baseline", "--gpus", "2", "--workload", "c2",
                        "--width", "320", "--height", "192", "--steps", "8", "--warmup", "2", "--phase-deadline", "12", "--test-stall", "1:communicator"],
                       capture_output=True, text=True, timeout=600, env=env)
    took = time.time() - t0
    assert r.returncode != 0 and r.stdout.strip() == "", (r.returncode, r.stdout)
    assert "phase 'communicator' exceeded its deadline of 12 s" in r.stderr and "rt_comm_last_error()" in r.stderr, r.stderr[-3000:]
    assert "phases completed: process group, scene" in r.stderr
    assert took < 150, took                                          # start-up + the 12 s deadline + torchrun's tear-down; not the mock's own 120 s


def test_cpp_tiled_application(rt, orc, scenes, blob5k, tmp_path):
    """examples/tiled_main.cpp: ONE C++ process, a scene replica and an RCCL communicator per visible GPU
    (rt_comm_init_all), the frame through rt_render_tiled_all -- no Python, no torch in the process.  Its PNG must equal
    the oracle's frame, for a plain frame and for a 4-spp frame with bounces and shadow rays."""
    root = ROOT
    exe = str(tmp_path / "tiled")
    pkg = os.path.join(root, "cuda-raytracing_amd")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    "-I" + os.path.join(root, "include"), "-I" + os.path.join(pkg, "csrc", "host"),
                    os.path.join(root, "examples", "tiled_main.cpp"), "-L" + pkg, "-lrt_host", "-lrt_hip", "-Wl,-rpath," + pkg,
                    "-o", exe], check=True)
    W, H = 640, 360
    desc = sd.SceneDesc([((0.9, 0.5, 0.2), None, dict(roughness=0.05, metallic=0.4))], [("obj", blob5k)], [(0, 0, (0,) * 6, (1, 1, 1))])
    so = desc.build_oracle(orc)
    for opts in ((1, 0, 0), (4, 2, 1)):
        png = str(tmp_path / ("tiled_%d.png" % opts[0]))
        r = subprocess.run([exe, blob5k, png, str(W), str(H)] + [str(v) for v in opts], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "GPU(s)" in r.stdout and "wrote" in r.stdout
        ref = so.render_ex(W, H, scenes.scaled_K(W), scenes.D_REF, (0.0, -1.6, 0.2, 0, 0, 0), *opts, threads=16)["img"]
        assert np.array_equal(rt.read_image(png), ref), opts
    so.close()


def test_tiled_all_with_several_ranks_over_mock_rccl(blob5k, tmp_path):
    """rt_comm_init_all + rt_render_tiled_all with 2, 3 and 8 ranks.  Real RCCL refuses two ranks on one device, so the ranks
    of this test share the box's one GPU and the collective is tests/mock_rccl (device-to-device copies when the group
    closes), loaded through RT_RCCL_LIBRARY in a child process.  What runs for real is the product's N-rank code: stripes of
    every rank, scratch sizing, offsets of the gathered blocks, roots other than 0, ragged last stripes, the un-stripe."""
    mock = str(tmp_path / "librccl_mock.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-o", mock, os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp"), "-lrt"],
                   check=True, timeout=600)
    env = dict(os.environ, RT_RCCL_LIBRARY=mock)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mock_rccl", "tiled_all_ranks.py"), blob5k], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.strip().splitlines()[-1] == "OK 21"
