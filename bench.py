#!/usr/bin/env python3
"""bench.py -- headline benchmark of the raycast hot path on MI355X.

One "step" = one frame of the workload BASELINE.json's metric is quoted on (configs[1]): the
69 936-triangle synthetic "bunny-class" OBJ, 1920x1080, 1 primary ray per pixel (the reference
casts exactly one ray per pixel: raycast.cu:204), scene resident in HBM before timing starts.

  python bench.py [--gpus N --steps K --warmup W] [--camera far|mid|near] [--no-cpu-baseline]

Frames are issued in groups of F = min(32, K) (K // F full groups plus one shorter last group) through Camera::render_scene_batch /
rt_render_batch: one launch renders F complete frames into F buffers, so the last long rays of one frame
overlap the bulk of the next (the reference's own loop issues two renders per synchronise,
kernel.cu:277-279).  K steps = K frames = K/F launches; F = 1 gives one launch per frame.

N > 1 (launched by torch.distributed.run, one rank per GPU): the scene is replicated, every frame is
cut into 16-row stripes dealt round-robin to the ranks (rt_render_stripes_batch), and every group of F
frames ends with ONE RCCL collective plus rt_unstripe_batch: by default a gather whose root rotates over the
frames of the group, fused into one all-to-all (each rank assembles F/N frames; --gather root0 gathers every
frame to rank 0).  Consecutive groups alternate between two compute streams, the collective runs on RCCL's
stream and the un-stripe pass on a fourth, so the exchange of group i overlaps the render of group i+1.
Total work is fixed, so "scaling" is "strong".  Every rank checks the frames it assembled; rank 0 prints ONE JSON line
(stdout carries nothing else: library banners are redirected to stderr).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np
import torch                      # imported BEFORE librt_hip.so so both share one HIP runtime
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
rt = importlib.import_module("cuda-raytracing_amd")
scenes = importlib.import_module("cuda-raytracing_amd.scenes")
tiling = importlib.import_module("cuda-raytracing_amd.tiling")

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STRIPE_ROWS = 16


def algorithmic_bytes(st):
    """bytes = 24*N_aabb + 8*N_interior_pops + 8*N_leaf_pops + 52*N_tri_tests
    + 24*N_inside_hits + 3*rays, N_interior_pops = N_aabb / 2."""
    interior = st["aabb"] // 2
    leaf = st["pops"] - interior
    return 24 * st["aabb"] + 8 * interior + 8 * leaf + 52 * st["tris"] + 24 * st["inside"] + 3 * st["rays"]


def scene_path(workload="c2"):
    d = os.path.join(ROOT, ".scene_cache")
    os.makedirs(d, exist_ok=True)
    if workload == "c4":
        p = os.path.join(d, "atrium.obj")
        if not os.path.exists(p):
            scenes.write_atrium_obj(p)
        return p
    p = os.path.join(d, "blob70k.obj")
    if not os.path.exists(p):
        scenes.write_blob_obj(p, *scenes.blob_dims_for(scenes.C2["n_tris"]))
    return p


def cpu_baseline(obj, W, H, K, D, pose, gpu_stats, albedo):
    """The oracle (oracle/rt_oracle.c, kind "port") timed on this host: one full frame of the same
    workload, single-threaded and on all cores (row bands).  Also cross-checks the GPU's counters."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    orc.build_oracle()
    o = orc.oracle()
    m = o.obj_load(obj)
    s = orc.OracleScene(o)
    s.add_material(albedo)
    s.add_mesh(m)
    s.add_instance(0, 0)
    rows1 = max(8, H // 8)                                 # 1/8 of the frame, centred, single thread
    y0 = (H - rows1) // 2
    t = time.perf_counter()
    s.render(W, H, K, D, pose, y0=y0, y1=y0 + rows1, planes=False, threads=1)
    t1 = time.perf_counter() - t
    cores = min(os.cpu_count() or 1, 32)
    reps, tn = 0, 0.0
    full = None
    while tn < 10.0 and reps < 20:                         # ~10 s of all-core work
        t = time.perf_counter()
        full = s.render(W, H, K, D, pose, planes=False, threads=cores)
        tn += time.perf_counter() - t
        reps += 1
    st = full["stats"]
    counters_match = all(int(st[k]) == int(gpu_stats[k]) for k in ("pops", "aabb", "tris", "inside"))
    s.close()
    return {"value": round(W * H * reps / tn / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d full %dx%d frames of the same scene/camera, row bands over %d threads" % (reps, W, H, cores),
            "value_1core": round(W * rows1 / t1 / 1e6, 3), "sample_1core": "%d centre rows, 1 thread" % rows1,
            "gpu_counters_match_oracle": bool(counters_match)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=960)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="c2", choices=["c2", "c4"],
                    help="c2 = the metric's workload (70k-triangle blob, 1920x1080); c4 = BASELINE configs[3] scene (260k-triangle atrium, 3840x2160)")
    ap.add_argument("--camera", default="mid", choices=sorted(scenes.C2_CAMERAS))
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--frames-per-launch", type=int, default=0, help="0 = as many as one launch takes (32), reduced to a divisor of --steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--debug-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N > 1 logic with host-staged gathers (several ranks may share one GPU); never for numbers")
    ap.add_argument("--gather", default="rotate", choices=["rotate", "root0"],
                    help="N > 1: rotate = the gather's root rotates over the frames of a group, fused into one all-to-all (every rank "
                         "assembles 1/N of the frames); root0 = every frame is gathered to rank 0")
    ap.add_argument("--one-stream", action="store_true", help="N > 1: launch every group on the same stream (no overlap of consecutive launches)")
    ap.add_argument("--two-streams", action="store_true", help="one GPU: alternate consecutive groups between two streams as the N > 1 path does")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the N > 1 path (stripes, RCCL gather, un-stripe) even with one rank: a check of that path on a one-GPU box, not the N = 1 number")
    ap.add_argument("--latency-probe", action="store_true", help="also time 20 single-frame launches (adds launches of the same kernel)")
    args = ap.parse_args()

    # stdout carries exactly one line, the result: anything libraries print there (RCCL's version banner, for one) goes to
    # stderr instead -- file descriptor 1 is pointed at stderr and the JSON is written to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    dist_on = world > 1 or args.force_collective                # stripes + gather + un-stripe instead of whole-frame launches
    rehearsal = dist_on and args.debug_backend == "gloo"
    if rehearsal:
        local_rank %= max(torch.cuda.device_count(), 1)          # ranks may share a device in the rehearsal
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    rt.build()
    rt.libs()
    wl = scenes.C4 if args.workload == "c4" else scenes.C2
    W, H = args.width or wl["width"], args.height or wl["height"]
    K, D = scenes.scaled_K(W), scenes.D_REF
    pose = scenes.C4["cam_pose"] if args.workload == "c4" else scenes.C2_CAMERAS[args.camera]
    obj = scene_path(args.workload) if rank == 0 else None
    if dist_on:
        dist.barrier()
        obj = scene_path(args.workload)

    # ---- scene: the reference's call sequence (kernel.cu:166-243) through the host C++ API ----
    mesh = rt.Mesh.load_obj(obj)
    scene = rt.Scene()
    scene.add_material(wl["albedo"])
    scene.add_mesh(mesh)
    scene.add_mesh_instance(0, 0)
    scene.upload_to_device()
    cam = rt.Camera(W, H, K, D)
    cam.set_pose(pose)
    stream = torch.cuda.current_stream().cuda_stream
    cam.set_stream(stream)
    # N > 1: consecutive groups are launched on two alternating streams (one camera object bound to each): the last waves
    # of a launch -- a few long silhouette rays on an otherwise empty chip -- then overlap the first waves of the next one,
    # which matters when a rank's share of a group is short.  One GPU keeps everything on one stream by default (worth 1-4 %
    # there, --two-streams): rocprofv3's per-kernel durations then stay comparable with the hipEvent figure below, while
    # overlapping launches each look longer than they cost.
    two = (dist_on or args.two_streams) and not args.one_stream and not rehearsal   # (the rehearsal stages through the host on one stream)
    cstreams = [torch.cuda.Stream(), torch.cuda.Stream()] if two else None
    cams = [cam, cam]
    if two:
        cams = [rt.Camera(W, H, K, D), rt.Camera(W, H, K, D)]
        for c_, s_ in zip(cams, cstreams):
            c_.set_pose(pose)
            c_.set_stream(s_.cuda_stream)

    # A launch should carry several frames' worth of work for THIS GPU (with N GPUs a rank renders only 1/N of each
    # frame), so frames go in groups as large as one launch allows (RT_MAX_BATCH = 32): K frames = K // F full groups
    # plus one last group of K % F frames.
    f_max = args.frames_per_launch if args.frames_per_launch > 0 else 32
    F = max(1, min(f_max, 32, args.steps if args.steps > 0 else 1))
    groups = [F] * (args.steps // F) + ([args.steps % F] if args.steps % F else [])
    warmup_req = args.warmup
    warm_groups = [F] * ((args.warmup + F - 1) // F)             # whole groups: at least the requested warm-up
    args.warmup = len(warm_groups) * F
    pitch = W * 3
    hlib = rt.libs()[0]
    rotate = dist_on and args.gather == "rotate"
    # finished frames of one group that THIS rank holds: all F (one GPU, or root0 on rank 0), or its share of a rotating gather
    my_frames = (lambda c: tiling.rotating_plan(c, world)[3][rank]) if rotate else (lambda c: c if (rank == 0 or not dist_on) else 0)
    slots = tiling.rotating_plan(F, world)[0] if rotate else F   # frame slots per local buffer (>= world for the rotating exchange)
    frames = torch.empty((max(my_frames(F), 1), H, pitch), dtype=torch.uint8, device=dev)
    frames_b = [frames, torch.empty_like(frames)] if not dist_on else None   # one GPU: groups alternate between two frame sets
    if dist_on:
        rows = []
        for r in range(world):
            n = C.c_int32(0)
            rt.check(hlib.rt_stripe_rows(H, STRIPE_ROWS, r, world, C.byref(n)))
            rows.append(n.value)
        max_rows = max(rows)
        # per buffer: F frames x this rank's (padded) stripe rows; what comes back is source-rank-major
        local = [torch.zeros((slots * max_rows, pitch), dtype=torch.uint8, device=dev) for _ in range(2)]
        if rotate:
            gathered = [torch.empty((world * tiling.rotating_plan(F, world)[1][rank] * max_rows, pitch), dtype=torch.uint8, device=dev) for _ in range(2)]
        else:
            gathered = [torch.empty((world, F * max_rows, pitch), dtype=torch.uint8, device=dev) if rank == 0 else None for _ in range(2)]
        if rehearsal:                                            # gloo cannot move device tensors: stage through the host
            local_dev, gathered_dev = local, gathered
            local = [torch.zeros_like(t, device="cpu") for t in local_dev]
            gathered = [torch.empty_like(t, device="cpu") if t is not None else None for t in gathered_dev]

    timer = rt.Timer()

    frame_ptrs = [[fb[f].data_ptr() for f in range(F)] for fb in frames_b] if not dist_on else None
    dev_local = (local_dev if rehearsal else local) if dist_on else None
    dev_gathered = (gathered_dev if rehearsal else gathered) if dist_on else None
    local_ptrs = [tiling.batch_local_ptrs(dev_local[b].data_ptr(), F, max_rows, pitch) for b in range(2)] if dist_on else None

    counts = sorted(set(groups + [F]))                           # a full group and, possibly, the shorter last one
    render_single_calls = {(b, c): cams[b].prepared_batch(scene, [pose] * c, frame_ptrs[b][:c], pitch)
                           for b in range(2) for c in counts} if not dist_on else None
    render_local_calls = {(b, c): cams[b].prepared_batch(scene, [pose] * c, local_ptrs[b][:c], pitch, stripes=(STRIPE_ROWS, rank, world))
                          for b in range(2) for c in counts} if dist_on else None
    timing_call = (cam.prepared_batch(scene, [pose] * F, local_ptrs[0][:F], pitch, stripes=(STRIPE_ROWS, rank, world)) if dist_on else
                   cam.prepared_batch(scene, [pose] * F, frame_ptrs[0][:F], pitch))      # on `stream`, for the hipEvent timing below
    group_count = [F, F]                                         # frames in the group that currently occupies buffer b

    def render_local(b):
        render_local_calls[(b, group_count[b])]()
        if rehearsal:
            local[b].copy_(dev_local[b])

    def unstripe(b):
        count = my_frames(group_count[b])
        if count == 0:
            return
        if rehearsal:
            dev_gathered[b].copy_(gathered[b])
        # rank r's block holds its stripes of my `count` frames (rotate) or of all F frame slots (root0)
        rank_stride = (tiling.rotating_plan(group_count[b], world)[1][rank] if rotate else F) * max_rows * pitch
        rt.check(hlib.rt_unstripe_batch(dev_gathered[b].data_ptr(), pitch, rank_stride, max_rows * pitch, frames.data_ptr(), pitch, H * pitch,
                                        count, W, H, STRIPE_ROWS, world, torch.cuda.current_stream().cuda_stream))

    def exchange(b):
        if rotate:
            return tiling.exchange_rotating(local[b], gathered[b], group_count[b], world, max_rows)
        return tiling.exchange_to_root(local[b], gathered[b], rank)

    # un-stripe passes run on their own stream: a rank renders group i+1 while group i is exchanged and re-ordered
    side = torch.cuda.Stream() if dist_on and not rehearsal else None
    pipe = tiling.StripePipeline(render_local, exchange, unstripe, assembles=rotate or rank == 0, side_stream=side,
                                 compute_streams=cstreams) if dist_on else None

    def step_group(i, count):
        if not dist_on:
            render_single_calls[(i & 1, count)]()
        else:
            pipe.release(i & 1)                                  # buffer i & 1 is about to be reused: its count changes below
            group_count[i & 1] = count
            pipe.step(i)

    def sync():
        if dist_on:
            pipe.drain()
            dist.barrier()
        torch.cuda.synchronize()

    for i, c in enumerate(warm_groups):
        step_group(i, c)
    sync()
    t0 = time.perf_counter()
    for i, c in enumerate(groups):
        step_group(i, c)
    t_issue = time.perf_counter() - t0                           # host time to issue all groups (must stay below the GPU's)
    sync()
    dt = time.perf_counter() - t0
    if dist_on:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- kernel-only duration with hipEvents on the launch stream (roofline numerator) ----
    kernel_ms = None
    if rank == 0:
        n = max(10, min(len(groups), 100))
        group_count[0] = F
        torch.cuda.synchronize()
        timer.start(stream)
        for _ in range(n):
            timing_call()
        timer.stop(stream)
        kernel_ms = timer.elapsed_ms() / n                      # one launch = F frames (this rank's stripes of them)
        single_ms = None
        if args.latency_probe:                                  # latency of a single-frame launch, for reference
            one = rt.DeviceBuffer(width_bytes=W * 3, height=H)
            torch.cuda.synchronize()
            timer.start(stream)
            for _ in range(20):
                cam.render_scene(scene, one.ptr, one.pitch)
            timer.stop(stream)
            single_ms = timer.elapsed_ms() / 20
            one.free()
    if dist_on:
        dist.barrier()

    # ---- every rank checks the frames it assembled in the last group against the debug kernel's frame ----
    dbg = rt.render_debug(scene, cam)
    mine = my_frames(groups[-1] if groups else F)
    last_frames = frames if dist_on else frames_b[(len(groups) - 1) & 1]
    frames_host = last_frames.cpu().numpy().reshape(-1, H, W, 3)
    frame_ok = bool(all(np.array_equal(frames_host[f], dbg["img"]) for f in range(mine)))
    if dist_on:
        flag = torch.tensor([int(frame_ok)], dtype=torch.int32, device="cpu" if rehearsal else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        frame_ok = bool(flag.item())

    if rank == 0:
        # ---- per-frame work counters from the debug kernel (same traversal, extra stores) ----
        st = {"rays": W * H, "pops": int(dbg["pops"].sum()), "aabb": int(dbg["aabb"].sum()), "tris": int(dbg["tris"].sum()),
              "inside": int(dbg["inside"].sum()), "hits": int((dbg["hit_tri"] >= 0).sum())}
        alg_bytes = algorithmic_bytes(st)                   # per frame
        share = F / world                                   # one launch = F frames; rank 0's stripes ~ 1/N of each
        achieved = alg_bytes * share / (kernel_ms * 1e-3) / 1e9
        # committed rocprofv3 PMC summary of this workload (tools/profile_bench.sh -> profiles/r01_traffic.json), if there is one:
        # physical HBM bytes per launch, and what actually limits the kernel (VALU issue slots, lanes active per instruction)
        traffic, pmc = None, {}
        tp = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tp):
            try:
                entry = json.load(open(tp)).get("%s_%dx%d_f%d" % (args.camera if args.workload == "c2" else args.workload, W, H, F), {})
                traffic = entry.get("hbm_bytes_per_launch")
                c = {k: v["mean"] for k, v in entry.get("counters", {}).items()}
                if all(k in c for k in ("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU")):
                    # 1024 SIMDs, 2 cycles per wave64 VALU instruction; GRBM_GUI_ACTIVE is summed over the 8 XCDs
                    pmc = {"limiter": "valu_issue", "valu_issue_frac_profiled": round(c["SQ_INSTS_VALU"] * 2 / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 3),
                           "lanes_active_per_valu_profiled": round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"], 1),
                           "profile": entry.get("tag")}
            except Exception:
                traffic, pmc = None, {}
        out = {
            "metric": "Mrays/sec + ms/frame, 70k-tri OBJ at 1920x1080 1spp; 1/2/4/8 MI355X",
            "value": round(W * H * args.steps / dt / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, **({"REHEARSAL_NOT_A_MEASUREMENT": "gloo backend, host-staged gathers"} if rehearsal else {}),
            **({"FORCED_COLLECTIVE_PATH": "N > 1 code path run with one rank"} if args.force_collective and world == 1 else {}), "steps": args.steps, "warmup": warmup_req, "warmup_frames_done": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("C4 Sponza-class atrium OBJ (%d tris, %d BVH nodes), %dx%d, 1 primary ray/pixel, camera inside %s"
                                    % (mesh.num_triangles, mesh.num_nodes, W, H, str(tuple(pose[:3]))))
                                   if args.workload == "c4" else
                                   ("C2 bunny-class blob OBJ (69936 tris, 130227 BVH nodes), %dx%d, 1 primary ray/pixel, camera '%s' %s"
                                    % (W, H, args.camera, str(tuple(pose[:3])))),
                       "parallelism": "replicated scene, %d-row stripes round-robin over %d GPU(s)%s"
                                      % (STRIPE_ROWS, world, (", one RCCL all-to-all per %d frames (the gather's root rotates: each rank assembles 1/N of the frames)" % F if rotate
                                                           else ", one RCCL gather to rank 0 per %d frames" % F) if dist_on else ""),
                       "frames_per_launch": F, "host_issue_ms_per_launch": round(t_issue / max(len(groups), 1) * 1e3, 3), "single_frame_launch_ms": None if single_ms is None else round(single_ms, 4),
                       "coverage": round(st["hits"] / st["rays"], 4),
                       "per_ray": {k: round(st[k] / st["rays"], 3) for k in ("pops", "aabb", "tris", "inside")},
                       "algorithmic_bytes_per_ray": round(alg_bytes / st["rays"], 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "render_kernel<false,false>", "kernel_ms": round(kernel_ms, 4),
                         "frames_per_launch": F, "algorithmic_bytes_per_launch": int(alg_bytes * share), **pmc},
            "frame_matches_debug_kernel": frame_ok,
        }
        if not dist_on and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(obj, W, H, K, D, pose, st, wl["albedo"])
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
