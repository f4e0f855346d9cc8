#!/usr/bin/env python3
"""bench.py -- the raycast hot path on MI355X, one JSON line per run.

  python bench.py [--gpus N --steps K --warmup W] [--workload c2|c3|c4|c5] [--camera far|mid|near]

Workloads (BASELINE.json configs[1..4]; scenes are synthetic, see cuda-raytracing_amd/scenes.py):
  c2  70k-triangle blob, 1920x1080, 1 primary ray per pixel -- the workload the metric is quoted on; the default
  c3  the same mesh, 64 spp, 8 specular bounces, sun + shadow rays     (extension kernel)
  c4  260k-triangle atrium, 3840x2160, 16 spp                           (--spp 1: the primary kernel at 4K)
  c5  the c3 workload at 7680x4320, meant for --gpus 8 (frame tiled over the ranks, RCCL gather to rank 0)
  c6  the atrium generator at 4.07 M triangles (395 MB of records: larger than the 256 MiB Infinity Cache), 3840x2160, 1 primary
      ray per pixel -- the HBM regime; `--bounces 2 --metallic 1 --roughness 0.3` is its incoherent variant
One "step" = one frame.  `value` = primary rays (width x height x spp) per second of wall clock over K steps, scene and
frame buffers resident in HBM, timed between two barrier + synchronise pairs, max over ranks.  The untimed warm-up is at
least W steps and, for the 1-spp stream, at least 320 frames (a frame is 0.13 ms; the GPU's clock needs tens of
milliseconds of load to settle): the line reports the requested and the executed warm-up.

c2 (and any 1-spp run) is a frame STREAM: frames are issued in groups of F = min(32, K) through
Camera::render_scene_batch / rt_render_batch (one launch renders F frames along a short camera path: every frame of a
group has its own pose), because a single 1080p frame ends in a tail of a few long rays that leaves most of the chip idle.
The line also carries the latency figures the batch hides (config.latency): one frame per launch, two frames per launch
and the reference's own loop of two renders per synchronise (kernel.cu:277-279).
c3 / c4 / c5 render one frame per step (rt_render_ex: spp sample planes per launch, then a resolve pass).

N > 1: the scene is replicated, the frame is cut into 16-row stripes dealt round-robin to the ranks, and the exchange
step runs over RCCL through the C-ABI (include/rt_hip.h: rt_gather / rt_all_to_all / rt_render_tiled).  Single frames
(c3-c5) are gathered to rank 0 (Camera::render_scene_tiled).  The c2 stream exchanges once per group of F frames, by
default with a rotating root (each rank assembles F/N frames; one fused all-to-all, --gather root0 for the plain gather),
double-buffered so that the exchange of group i overlaps the render of group i + 1 (tiling.StripePipeline).
torch.distributed carries only the control plane (barriers, the max over ranks, the RCCL unique id).
`python bench.py --gpus N` started WITHOUT torchrun launches its N ranks itself (fresh child processes via
torch.distributed.run, before this process touches a GPU) and relays their one line.

stdout carries exactly one line, the result; everything else (library banners included) goes to stderr.
"""
import argparse
import ctypes as C
import importlib
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch                      # imported BEFORE librt_hip.so so both share one HIP runtime (and one RCCL)
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
rt = importlib.import_module("cuda-raytracing_amd")
scenes = importlib.import_module("cuda-raytracing_amd.scenes")
tiling = importlib.import_module("cuda-raytracing_amd.tiling")

METRIC = "Mrays/sec + ms/frame, 70k-tri OBJ at 1920x1080 1spp; 1/2/4/8 MI355X"
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CLOCK_HZ = 2.4e9                 # max shader clock
N_SIMD = 256 * 4
VALU_PEAK_GINST = N_SIMD * CLOCK_HZ / 2 / 1e9     # wave64 VALU instructions per second: 2 cycles each on a SIMD-32 (MICROARCH "Wave scheduling")
L1_CALIBRATION_JSON = os.path.join(ROOT, "profiles", "r05_l1_calibration.json")


def l1_peak_gaccess():
    """The vector L1's ceiling in TCP_TOTAL_CACHE_ACCESSES per second, MEASURED for the traversal's fetch (tools/l1_calibration.hip:
    four global_load_dwordx4 per lane from an L1-resident table, 8 waves per SIMD, every CU): 1 031 G accesses/s = 1.68 per clock and
    CU once the lanes of a wave-instruction hold 4 or more different 64-B records (one access per lane then; 38 clocks per
    wave-instruction), and the data path's 64 B per clock and CU (16 clocks per wave-instruction) when they all hold the same one.
    Rounds 1-4 priced accesses at an ASSUMED 64 B per clock and CU (= 614 G accesses/s): the L1 fraction printed then was 1.68 x
    too high."""
    try:
        return json.load(open(L1_CALIBRATION_JSON))["ceiling"]["tcp_accesses_per_second"] / 1e9, "measured (profiles/r05_l1_calibration.json)"
    except Exception:
        return 1031.4, "measured in round 5 (profiles/r05_l1_calibration.json not found: the constant)"


L1_DATA_CLK_PER_WAVE_LOAD = 16.3                  # clocks a CU's L1 spends on one full-width global_load_dwordx4 even when every lane hits one line (same calibration)
SALU_PEAK_GINST = 256 * CLOCK_HZ / 1e9            # one scalar unit per CU (MICROARCH glossary "CU"), one instruction per clock
STRIPE_ROWS = 16
MIN_WARM_FRAMES = 320            # 1-spp stream workloads: untimed frames before the timed region, whatever --warmup asks (see run_stream)
COUNTERS_JSON = os.path.join(ROOT, "profiles", "r06_counters.json")
_build = importlib.import_module("cuda-raytracing_amd._build")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class PhaseWatchdog:
    """Deadlines for the phases of a multi-rank run.  A rank stuck inside a collective (a peer that never joined, a wedged
    link) would otherwise only show as the driver's time limit.  A daemon thread watches the clock: when the phase the main
    thread is in outlives its deadline, the rank prints which phase, the communicator's last error text and the phases it
    had completed, and leaves with a non-zero code through os._exit -- the main thread may be blocked inside RCCL, and a
    process that has touched the GPU is never re-executed.  torchrun then ends the other ranks."""

    def __init__(self, rank, seconds, enabled):
        import threading
        self.rank, self.seconds, self.enabled = rank, seconds, enabled
        self.phase, self.deadline, self.done = None, None, []
        self.lock = threading.Lock()
        if enabled:
            threading.Thread(target=self._watch, daemon=True).start()

    def enter(self, name, scale=1.0):
        with self.lock:
            if self.phase is not None:
                self.done.append(self.phase)
            self.phase, self.deadline = name, time.monotonic() + self.seconds * scale
        if self.enabled:
            log("bench.py rank %d: phase '%s' (deadline %.0f s)" % (self.rank, name, self.seconds * scale))

    def finish(self):
        with self.lock:
            if self.phase is not None:
                self.done.append(self.phase)
            self.phase, self.deadline = None, None

    def _watch(self):
        while True:
            time.sleep(0.25)
            with self.lock:
                late = self.phase is not None and time.monotonic() > self.deadline
                phase, done = self.phase, list(self.done)
            if late:
                try:
                    err = rt.Comm.last_error_any()                      # (rt_comm_last_error() is per thread: this thread made no comm call)
                except Exception as e:                                  # (the library may not even be loaded yet)
                    err = "unavailable (%s)" % e
                log("bench.py rank %d: phase '%s' exceeded its deadline of %.0f s -- giving up (exit code 3).  rt_comm_last_error(): %r; "
                    "phases completed: %s; NCCL_DEBUG=%s" % (self.rank, phase, self.seconds, err, ", ".join(done) or "none", os.environ.get("NCCL_DEBUG", "")))
                sys.stderr.flush()
                os._exit(3)


REPEATS_SHORT = 9                # a timed region of ONE launch is measured this many times, back to back; the line reports the median


class NodeBarrier:
    """The barrier that brackets the timed region when every rank is a process of ONE node (what the driver launches): a
    rendezvous through a page of shared memory (a file under /dev/shm mapped by every rank) -- rank r publishes the number of the
    barrier it has reached in its own cache line (one writer per line: no atomics needed) and waits until every line has reached
    it -- a few microseconds, against the 50-100 us of an all-reduce through RCCL (dist.barrier()).  That difference is part of
    what the line times: for the driver's `--steps 20` an 8-rank region is about half a millisecond.  Ranks on several nodes, or
    any failure to set the page up on any rank, fall back to dist.barrier() on every rank; the line's `barrier` says which ran."""

    def __init__(self, rank, world, on_cpu):
        self.rank, self.world, self.epoch, self.slots, self.path = rank, world, 0, None, None
        self.kind = "torch.distributed barrier"
        if world < 2:
            return
        one_node = int(os.environ.get("LOCAL_WORLD_SIZE", "0") or 0) == world
        box = [None]
        if rank == 0 and one_node and os.path.isdir("/dev/shm"):
            try:
                path = "/dev/shm/rt_bench_barrier_%d_%s" % (os.getpid(), os.environ.get("MASTER_PORT", "0"))
                with open(path, "wb") as f:
                    f.write(bytes(64 * world))
                box[0] = path
            except OSError as e:
                log("bench.py: no shared-memory barrier (%s)" % e)
        dist.broadcast_object_list(box, src=0)                   # (collectives stay outside the try blocks: every rank makes them)
        slots = None
        if box[0] is not None:
            try:
                slots = np.memmap(box[0], dtype=np.int64, mode="r+", shape=(world, 8))
            except (OSError, ValueError) as e:
                log("bench.py rank %d: cannot map %s (%s)" % (rank, box[0], e))
        flag = torch.tensor([1 if slots is not None else 0], dtype=torch.int32, device="cpu" if on_cpu else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)              # every rank uses the same kind of barrier
        if rank == 0 and box[0] is not None:
            # every rank that could map the page has mapped it (the all-reduce above is behind their mmap): the NAME can go now, the
            # mappings keep the page alive -- nothing is left under /dev/shm if a rank dies later
            try:
                os.remove(box[0])
            except OSError:
                pass
        if int(flag.item()) == 1:
            self.slots = slots
            self.kind = "shared-memory rendezvous of the node's %d ranks (/dev/shm)" % world
        elif not one_node:
            self.kind += " (LOCAL_WORLD_SIZE = %s is not the world size %d: the ranks are not known to share a node)" % (os.environ.get("LOCAL_WORLD_SIZE", "unset"), world)

    def wait(self):
        if self.slots is None:
            dist.barrier()
            return
        self.epoch += 1
        self.slots[self.rank, 0] = self.epoch
        deadline = time.monotonic() + 900.0
        spins = 0
        while int(self.slots[:, 0].min()) < self.epoch:
            spins += 1
            if spins > 20000:                                    # (a few milliseconds of spinning: a peer is late, not racing -- stop burning its core)
                time.sleep(0.0002)
            if spins % 4096 == 0 and time.monotonic() > deadline:
                raise RuntimeError("bench.py rank %d: a rank never reached barrier %d" % (self.rank, self.epoch))

    def close(self):
        self.slots = None


def aggregate_repeats(dts, steps):
    """dts = wall-clock seconds of each repeat of the K-step timed region (max over ranks already taken per repeat).
    -> dict(dt = the region's time the line is priced with (the median), fields = what the line says about it)."""
    d = sorted(float(v) for v in dts)
    n = len(d)
    med = d[n // 2] if n % 2 else 0.5 * (d[n // 2 - 1] + d[n // 2])
    fields = {"repeats": n}
    if n > 1:
        fields.update({"ms_per_step_min": round(d[0] / steps * 1e3, 4), "ms_per_step_max": round(d[-1] / steps * 1e3, 4),
                       "timed_region": "median of %d back-to-back repeats of the %d-step region (each between barrier + synchronise pairs)" % (n, steps)})
    return {"dt": med, "fields": fields}


def multi_rank_report(per_rank, comm):
    """N > 1 stream runs: the stage times every rank measured (tiling.StripePipeline.stage_times: hipEvents at the stage
    boundaries of every timed group), how unevenly the stripes loaded the ranks, and what is known about the RCCL in use."""
    renders = [r.get("render_ms_per_group") for r in per_rank if r and r.get("render_ms_per_group") is not None]
    out = {"per_rank": per_rank,
           "stripe_share_imbalance": round(max(renders) / (sum(renders) / len(renders)), 4) if renders and sum(renders) > 0 else None}
    ver = None
    try:
        v = C.c_int32(0)
        if rt.libs()[0].rt_comm_available(C.byref(v)) == 0:
            ver = v.value
    except Exception:
        pass
    out["rccl"] = {"version": ver, "through": "rt_comm (C-ABI)" if comm is not None else "torch.distributed",
                   "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"), "NCCL_MIN_NCHANNELS": os.environ.get("NCCL_MIN_NCHANNELS"),
                   "note": "channel count = RCCL's own choice unless NCCL_MAX_NCHANNELS is set (--rccl-max-channels); RCCL exposes no query for it"}
    return out


def camera_path(base, n):
    """n poses on a small closed loop around `base` (a few millimetres of travel and a fraction of a degree of yaw):
    every frame of a group is a different frame, the work per frame stays that of the named camera."""
    out = []
    for k in range(n):
        a = 2.0 * math.pi * k / max(n, 1)
        p = list(base)
        p[0] += 0.004 * math.sin(a)
        p[2] += 0.004 * (math.cos(a) - 1.0)                     # (k = 0 is `base` itself: the pinned frame)
        p[3] += 0.002 * math.sin(a)
        out.append(tuple(p))
    return out


def algorithmic_bytes(st):
    """bytes = 24*N_aabb + 8*N_interior_pops + 8*N_leaf_pops + 52*N_tri_tests
    + 24*N_inside_hits + 3*rays, N_interior_pops = N_aabb / 2.  Every fetch counted as if it missed all caches."""
    interior = st["aabb"] // 2
    leaf = st["pops"] - interior
    return 24 * st["aabb"] + 8 * interior + 8 * leaf + 52 * st["tris"] + 24 * st["inside"] + 3 * st["rays"]


def scene_path(workload):
    d = os.path.join(ROOT, ".scene_cache")
    os.makedirs(d, exist_ok=True)
    if workload == "c4":
        p = os.path.join(d, "atrium.obj")
        if not os.path.exists(p):
            scenes.write_atrium_obj(p)
        return p
    if workload == "c6":                                                # (296 MB of text, written in about 15 s)
        p = os.path.join(d, "atrium_c6.obj")
        if not os.path.exists(p):
            log("bench.py: writing the c6 scene (%d triangles) to %s" % (scenes.C6["n_tris"], p))
            scenes.write_atrium_obj(p, **scenes.C6["atrium"])
        return p
    if workload == "demo":                                              # (area, board as the demo places it, board with the translation baked in)
        ps = tuple(os.path.join(d, n) for n in ("demo_area.obj", "demo_board.obj", "demo_board_baked.obj"))
        if not all(os.path.exists(q) for q in ps):
            scenes.write_demo_objs(ps[0], ps[1])
            scenes.write_demo_objs(ps[0], ps[2], offset=scenes.DEMO["board_pose"][:3])
        return ps
    p = os.path.join(d, "blob70k.obj")
    if not os.path.exists(p):
        scenes.write_blob_obj(p, *scenes.blob_dims_for(scenes.C2["n_tris"]))
    return p


def scene_parts(workload, wl, obj, baked=False):
    """The scene of a workload as lists both sides build from: materials [(albedo, texture or None, dict)], mesh OBJ paths,
    instances [(mesh, material, pose, scale)].  demo = the reference's own scene shape (kernel.cu:166-240): two OBJ meshes, two
    textured materials, the second instance translated; baked = its twin with the translation folded into the vertices."""
    ident = (0.0,) * 6
    if workload == "demo":
        area_tex, board_tex = scenes.demo_textures()
        return ([(wl["albedo"], area_tex, {}), (wl["albedo"], board_tex, {})], [obj[0], obj[2] if baked else obj[1]],
                [(0, 0, ident, (1, 1, 1)), (1, 1, ident if baked else wl["board_pose"], (1, 1, 1))])
    return ([(wl["albedo"], None, dict(roughness=wl.get("roughness", 0.0), metallic=wl.get("metallic", 0.0)))], [obj], [(0, 0, ident, (1, 1, 1))])


def oracle_scene(obj, wl, workload="c2", baked=False):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    orc.build_oracle()
    o = orc.oracle()
    s = orc.OracleScene(o)
    mats, objs, insts = scene_parts(workload, wl, obj, baked)
    for albedo, tex, extra in mats:
        s.add_material(albedo, tex, **extra)
    for path in objs:
        s.add_mesh(o.obj_load(path))
    for mesh, mat, pose, scale in insts:
        s.add_instance(mesh, mat, pose, scale)
    return s


def cpu_baseline_stream(obj, wl, W, H, K, D, pose, gpu_stats, workload="c2", baked=False):
    """The oracle (oracle/rt_oracle.c, kind "port") timed on this host on the same frame: all cores (row bands, about
    10 s of work) and one core (one full frame).  Also cross-checks the debug kernel's visit counters."""
    s = oracle_scene(obj, wl, workload, baked)
    cores = min(os.cpu_count() or 1, 32)
    reps, tn, full = 0, 0.0, None
    while tn < 10.0 and reps < 20:
        t = time.perf_counter()
        full = s.render(W, H, K, D, pose, planes=False, threads=cores)
        tn += time.perf_counter() - t
        reps += 1
    t = time.perf_counter()
    s.render(W, H, K, D, pose, planes=False, threads=1)
    t1 = time.perf_counter() - t
    st = full["stats"]
    ok = all(int(st[k]) == int(gpu_stats[k]) for k in ("pops", "aabb", "tris", "inside"))
    s.close()
    return {"value": round(W * H * reps / tn / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d full %dx%d frames of the same scene/camera, row bands over %d threads" % (reps, W, H, cores),
            "value_1core": round(W * H / t1 / 1e6, 3), "sample_1core": "1 full %dx%d frame, 1 thread" % (W, H),
            "gpu_counters_match_oracle": bool(ok)}, full["img"]


def cpu_baseline_frame(obj, wl, W, H, K, D, pose, opts, gpu_img):
    """Frame workloads: the oracle renders centred row bands of the same frame (all samples, bounces and shadow rays) on
    all cores until about 10 s of work are done; the same rows of the GPU frame must be equal."""
    s = oracle_scene(obj, wl)
    cores = min(os.cpu_count() or 1, 32)
    spp, bounces, lighting = opts
    done, tn, rows, rays_all, ok = 0, 0.0, cores, 0, True
    y = max(0, H // 2 - cores // 2)
    while tn < 10.0 and y + rows <= H and done < H:
        t = time.perf_counter()
        ref = s.render_ex(W, H, K, D, pose, spp, bounces, lighting, threads=cores, y0=y, y1=y + rows)
        tn += time.perf_counter() - t
        ok = ok and np.array_equal(ref["img"][y:y + rows], gpu_img[y:y + rows])
        rays_all += ref["stats"]["rays"]
        done += rows
        y += rows
        rows = min(rows * 2, H - y) if y < H else 0
        if rows <= 0:
            break
    s.close()
    return {"value": round(W * done * spp / tn / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d rows of the same %dx%d frame (%d spp, %d bounces, lighting %d: %.1f rays cast per primary ray), row bands over %d threads"
                      % (done, W, H, spp, bounces, lighting, rays_all / max(W * done * spp, 1), cores),
            "gpu_rows_match_oracle": bool(ok)}


def self_launch(args):
    """`python bench.py --gpus N` without torchrun: start the N ranks as fresh child processes (nothing in THIS process
    has touched a GPU) and relay their one result line."""
    if args.debug_backend == "nccl" and torch.cuda.device_count() < args.gpus:       # (device_count does not initialise HIP)
        sys.exit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, torch.cuda.device_count()))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: launching %d ranks: %s" % (args.gpus, " ".join(cmd)))
    # the ranks report RCCL's own warnings (stderr is inherited: they arrive as they are printed) and watch their phases
    # themselves (PhaseWatchdog); the limit here is the backstop for a launcher that never returns
    env = dict(os.environ)
    env.setdefault("NCCL_DEBUG", "WARN")
    limit = args.phase_deadline * 8 + 600
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)                             # (the group this launcher started, by its id)
        proc.communicate()
        sys.exit("bench.py: the %d-rank run did not end within %d s and was killed" % (args.gpus, limit))
    r = proc
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.startswith('{"metric"')]
    for ln in out.decode(errors="replace").splitlines():
        if not ln.startswith('{"metric"'):
            log(ln)
    if r.returncode != 0 or len(lines) != 1:
        sys.exit("bench.py: the %d-rank run failed (exit code %d, %d result lines): see the ranks' messages above (phase, rt_comm_last_error, "
                 "RCCL warnings)" % (args.gpus, r.returncode, len(lines)))
    print(lines[0], flush=True)
    sys.exit(0)


def counters_entry(key):
    """Committed rocprofv3 PMC summary of this workload, normalised per frame (tools/summarize_profile.py)."""
    try:
        return json.load(open(COUNTERS_JSON)).get(key)
    except Exception:
        return None


def roofline(kernel, key, kernel_ms, frames_per_launch, share, alg_bytes_per_frame=None, code_hash=None):
    """What bounds the kernel.  Instruction and cache-access counts per frame are properties of (code, scene, camera, size) --
    the kernel's control flow depends on nothing else -- so they come from the committed PMC pass of the same workload
    (COUNTERS_JSON); the time they are divided by is measured live, here.  `share` = the fraction of
    every frame this rank renders.  The counters entry carries the hash of the kernel source and build flags it was taken
    from (cuda-raytracing_amd/_build.py kernel_code_hash); the library carries the hash it was compiled from
    (rt_build_info): when the two differ the instruction count is not this binary's, so the line says `profile_stale` and
    prices nothing."""
    e = counters_entry(key)
    sec = kernel_ms * 1e-3
    # the hash of the library that RAN (compiled into it, rt_build_info), not of the sources lying next to it
    now = code_hash if code_hash is not None else rt.library_hash()
    out = {"kernel": kernel, "kernel_ms": round(kernel_ms, 4), "frames_per_launch": frames_per_launch, "profile_key": key, "code_hash": now,
           "code_hash_of": "the loaded librt_hip.so (rt_build_info)", "sources_code_hash": _build.kernel_code_hash()}
    if e is None:
        out.update({"bound": "valu_issue", "achieved": None, "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instr/s", "frac": None,
                    "traffic": None, "profile_stale": False,
                    "note": "no PMC profile committed for this workload key: only the live kernel time is known"})
    elif e.get("code_hash") != now:
        out.update({"bound": "valu_issue", "achieved": None, "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instr/s", "frac": None,
                    "traffic": None, "profile_stale": True, "profile": e.get("tag"), "profile_code_hash": e.get("code_hash"),
                    "note": "the committed PMC profile of this workload was taken from other kernel code or build flags: re-run "
                            "tools/profile_bench.sh + tools/summarize_profile.py; only the live kernel time is reported"})
    else:
        out["profile_stale"] = False
        n = frames_per_launch * share
        valu = e["valu_insts_per_frame"] * n / sec / 1e9
        l1_peak, l1_peak_from = l1_peak_gaccess()
        l1 = e["tcp_accesses_per_frame"] * n / sec / 1e9 if e.get("tcp_accesses_per_frame") else None     # G accesses/s
        hbm = e["hbm_bytes_per_frame"] * n / sec / 1e9 if e.get("hbm_bytes_per_frame") is not None else None
        salu = e["per_frame"]["SQ_INSTS_SALU"] * n / sec / 1e9 if e.get("per_frame", {}).get("SQ_INSTS_SALU") else None
        # the bound = whichever unit is busiest (fractions of: VALU issue slots, L1 accesses, the CU's scalar unit, HBM bytes)
        fr = {"valu_issue": valu / VALU_PEAK_GINST}
        if l1 is not None:
            fr["l1"] = l1 / l1_peak
        if salu is not None:
            fr["salu_issue"] = salu / SALU_PEAK_GINST
        if hbm is not None:
            fr["hbm"] = hbm / HBM_PEAK_GBS
        bound = max(fr, key=fr.get)
        ach, peak, unit = {"valu_issue": (valu, VALU_PEAK_GINST, "G wave-instr/s"), "l1": (l1, l1_peak, "G TCP accesses/s"),
                           "salu_issue": (salu, SALU_PEAK_GINST, "G instr/s"), "hbm": (hbm, HBM_PEAK_GBS, "GB/s")}[bound]
        out.update({"bound": bound, "achieved": round(ach, 1), "peak": round(peak, 1), "unit": unit, "frac": round(fr[bound], 4),
                    "traffic": int(e["hbm_bytes_per_frame"] * n) if e.get("hbm_bytes_per_frame") is not None else None,
                    "fractions": {k: round(v, 4) for k, v in fr.items()},
                    "lanes_active_per_valu": e.get("lanes_active_per_valu"),
                    "useful_lane_slot_frac": round(valu / VALU_PEAK_GINST * e["lanes_active_per_valu"] / 64.0, 4) if e.get("lanes_active_per_valu") else None,
                    "valu_issue_frac": round(valu / VALU_PEAK_GINST, 4),
                    "l1_access_frac": round(l1 / l1_peak, 4) if l1 is not None else None,
                    # the other limit of the same unit: every vector load instruction occupies the L1's data path for at least 16 clocks
                    # (all 64 lanes active); the truth lies between the two figures
                    "l1_data_path_upper_bound": round(e["per_frame"]["SQ_INSTS_VMEM_RD"] * n / sec * L1_DATA_CLK_PER_WAVE_LOAD / (256 * CLOCK_HZ), 4)
                                                if e.get("per_frame", {}).get("SQ_INSTS_VMEM_RD") else None,
                    "hbm_physical_frac": round(hbm / HBM_PEAK_GBS, 5) if hbm is not None else None,
                    "profile": e.get("tag"),
                    "peaks": "VALU: 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction; L1: %.0f G TCP accesses/s, %s; SALU: 256 CUs x 1 instruction/clk; HBM 8 TB/s"
                             % (l1_peak, l1_peak_from)})
        if hbm is not None:
            # FETCH_SIZE / WRITE_SIZE count what leaves the L2s (Infinity Cache hits included): an upper bound on HBM bytes.
            # No x2 on FETCH_SIZE here: tools/fetch_calibration.sh measures 0.99 bytes reported per byte for this kernel's
            # access pattern (a gather of 64-B records; the guide's x2 is for 16 B-per-lane streaming reads).
            out["hbm"] = {"physical_GBs": round(hbm, 1), "frac_of_8TBs": round(hbm / HBM_PEAK_GBS, 5), "frac_of_6.29TBs": round(hbm / 6290.0, 5),
                          "fetch_bytes_per_launch": int(e.get("fetch_bytes_per_frame", 0) * n), "write_bytes_per_launch": int(e.get("write_bytes_per_frame", 0) * n),
                          "physical_over_algorithmic": round(e["hbm_bytes_per_frame"] / alg_bytes_per_frame, 5) if alg_bytes_per_frame else None,
                          "counted": "L2 misses towards the fabric (FETCH_SIZE + WRITE_SIZE, Infinity Cache hits included): an upper bound on HBM traffic"}
    if alg_bytes_per_frame is not None:
        a = alg_bytes_per_frame * frames_per_launch * share / sec / 1e9
        out["hbm_algorithmic"] = {"achieved_GBs": round(a, 1), "frac_of_8TBs": round(a / HBM_PEAK_GBS, 4), "bytes_per_launch": int(alg_bytes_per_frame * frames_per_launch * share),
                                  "note": "SURVEY 8(d) bytes (every record fetch counted as an HBM miss) / time; exceeds 1 because the working set "
                                          "is L1/L2/MALL resident -- not a bound, kept for comparison with round 1"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="frames to time (default: 960 for the 1-spp stream, 20 for spp / bounce workloads)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed frames before (default 64 / 3)")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5", "c6", "demo"])
    ap.add_argument("--baked", action="store_true", help="demo: the twin scene with the board's translation folded into its vertices (an identity instance)")
    ap.add_argument("--camera", default="mid", choices=sorted(scenes.C2_CAMERAS))
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0, help="0 = the workload's own (c2 1, c3 64, c4 16, c5 64)")
    ap.add_argument("--bounces", type=int, default=-1)
    ap.add_argument("--lighting", type=int, default=-1)
    ap.add_argument("--metallic", type=float, default=-1.0, help="override the workload's Material::metallic (bounce weight)")
    ap.add_argument("--roughness", type=float, default=-1.0, help="override the workload's Material::roughness")
    ap.add_argument("--frames-per-launch", type=int, default=0, help="stream workloads: 0 = as many as one launch takes (32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the F = 1 / F = 2 latency figures")
    ap.add_argument("--phase-deadline", type=float, default=180.0,
                    help="N > 1: seconds every phase of a rank may take (process group, scene, communicator, first exchange, timed loop x 4, check) "
                         "before the rank reports the phase and rt_comm_last_error() and exits non-zero")
    ap.add_argument("--test-stall", default="", help="tests only: 'RANK:PHASE' makes that rank sleep forever on entering that phase")
    ap.add_argument("--debug-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N > 1 logic with host-staged exchanges (several ranks may share one GPU); never for numbers")
    ap.add_argument("--gather", default="auto", choices=["auto", "rotate", "root0"],
                    help="N > 1 stream workloads: rotate = the gather's root rotates over the frames of a group (one fused all-to-all), "
                         "root0 = every frame to rank 0; auto = rotate for streams, root0 for single frames")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "torch"],
                    help="N > 1 data path: rccl = RtComm of the C-ABI (rt_gather / rt_all_to_all / rt_render_tiled), torch = the same buffers "
                         "through torch.distributed's nccl backend (what a failed RtComm creation falls back to, announced in the line)")
    ap.add_argument("--rccl-max-channels", type=int, default=0,
                    help="N > 1: NCCL_MAX_NCHANNELS for the ranks (0 = RCCL's own choice): bounds the CUs RCCL's kernels take from render_kernel, "
                         "for an A/B of the contention between the exchange of group i and the render of group i + 1")
    ap.add_argument("--subgroups", type=int, default=0,
                    help="N > 1 stream workloads whose timed region is ONE group (--steps <= 32, the driver's --steps 20): render it as a pipeline of "
                         "up to this many sub-groups, the exchange of sub-group k beside the render of k + 1 (0 = auto: 4; 1 = one launch, as rounds 1-5)")
    ap.add_argument("--one-stream", action="store_true", help="N > 1: launch every group on the same compute stream")
    ap.add_argument("--two-streams", action="store_true", help="one GPU: alternate consecutive groups between two streams")
    ap.add_argument("--predict-scaling", default=None, const="2,4,8", nargs="?", metavar="N[,N...]",
                    help="one GPU: time the render side of every virtual rank of an N-GPU run (its stripes of every group of frames, on the "
                         "pipeline's two compute streams, no exchange) and print the predicted render-side speedup per N (default 2,4,8)")
    ap.add_argument("--no-owner-rotation", action="store_true",
                    help="N > 1 stream workloads: every rank renders its OWN stripes of every frame (rounds 1-4) instead of the stripe owner "
                         "rotating over the frames of a group (rt_render_stripes_batch_rotating: equal shares for all ranks)")
    ap.add_argument("--stripe-rows", type=int, default=0,
                    help="N > 1 and --predict-scaling: rows per stripe (stripe s belongs to rank s %% N); 0 = the default (STRIPE_ROWS)")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the N > 1 path (stripes, RCCL exchange, un-stripe) even with one rank: a check of that path on a one-GPU box")
    args = ap.parse_args()
    wl0 = scenes.WORKLOADS[args.workload]
    one_spp = ((args.spp or wl0.get("spp", 1)), (args.bounces if args.bounces >= 0 else wl0.get("bounces", 0)),
               (args.lighting if args.lighting >= 0 else wl0.get("lighting", 0))) == (1, 0, 0)
    if args.steps is None:
        args.steps = 960 if one_spp else 20
    if args.warmup is None:
        args.warmup = 64 if one_spp else 3
    if args.steps < 1:
        sys.exit("bench.py: --steps must be at least 1")

    global STRIPE_ROWS
    if args.stripe_rows > 0:
        STRIPE_ROWS = args.stripe_rows
    if args.rccl_max_channels > 0:                               # (before any communicator exists; inherited by the ranks self_launch starts)
        os.environ["NCCL_MAX_NCHANNELS"] = str(args.rccl_max_channels)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    # stdout carries exactly one line, the result: fd 1 is pointed at stderr and the JSON goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    wd = PhaseWatchdog(rank, args.phase_deadline, enabled=world > 1)

    def phase(name, scale=1.0):
        wd.enter(name, scale)
        if args.test_stall == "%d:%s" % (rank, name):
            log("bench.py rank %d: --test-stall: sleeping in phase '%s'" % (rank, name))
            time.sleep(1e9)
    dist_on = world > 1 or args.force_collective                # stripes + exchange + un-stripe instead of whole-frame launches
    rehearsal = dist_on and args.debug_backend == "gloo"
    if rehearsal:
        local_rank %= max(torch.cuda.device_count(), 1)          # ranks may share a device in the rehearsal
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        phase("process group")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    phase("scene")
    rt.build()
    hlib = rt.libs()[0]
    wl = dict(scenes.WORKLOADS[args.workload])
    if args.metallic >= 0:
        wl["metallic"] = args.metallic
    if args.roughness >= 0:
        wl["roughness"] = args.roughness
    W, H = args.width or wl["width"], args.height or wl["height"]
    spp = args.spp or wl.get("spp", 1)
    bounces = args.bounces if args.bounces >= 0 else wl.get("bounces", 0)
    lighting = args.lighting if args.lighting >= 0 else wl.get("lighting", 0)
    stream_mode = (spp, bounces, lighting) == (1, 0, 0)           # the primary kernel, frames in groups; else one frame per step
    if not stream_mode and spp >= 4 and args.stripe_rows <= 0:
        # From 4 samples per pixel on a wave of the extension kernel is 2 x 2 pixels x 16 samples or 1 pixel x 64 samples and a
        # workgroup tile 4 or 2 rows high: stripes can be that thin at no cost in coherence, and thin stripes deal every rank
        # the same mix of rows (c5 at 8 ranks, predicted on one GPU: 7.77 x with 16-row stripes, 7.87 x with 4, 7.94 x with 2)
        STRIPE_ROWS = 4
    K, D = scenes.scaled_K(W), scenes.D_REF
    atrium = args.workload in ("c4", "c6")
    demo = args.workload == "demo"
    base_pose = wl["cam_pose"] if (atrium or demo) else scenes.C2_CAMERAS[args.camera]
    cam_name = "inside" if atrium else (("baked" if args.baked else "posed") if demo else args.camera)
    key = "%s_%s_%dx%d_%d_%d_%d" % (args.workload, cam_name, W, H, spp, bounces, lighting)
    if args.metallic >= 0 or args.roughness >= 0:
        key += "_m%g_r%g" % (wl.get("metallic", 0.0), wl.get("roughness", 0.0))
    obj = scene_path(args.workload) if rank == 0 else None
    if dist_on:
        dist.barrier()
        obj = scene_path(args.workload)

    # ---- scene: the reference's call sequence (kernel.cu:166-243) through the host C++ API ----
    mats, objs, insts = scene_parts(args.workload, wl, obj, args.baked)
    meshes = [rt.Mesh.load_obj(q) for q in objs]
    scene = rt.Scene()
    for albedo, tex, extra in mats:
        scene.add_material(albedo, texture_bgr=tex, **extra)
    for m in meshes:
        scene.add_mesh(m)
    for mi, ma, ipose, iscale in insts:
        scene.add_mesh_instance(mi, ma, ipose, iscale)
    scene.upload_to_device()
    mesh = meshes[0]
    scene_tris, scene_nodes = sum(m.num_triangles for m in meshes), sum(m.num_nodes for m in meshes)
    stream = torch.cuda.current_stream().cuda_stream

    def make_camera(s=None):
        c = rt.Camera(W, H, K, D)
        c.set_pose(base_pose)
        c.set_options(spp, bounces, lighting)
        c.set_stream(stream if s is None else s.cuda_stream)
        return c
    cam = make_camera()
    pitch = W * 3
    timer = rt.Timer()

    # ---- the exchange backend (N > 1): RCCL through the C-ABI, or the gloo rehearsal ----
    comm = exchange = None
    exchange_note = None
    mock_transport = False
    ranks_seen = 1
    if dist_on:
        phase("communicator")
        # rehearsal with RT_RCCL_LIBRARY set (the tests' shared-memory mock of RCCL, tests/mock_rccl): the ranks share one GPU, torch
        # talks gloo, and the DATA path is the product's -- RtComm, rt_all_to_all / rt_gather / rt_render_tiled, the pipeline on
        # its streams -- with only the transport faked.  Rehearsal without it: host-staged exchange through gloo.
        mock_transport = rehearsal and args.exchange == "rccl" and bool(os.environ.get("RT_RCCL_LIBRARY"))
        if rehearsal and not mock_transport:
            exchange = tiling.TorchExchange(rank, world)
            ranks_seen = dist.get_world_size()
        else:
            # the product path: RtComm (RCCL through the C-ABI).  If it cannot be created on every rank the run goes on with
            # torch.distributed's own RCCL communicator moving the same buffers, and SAYS SO in the result line.
            err = ""
            try:
                box = [rt.Comm.unique_id() if rank == 0 else None]
            except rt.RtError as e:
                box, err = [None], str(e)
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            if box[0] is not None and args.exchange == "rccl":
                try:
                    comm = rt.Comm(box[0], rank, world)
                except rt.RtError as e:
                    err = str(e)
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device="cpu" if rehearsal else dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                exchange = tiling.RcclExchange(comm)
                ranks_seen = comm.info()[1]                      # the size RCCL itself reports for the communicator
            else:
                ranks_seen = dist.get_world_size()
                if comm is not None:
                    comm.close()
                    comm = None
                if rehearsal:
                    sys.exit("bench.py: RT_RCCL_LIBRARY is set but the communicator could not be created: " + err)
                exchange = tiling.TorchExchange(rank, world, on_device=True)
                exchange_note = "torch.distributed (nccl backend) instead of rt_comm: " + (err or ("--exchange torch" if args.exchange != "rccl" else "another rank failed to create its RtComm"))
                log("bench.py rank %d: %s" % (rank, exchange_note))
        rows = [tiling.stripe_rows(H, STRIPE_ROWS, r, world) for r in range(world)]
        max_rows = max(rows)

    node_barrier = NodeBarrier(rank, world, on_cpu=rehearsal) if dist_on else None

    def sync():
        if dist_on:
            torch.cuda.synchronize()
            node_barrier.wait()
        torch.cuda.synchronize()

    phase("first exchange")
    if args.predict_scaling is not None:
        if dist_on:
            sys.exit("bench.py --predict-scaling runs on one GPU (no --gpus, no --force-collective)")
        res = predict_scaling(args, locals())
    elif stream_mode:
        res = run_stream(args, locals())
    else:
        res = run_frames(args, locals())
    if rank == 0:
        os.write(result_fd, (json.dumps(res) + "\n").encode())
    if dist_on:
        phase("shutdown")
        dist.barrier()
        node_barrier.close()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()
    wd.finish()


def base_line(args, env, value, dt, warmup_done, config, roof, extra):
    world, rehearsal = env["world"], env["rehearsal"]
    out = {"metric": METRIC, "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "ranks_seen": env.get("ranks_seen", 1)}
    if rehearsal:
        out["REHEARSAL_NOT_A_MEASUREMENT"] = ("gloo control plane; data path = the C-ABI exchange over the library RT_RCCL_LIBRARY names (the tests' shared-memory mock)"
                                              if env.get("mock_transport") else "gloo backend, host-staged exchanges")
    if args.force_collective and world == 1:
        out["FORCED_COLLECTIVE_PATH"] = "N > 1 code path run with one rank"
    if env.get("exchange_note"):
        out["EXCHANGE_FALLBACK"] = env["exchange_note"]
    if env.get("node_barrier") is not None:
        out["barrier"] = env["node_barrier"].kind               # what brackets the timed region
    out.update({"steps": args.steps, "warmup": args.warmup, "warmup_frames_done": warmup_done,
                "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic", "config": config, "roofline": roof})
    out.update(extra)
    return out


def run_stream(args, env):
    """1 primary ray per pixel (render_kernel): K frames in groups of F."""
    g = env
    rank, world, dist_on, rehearsal, dev = g["rank"], g["world"], g["dist_on"], g["rehearsal"], g["dev"]
    W, H, K, D, pitch, scene, cam, stream, hlib, timer = g["W"], g["H"], g["K"], g["D"], g["pitch"], g["scene"], g["cam"], g["stream"], g["hlib"], g["timer"]
    base_pose, exchange, make_camera = g["base_pose"], g["exchange"], g["make_camera"]
    rotate = dist_on and args.gather in ("auto", "rotate")
    two = (dist_on or args.two_streams) and not args.one_stream and not (rehearsal and not (exchange is not None and exchange.on_device))
    cstreams = [torch.cuda.Stream(), torch.cuda.Stream()] if two else None
    cams = [make_camera(cstreams[0]), make_camera(cstreams[1])] if two else [cam, cam]

    f_max = args.frames_per_launch if args.frames_per_launch > 0 else 32
    F = max(1, min(f_max, 32, args.steps if args.steps > 0 else 1))
    groups = [F] * (args.steps // F) + ([args.steps % F] if args.steps % F else [])
    # units = the pipeline steps of the timed region, (first frame within its group, frames) each.  Normally a unit is a group.  When
    # the region is ONE group on N ranks (the driver's `--steps 20`: 2.5 frames' worth of work per rank at N = 8, then an exchange
    # with nothing to hide behind), the group is rendered as a pipeline of sub-groups: the all-to-all and un-stripe pass of
    # sub-group k run on the comm stream while sub-group k + 1 renders on the other compute stream (tiling.sub_groups).
    subgrouped = dist_on and rotate and len(groups) == 1 and args.subgroups != 1
    units = tiling.sub_groups(groups[0], world, args.subgroups or 4) if subgrouped else [(0, c) for c in groups]
    subgrouped = subgrouped and len(units) > 1
    # Warm-up: whole groups, at least the requested W steps -- and at least MIN_WARM_FRAMES frames (about 45 ms of GPU work):
    # a step here is 0.13 ms, and a GPU that has just been idle needs tens of milliseconds of load to reach its steady
    # clock (measured: the same 20 timed steps take 0.153 ms each after 20 warm-up frames, 0.143 after 80, 0.135 after 320
    # or 1280).  Both numbers are in the line ("warmup" = requested, "warmup_frames_done").
    warm_frames = args.warmup if rehearsal else max(args.warmup, MIN_WARM_FRAMES * world)    # (a rank of N renders 1/N of every frame)
    warm_units = (units if subgrouped else [(0, F)]) * ((warm_frames + F - 1) // F)    # (the launch shapes of the timed region)
    poses = camera_path(base_pose, F)                            # frame f of every group uses poses[f]
    shapes = sorted(set(units + [(0, F)] + ([] if subgrouped else [(0, c) for c in groups])))

    if not dist_on:
        frames_b = [torch.empty((F, H, pitch), dtype=torch.uint8, device=dev) for _ in range(2)]   # groups alternate between two frame sets
        calls = {(b, c): cams[b].prepared_batch(scene, poses[:c], [frames_b[b][f].data_ptr() for f in range(c)], pitch)
                 for b in range(2) for _, c in shapes}

        def step_group(i, unit):
            if two:
                with torch.cuda.stream(cstreams[i & 1]):
                    calls[(i & 1, unit[1])]()
            else:
                calls[(i & 1, unit[1])]()
        timing_call = cam.prepared_batch(scene, poses, [frames_b[0][f].data_ptr() for f in range(F)], pitch)
        pipe = None
        my_frames = lambda unit: list(range(unit[0], unit[0] + unit[1]))      # frame indices (within the group) this rank holds afterwards
        frames_of = lambda b: frames_b[b]
    else:
        max_rows = g["max_rows"]
        slots = max(tiling.rotating_plan(c, world)[0] for _, c in shapes) if rotate else F
        on_dev = exchange.on_device
        mk = lambda shape, zero=False: (torch.zeros if zero else torch.empty)(shape, dtype=torch.uint8, device=dev)
        local_dev = [mk((slots * max_rows, pitch), True) for _ in range(2)]
        if rotate:
            recv_shape = (world * max(tiling.rotating_plan(c, world)[1][rank] for _, c in shapes) * max_rows, pitch)
            gathered_dev = [mk(recv_shape) for _ in range(2)]
        else:
            gathered_dev = [mk((world, F * max_rows, pitch)) if rank == 0 else None for _ in range(2)]
        frames_b = [mk((F, H, pitch)) for _ in range(2)]
        local = local_dev if on_dev else [torch.zeros_like(t, device="cpu") for t in local_dev]
        gathered = gathered_dev if on_dev else [torch.empty_like(t, device="cpu") if t is not None else None for t in gathered_dev]
        local_ptrs = [tiling.batch_local_ptrs(local_dev[b].data_ptr(), max(slots, F if not rotate else 0), max_rows, pitch) for b in range(2)]
        # the stripe owner rotates over the frames of a group: frame f of the group is rendered as owner (rank + f) % world
        # (a sub-group that starts at frame `first` of its group goes on where the one before stopped: first_frame = first)
        owner_rotation = not args.no_owner_rotation
        stripes_of = lambda first: (STRIPE_ROWS, rank, world, first) if owner_rotation else (STRIPE_ROWS, rank, world)
        calls = {(b, first, c): cams[b].prepared_batch(scene, poses[first:first + c], local_ptrs[b][:c], pitch, stripes=stripes_of(first))
                 for b in range(2) for first, c in shapes}
        timing_call = cam.prepared_batch(scene, poses[:units[0][1]], local_ptrs[0][:units[0][1]], pitch, stripes=stripes_of(0))
        group_unit = [units[0], units[0]]                        # the (first frame, frames) that currently occupies buffer set b

        def my_frames(unit):                                     # frame indices (within the group) this rank assembles of a unit
            first, c = unit
            if rotate:
                _, cnt, off, real = tiling.rotating_plan(c, world)
                return [first + off[rank] + k for k in range(real[rank])]
            return list(range(first, first + c)) if rank == 0 else []

        def render_fn(b):
            calls[(b,) + tuple(group_unit[b])]()
            if not on_dev:
                local[b].copy_(local_dev[b])                    # (synchronous: the rehearsal stages through the host)

        def exchange_fn(b):
            if rotate:
                exchange.rotating(local[b], gathered[b], group_unit[b][1], max_rows)
            else:
                exchange.to_root(local[b], gathered[b] if rank == 0 else None, 0)

        def unstripe_fn(b):
            n = len(my_frames(group_unit[b]))
            if n == 0:
                return
            if not on_dev:
                gathered_dev[b].copy_(gathered[b])
            # rank r's block holds its stripes of my frames (rotate) or of all F frame slots (root0)
            plan = tiling.rotating_plan(group_unit[b][1], world) if rotate else None
            rank_stride = (plan[1][rank] if rotate else F) * max_rows * pitch
            if owner_rotation:                                  # my first frame's index in the group: which owner each source rank played
                first = group_unit[b][0] + (plan[2][rank] if rotate else 0)
                rt.check(hlib.rt_unstripe_batch_rotating(gathered_dev[b].data_ptr(), pitch, rank_stride, max_rows * pitch, frames_b[b].data_ptr(), pitch,
                                                         H * pitch, n, W, H, STRIPE_ROWS, world, first, torch.cuda.current_stream().cuda_stream))
            else:
                rt.check(hlib.rt_unstripe_batch(gathered_dev[b].data_ptr(), pitch, rank_stride, max_rows * pitch, frames_b[b].data_ptr(), pitch,
                                                H * pitch, n, W, H, STRIPE_ROWS, world, torch.cuda.current_stream().cuda_stream))

        pipe = tiling.StripePipeline(render_fn, exchange_fn, unstripe_fn, assembles=rotate or rank == 0,
                                     compute_streams=cstreams, comm_stream=torch.cuda.Stream() if on_dev else None)

        def step_group(i, unit):
            group_unit[i & 1] = unit                             # (read by the callbacks while they issue unit i, i.e. inside step())
            pipe.step(i)
        frames_of = lambda b: frames_b[b]

    sync = g["sync"]
    for i, u in enumerate(warm_units):
        step_group(i, u)
    sync()
    if pipe is not None:
        # per unit: wait for buffer / render / exchange / un-stripe (events created here, only recorded inside the timed region)
        pipe.stage_timing(True, expect_groups=len(units) * (REPEATS_SHORT if len(groups) == 1 else 1))
    g["phase"]("timed loop", 4.0)
    # The timed region = the K steps between two barrier + synchronise pairs.  When K fits ONE launch (the driver's
    # --steps 20) that region is a single 2.7 ms sample: it is then measured REPEATS_SHORT times back to back and the line
    # reports the median (with min / max / repeats); still a few tens of milliseconds in all.
    repeats = REPEATS_SHORT if len(groups) == 1 else 1
    dts, t_issue = [], 0.0
    for rep in range(repeats):
        t0 = time.perf_counter()
        for i, u in enumerate(units):
            step_group(i + rep * len(units), u)
        t_issue = time.perf_counter() - t0                       # host time to issue all groups (must stay below the GPU's)
        if pipe is not None:
            pipe.drain()
        sync()
        dts.append(time.perf_counter() - t0)
    if dist_on:
        tmax = torch.tensor(dts, dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)              # per repeat: the slowest rank
        dts = [float(v) for v in tmax.tolist()]
    timing = aggregate_repeats(dts, args.steps)
    dt = timing["dt"]
    # where a rank's time went, per group of F frames: every rank reports, rank 0 prints all of them
    per_rank = None
    if pipe is not None:
        mine_t = dict(pipe.stage_times(groups_per_region=len(units)) or {}, rank=rank)
        pipe.stage_timing(False)
        box = [None] * world
        if world > 1:
            dist.all_gather_object(box, mine_t)
        else:
            box = [mine_t]
        per_rank = box

    # ---- kernel-only duration with hipEvents on the launch stream (roofline denominator); latency figures ----
    g["phase"]("check", 4.0)
    kernel_ms, latency = None, None
    if rank == 0:
        n = max(10, min(len(groups), 100))
        torch.cuda.synchronize()
        timer.start(stream)
        for _ in range(n):
            timing_call()
        timer.stop(stream)
        kernel_ms = timer.elapsed_ms() / n                      # one launch = FL frames (this rank's stripes of them)
        if not dist_on and not args.no_latency:
            latency = measure_latency(g, poses, kernel_ms / F)
    FL = units[0][1] if dist_on else F                           # frames of the launch kernel_ms times
    if dist_on:
        dist.barrier()

    # ---- every rank checks frames it assembled in the last unit -- and, when the region is a pipeline of sub-groups, in the one before
    # it, which sits in the other buffer set -- against the debug kernel at the same pose ----
    frame_ok, ids_ok, dbg0 = True, True, None
    total_units = repeats * len(units)
    for back in range(2 if (subgrouped and total_units >= 2) else 1):
        unit = units[(len(units) - 1 - back) % len(units)]
        held = frames_of((total_units - 1 - back) & 1).cpu().numpy().reshape(F, H, W, 3)
        mine = my_frames(unit)
        for k, f in enumerate(mine):
            if k not in (0, len(mine) // 2, len(mine) - 1):
                continue
            cam.set_pose(poses[f])
            cam.set_stream(stream)
            dbg = rt.render_debug(scene, cam)
            slot = k if (dist_on and rotate) else f - (unit[0] if dist_on else 0)    # rotating exchange: my k-th frame sits in slot k of my frame set
            frame_ok = frame_ok and bool(np.array_equal(held[slot], dbg["img"]))
            ids = rt.render_ids(scene, cam)                      # the production kernel's own hit ids
            ids_ok = ids_ok and bool(np.array_equal(ids["hit_tri"], dbg["hit_tri"]) and np.array_equal(ids["hit_inst"], dbg["hit_inst"]))
    if dist_on:
        flag = torch.tensor([int(frame_ok), int(ids_ok)], dtype=torch.int32, device="cpu" if rehearsal else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        frame_ok, ids_ok = bool(flag[0].item()), bool(flag[1].item())
    if rank != 0:
        return None

    cam.set_pose(poses[0])
    dbg = rt.render_debug(scene, cam)                           # per-frame work counters (same traversal, instrumented)
    st = {"rays": W * H, "pops": int(dbg["pops"].sum()), "aabb": int(dbg["aabb"].sum()), "tris": int(dbg["tris"].sum()),
          "inside": int(dbg["inside"].sum()), "hits": int((dbg["hit_tri"] >= 0).sum())}
    alg = algorithmic_bytes(st)
    mesh, wl = g["mesh"], g["wl"]
    name = {"c2": "C2 bunny-class blob OBJ", "c4": "C4 Sponza-class atrium OBJ", "c6": "C6 atrium OBJ at 16 x the triangles of C4 (records exceed the Infinity Cache)",
            "demo": "the reference demo's scene shape (kernel.cu:166-240): two OBJ meshes, two textured materials, the second instance %s"
                    % ("with its translation baked into the vertices (identity instance: the comparison twin)" if args.baked else "translated by (-0.6, 1.48, 0.73)")}.get(args.workload, args.workload)
    config = {"workload": "%s (%d tris, %d BVH nodes), %dx%d, 1 primary ray/pixel, camera '%s' %s; every frame of a group has its own pose "
                          "(a 4 mm loop around that camera)" % (name, g["scene_tris"], g["scene_nodes"], W, H, g["cam_name"], str(tuple(base_pose[:3]))),
              "key": g["key"] + ("_f1" if F == 1 and not dist_on else ""), "width": W, "height": H, "spp": 1, "bounces": 0, "lighting": 0,
              "parallelism": "replicated scene, %d-row stripes round-robin over %d GPU(s)%s%s"
                             % (STRIPE_ROWS, world, (", the stripe owner rotating over the frames of a group" if dist_on and not args.no_owner_rotation else ""),
                                (", one RCCL all-to-all per %d frames (the gather's root rotates: each rank assembles 1/N of the frames) through %s" % (F, "rt_all_to_all" if g["comm"] is not None else "torch.distributed") if rotate
                                                  else ", one RCCL gather to rank 0 per %d frames through %s" % (F, "rt_gather" if g["comm"] is not None else "torch.distributed")) if dist_on else ""),
              "mesh_on_octant_loops": all(scene.mesh_flags(k) == 0 for k in range(len(g["meshes"]))),      # (false: an unordered or NaN child box keeps a mesh on the generic slab loop)
              "frames_per_launch": FL, "frames_per_group": F, "host_issue_ms_per_launch": round(t_issue / max(len(units), 1) * 1e3, 3),
              "sub_groups": [list(u) for u in units] if subgrouped else None,
              "single_frame_launch_ms": None if latency is None else latency["f1_kernel_ms"], "latency": latency,
              "coverage": round(st["hits"] / st["rays"], 4),
              "per_ray": {k: round(st[k] / st["rays"], 3) for k in ("pops", "aabb", "tris", "inside")},
              "algorithmic_bytes_per_ray": round(alg / st["rays"], 1)}
    # one frame per launch goes through the heavy-first variant of the kernel: its own profile entry
    single = F == 1 and not dist_on
    # (fourth template argument: the kernel with the private overflow of the traversal stack, for trees deeper than its LDS part)
    spill = scene.info()["max_stack"] - 1 > 16 or os.environ.get("RT_STACK_SPILL", "")[:1] == "1"
    # (fifth: the launches of this run rendered through view records -- batches of four and more frames, rt_scene_view_stats; the
    # pre-pass that writes them is part of kernel_ms)
    views = scene.view_stats()
    config["view_records"] = views
    if not dist_on:
        # which traversal loop the waves of such a launch run (rt_scene_loop_stats: the same launch through an instrumented copy of the
        # kernel, after the timings): the hand-written gfx950 loop should carry (nearly) every wave x instance cast
        loops = scene.loop_stats(cam, poses[:F], [frames_of(0)[f].data_ptr() for f in range(F)], pitch, stream)
        cam.set_pose(poses[0])
        config["traversal_loops"] = loops
        config["asm_loop_frac"] = loops["asm_loop_frac"]
    # (third template argument: the heavy-first dispatch order -- single frames, and a rank's stripes of fewer than four frames' worth of tiles)
    thin = dist_on and world > 1 and FL > 1 and os.environ.get("RT_TILE_ORDER_STRIPES", "1")[:1] != "0" and \
        ((W + 7) // 8) * ((max(tiling.stripe_rows(H, STRIPE_ROWS, r, world) for r in range(world)) + 7) // 8) * FL < 4 * 32768
    roof = roofline("render_kernel<false,false,%s,%s,%s>" % ("true" if single or thin else "false", "true" if spill else "false",
                                                               "true" if views["launches"] > views["fallbacks"] and F >= 4 else "false"),
                    g["key"] + ("_f1" if single else ""), kernel_ms, FL, 1.0 / world, alg)
    # ms_per_step is the throughput figure of a batch (F frames per launch); what one frame takes on its own is spelled out next to it
    extra = {"frames_per_launch": FL,
             "ms_per_frame_single_launch": None if latency is None else latency["f1_kernel_ms"],
             "ms_per_frame_reference_loop": None if latency is None else latency["reference_loop_2_renders_per_sync_wall_ms_per_frame"],
             "frame_matches_debug_kernel": frame_ok, "production_hit_ids_match_debug_kernel": ids_ok}
    extra.update(timing["fields"])
    if per_rank is not None:
        extra.update(multi_rank_report(per_rank, g["comm"]))
    value = W * H * args.steps / dt / 1e6
    out = base_line(args, g, value, dt, sum(c for _, c in warm_units), config, roof, extra)
    if not dist_on and not args.no_cpu_baseline:
        cb, ref_img = cpu_baseline_stream(g["obj"], wl, W, H, K, D, poses[0], st, args.workload, args.baked)
        cb["gpu_frame_matches_oracle"] = bool(np.array_equal(ref_img, dbg["img"]))
        out["cpu_baseline"] = cb
    return out


def scaling_prediction(t1_ms, rank_ms):
    """One N of a prediction: t1_ms = what one GPU takes for the unit of work (a group of F frames, or one frame), rank_ms[r] =
    what virtual rank r of N takes for ITS share of the same unit.  The N-GPU run takes as long as its slowest rank."""
    worst, mean = max(rank_ms), sum(rank_ms) / len(rank_ms)
    n = len(rank_ms)
    return {"n_gpus": n, "render_ms_per_rank": [round(v, 4) for v in rank_ms], "one_gpu_ms": round(t1_ms, 4),
            "ideal_ms": round(t1_ms / n, 4), "stripe_share_imbalance": round(worst / mean, 4),
            "predicted_render_speedup": round(t1_ms / worst, 3), "predicted_render_efficiency": round(t1_ms / worst / n, 4)}


def predict_scaling(args, g):
    """The render side of an N-GPU run, measured on ONE GPU with the kernels as they are: every virtual rank r of N renders its
    stripes exactly as rank r of a real run would (same entry points, same launches, the pipeline's two compute streams), with no
    exchange and no un-stripe pass.  The slowest rank sets the pace of a real run, so T(one GPU) / max_r T_r bounds the speedup
    from above; what the exchange, RCCL's kernels and the host add is not in it."""
    W, H, K, D, pitch, scene, cam, dev = g["W"], g["H"], g["K"], g["D"], g["pitch"], g["scene"], g["cam"], g["dev"]
    base_pose, make_camera, stream_mode = g["base_pose"], g["make_camera"], g["stream_mode"]
    spp, bounces, lighting = g["spp"], g["bounces"], g["lighting"]
    Ns = sorted(set(int(v) for v in args.predict_scaling.split(",") if v.strip()))
    if not Ns or min(Ns) < 2:
        sys.exit("bench.py --predict-scaling: a list of GPU counts >= 2")
    out = {"metric": METRIC, "mode": "predict-scaling", "n_gpus": 1, "unit": "ms", "data": "synthetic", "dtype": "f32",
           "code_hash": rt.library_hash(), "stripe_rows": STRIPE_ROWS, "stripe_owner_rotates_over_frames": not args.no_owner_rotation,
           "what": "render side only, every virtual rank of N timed on one GPU through the entry points a real rank uses; "
                   "predicted_render_speedup = one-GPU time / slowest rank's time; exchange, un-stripe and host costs are not included"}
    cstreams = [torch.cuda.Stream(), torch.cuda.Stream()]
    if stream_mode:
        F = 32
        poses = camera_path(base_pose, F)
        G = max(6, min(48, args.steps // F if args.steps else 24))
        K20 = 20                                                # the driver's --steps 20: ONE group of 20 frames between synchronises
        cams = [make_camera(cstreams[0]), make_camera(cstreams[1])]

        def hold_clock(call):
            for _ in range(max(8, MIN_WARM_FRAMES // F)):
                call(0)
            torch.cuda.synchronize()

        def time_groups(call, groups, streams):                 # call(b) issues one group into buffer set b on cams[b]'s stream
            hold_clock(call)
            t0 = time.perf_counter()
            for i in range(groups):
                call(i % streams)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3 / groups

        def time_single(call20, reps=9):                        # the driver's shape: one group, synchronised on both sides; median
            hold_clock(lambda b: call20())
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                call20()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            return sorted(ts)[len(ts) // 2]

        # one GPU: whole frames (what `bench.py --gpus 1` times: one stream; two alternating streams beside it)
        frames_b = [torch.empty((F, H, pitch), dtype=torch.uint8, device=dev) for _ in range(2)]
        whole = [cams[b].prepared_batch(scene, poses, [frames_b[b][f].data_ptr() for f in range(F)], pitch) for b in range(2)]
        whole20 = cams[0].prepared_batch(scene, poses[:K20], [frames_b[0][f].data_ptr() for f in range(K20)], pitch)
        t1_one = time_groups(lambda b: whole[0](), G, 1)
        t1_two = time_groups(lambda b: whole[b](), G, 2)
        t1_20 = time_single(whole20)
        out["stream"] = {"workload": "%s, %dx%d, 1 primary ray/pixel, camera '%s'; groups of F = %d frames, each with its own pose"
                                     % (args.workload, W, H, g["cam_name"], F),
                         "frames_per_group": F, "groups_timed": G,
                         "one_gpu_ms_per_group": {"one_stream": round(t1_one, 4), "two_alternating_streams": round(t1_two, 4)},
                         "one_gpu_ms_driver_shape_20_frames": round(t1_20, 4), "per_n": [], "driver_shape_per_n": [], "driver_shape_one_launch_per_n": []}
        del frames_b
        for N in Ns:
            max_rows = max(tiling.stripe_rows(H, STRIPE_ROWS, r, N) for r in range(N))
            local = [torch.zeros((F * max_rows, pitch), dtype=torch.uint8, device=dev) for _ in range(2)]
            ptrs = [tiling.batch_local_ptrs(local[b].data_ptr(), F, max_rows, pitch) for b in range(2)]
            two, one, single, more, piped = [], [], [], [], []
            for r in range(N):
                own = (STRIPE_ROWS, r, N) if args.no_owner_rotation else (STRIPE_ROWS, r, N, 0)
                calls = [cams[b].prepared_batch(scene, poses, ptrs[b], pitch, stripes=own) for b in range(2)]
                call20 = cams[0].prepared_batch(scene, poses[:K20], ptrs[0][:K20], pitch, stripes=own)
                # ... and as the pipeline of sub-groups rank r of N issues for it (run_stream: tiling.sub_groups; alternating compute streams)
                subs = tiling.sub_groups(K20, N, args.subgroups or 4) if args.subgroups != 1 else [(0, K20)]
                sub_calls = [cams[j & 1].prepared_batch(scene, poses[first:first + c], ptrs[j & 1][:c], pitch,
                                                        stripes=(STRIPE_ROWS, r, N) if args.no_owner_rotation else (STRIPE_ROWS, r, N, first))
                             for j, (first, c) in enumerate(subs)]

                def call20_sub(sub_calls=sub_calls):
                    for c in sub_calls:
                        c()
                two.append(time_groups(lambda b: calls[b](), G, 2))
                if os.environ.get("PREDICT_STREAMS"):            # experiment: more than two compute streams
                    ns = int(os.environ["PREDICT_STREAMS"])
                    xs = [torch.cuda.Stream() for _ in range(ns)]
                    xc = [make_camera(x) for x in xs]
                    xl = [torch.zeros((F * max_rows, pitch), dtype=torch.uint8, device=dev) for _ in range(ns)]
                    xcalls = [xc[b].prepared_batch(scene, poses, tiling.batch_local_ptrs(xl[b].data_ptr(), F, max_rows, pitch), pitch, stripes=own) for b in range(ns)]
                    more.append(time_groups(lambda b: xcalls[b](), G, ns))
                one.append(time_groups(lambda b: calls[0](), G, 1))
                single.append(time_single(call20))
                piped.append(time_single(call20_sub))
            # the one-GPU side of the ratio is what `bench.py --gpus 1` measures (one stream); a real N-rank run renders on two
            pn = scaling_prediction(t1_one, two)
            pn["render_ms_per_rank_one_stream"] = [round(v, 4) for v in one]
            pn["predicted_render_speedup_one_stream_ranks"] = round(t1_one / max(one), 3)
            if more:
                pn["experiment_%s_streams_ms_per_rank" % os.environ["PREDICT_STREAMS"]] = [round(v, 4) for v in more]
            out["stream"]["per_n"].append(pn)
            # the driver's `--gpus N --steps 20`: one group of 20 frames per timed region.  `driver_shape_per_n` = as a rank renders it now
            # (sub-groups on two streams, heavy-first order for thin stripes); `driver_shape_one_launch_per_n` = one launch (rounds 1-5)
            dn = scaling_prediction(t1_20, piped)
            dn["sub_groups"] = [list(u) for u in subs]
            out["stream"]["driver_shape_per_n"].append(dn)
            out["stream"]["driver_shape_one_launch_per_n"].append(scaling_prediction(t1_20, single))
            log("bench.py --predict-scaling: N=%d stream: %s" % (N, json.dumps(pn)))
            del local
        last = out["stream"]["per_n"][-1]
        out["value"], out["predicted_for"] = last["predicted_render_speedup"], "N=%d, stream of F=%d-frame groups" % (last["n_gpus"], F)
    else:
        cam.set_options(spp, bounces, lighting)
        frame = torch.empty((H, pitch), dtype=torch.uint8, device=dev)
        reps = max(2, min(args.steps, 5))

        def time_frame(call):
            call()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3 / reps
        t1 = time_frame(lambda: cam.render_scene(scene, frame.data_ptr(), pitch))
        out["frame"] = {"workload": "%s, %dx%d, %d spp, %d bounces, lighting %d, camera '%s'" % (args.workload, W, H, spp, bounces, lighting, g["cam_name"]),
                        "frames_timed_per_figure": reps, "one_gpu_ms_per_frame": round(t1, 3), "per_n": []}
        for N in Ns:
            max_rows = max(tiling.stripe_rows(H, STRIPE_ROWS, r, N) for r in range(N))
            local = torch.zeros((max_rows, pitch), dtype=torch.uint8, device=dev)
            ranks = [time_frame(lambda r=r: cam.render_scene_stripes(scene, local.data_ptr(), pitch, STRIPE_ROWS, r, N)) for r in range(N)]
            pn = scaling_prediction(t1, ranks)
            out["frame"]["per_n"].append(pn)
            log("bench.py --predict-scaling: N=%d frame: %s" % (N, json.dumps(pn)))
        last = out["frame"]["per_n"][-1]
        out["value"], out["predicted_for"] = last["predicted_render_speedup"], "N=%d, one frame per step" % last["n_gpus"]
    out["higher_is_better"] = True
    return out


def measure_latency(g, poses, f32_ms_per_frame):
    """What batching hides: one frame per launch, two frames per launch, and the reference's own loop
    (two render calls, then a device synchronise: kernel.cu:277-279).  Every frame has its own pose."""
    W, H, scene, cam, stream, timer, dev, pitch = g["W"], g["H"], g["scene"], g["cam"], g["stream"], g["timer"], g["dev"], g["pitch"]
    bufs = torch.empty((2, H, pitch), dtype=torch.uint8, device=dev)
    n = 40
    singles = [cam.prepared_batch(scene, [poses[k % len(poses)]], [bufs[k & 1].data_ptr()], pitch) for k in range(n)]
    pairs = [cam.prepared_batch(scene, [poses[(2 * k) % len(poses)], poses[(2 * k + 1) % len(poses)]], [bufs[0].data_ptr(), bufs[1].data_ptr()], pitch)
             for k in range(n // 2)]
    # warm-up: single-frame launches dispatch their tiles in the order sorted from an earlier frame's costs (heavy-first), and a
    # new order is picked up by the first launch issued after its sort has finished -- synchronise between the first launches so
    # that the timed ones run with a settled order, as any frame of a running application does
    def hold_clock():
        # (building the launch descriptors above left the GPU idle for a while: ~45 ms of load before a timed section, for the
        # same reason as the warm-up floor of the main loop -- the chip needs tens of milliseconds of load to hold its clock)
        for _ in range(7):
            for c in pairs:
                c()
        torch.cuda.synchronize()
    hold_clock()
    for c in singles[:8]:
        c()
        torch.cuda.synchronize()
    timer.start(stream)
    for c in singles:
        c()
    timer.stop(stream)
    f1 = timer.elapsed_ms() / n
    timer.start(stream)
    for c in pairs:
        c()
    timer.stop(stream)
    f2 = timer.elapsed_ms() / n
    hold_clock()
    t0 = time.perf_counter()
    for k in range(0, n, 2):
        singles[k]()
        singles[k + 1]()
        torch.cuda.synchronize()
    ref_loop_one_stream = (time.perf_counter() - t0) * 1e3 / n
    # one frame per launch, consecutive frames on two alternating streams (an application that double-buffers its frames): the next
    # frame's costly tiles fill the chip while the previous frame's last workgroups drain.  Measured BEFORE the first default-stream
    # render_scene of this process and again after it: rt_render_overlapped creates a HIGH-PRIORITY stream for the first frame of a
    # pair, and on this runtime the existence of one slows two normal streams that alternate (0.112 -> 0.125 ms per frame here; see
    # rt_hip.h, RT_OVERLAP_PRIORITY) -- both figures are in the line
    make_camera = g["make_camera"]
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    cams2 = [make_camera(side[0]), make_camera(side[1])]
    m = 160
    alt = [cams2[k & 1].prepared_batch(scene, [poses[k % len(poses)]], [bufs[k & 1].data_ptr()], pitch) for k in range(m)]

    def two_alternating_streams():
        hold_clock()
        for c in alt[:8]:
            c()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in alt:
            c()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / m
    f1_two = two_alternating_streams()
    # The reference's loop as its application writes it (kernel.cu:275-279): camera.pose = ...; camera.render_scene(scene, d_img,
    # pitch); camera.render_scene(scene, d_img2, pitch); cudaDeviceSynchronize() -- through Camera::render_scene ITSELF, on the
    # default stream: the library lets the two frames overlap (rt_render_overlapped), nothing here helps it
    ref_cam = rt.Camera(W, H, g["K"], g["D"])                    # (no stream set: the default stream, as in the reference)
    ref_calls = [ref_cam.prepared_render(scene, poses[k % len(poses)], bufs[k & 1].data_ptr(), pitch) for k in range(n)]
    for c in ref_calls[:8]:
        c()
    torch.cuda.synchronize()
    hold_clock()
    t0 = time.perf_counter()
    for k in range(0, n, 2):
        ref_calls[k]()
        ref_calls[k + 1]()
        torch.cuda.synchronize()
    ref_loop = (time.perf_counter() - t0) * 1e3 / n
    ref_frames = {k: bufs[k & 1].cpu().numpy().reshape(H, W, 3) for k in (n - 2, n - 1)}   # the loop's last two frames (checked after the timings)
    ov = scene.overlap_stats()
    hold_clock()
    t0 = time.perf_counter()
    for k in range(n):
        singles[k]()
        torch.cuda.synchronize()
    f1_sync = (time.perf_counter() - t0) * 1e3 / n
    f1_two_after = two_alternating_streams()
    pcie = measure_download(g, poses)
    ref_ok = True
    for k, got in ref_frames.items():
        cam.set_pose(poses[k % len(poses)])
        ref_ok = ref_ok and bool(np.array_equal(got, rt.render_debug(scene, cam)["img"]))
    return {"f1_kernel_ms": round(f1, 4), "f1_launch_plus_sync_wall_ms": round(f1_sync, 4), "f2_batch_ms_per_frame": round(f2, 4),
            "pcie_inclusive": pcie,
            "f1_two_alternating_streams_wall_ms_per_frame": round(f1_two, 4),
            "f1_two_alternating_streams_after_the_default_stream_frames_wall_ms_per_frame": round(f1_two_after, 4),
            "reference_loop_2_renders_per_sync_wall_ms_per_frame": round(ref_loop, 4),
            "reference_loop_through": "Camera::render_scene(scene, img, pitch) twice into two images on the default stream, then a device synchronise "
                                      "(kernel.cu:277-279); the library alternates such frames between two blocking streams of its own (rt_render_overlapped)",
            "reference_loop_frames_match_debug_kernel": ref_ok,
            "reference_loop_overlapped_launches": ov[0], "reference_loop_cross_stream_waits": ov[1],
            "reference_loop_one_stream_wall_ms_per_frame": round(ref_loop_one_stream, 4), "f32_batch_kernel_ms_per_frame": round(f32_ms_per_frame, 4),
            "note": "kernel ms = hipEvents around back-to-back launches on one stream; wall ms include launch and hipDeviceSynchronize"}


def measure_download(g, poses, F=32, groups=12):
    """The PCIe-inclusive rate (never `value`): every frame also travels to the host, as display_image does (kernel.cu:33-37).
    Batches of F frames into two device frame sets; each set is copied to pinned host memory on a copy stream while the next
    batch renders (events order render -> copy -> re-use of the set); and, for comparison, the same with render and copy in turn
    on one stream."""
    W, H, scene, dev, pitch, make_camera = g["W"], g["H"], g["scene"], g["dev"], g["pitch"], g["make_camera"]
    F = min(F, len(poses))
    try:
        host = [torch.empty((F, H, pitch), dtype=torch.uint8).pin_memory() for _ in range(2)]
    except RuntimeError as e:
        return {"note": "pinned host memory unavailable: %s" % e}
    sets = [torch.empty((F, H, pitch), dtype=torch.uint8, device=dev) for _ in range(2)]
    rs, cs = torch.cuda.Stream(), torch.cuda.Stream()
    cam = make_camera(rs)
    calls = [cam.prepared_batch(scene, poses[:F], [sets[b][f].data_ptr() for f in range(F)], pitch) for b in range(2)]
    rendered = [torch.cuda.Event() for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]

    def run(overlap):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(groups):
            b = i & 1
            if overlap:
                if i >= 2:
                    rs.wait_event(copied[b])                    # the set is free again once its previous content has left
                calls[b]()
                rendered[b].record(rs)
                cs.wait_event(rendered[b])
                with torch.cuda.stream(cs):
                    host[b].copy_(sets[b], non_blocking=True)
                copied[b].record(cs)
            else:
                calls[b]()
                with torch.cuda.stream(rs):
                    host[b].copy_(sets[b], non_blocking=True)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / (groups * F)
    run(True)                                                    # warm-up (first touch of the pinned pages)
    over, serial = run(True), run(False)
    mb = F * H * W * 3 / 1e6
    return {"frames_per_launch": F, "overlapped_copy_stream_ms_per_frame": round(over, 4), "render_then_copy_one_stream_ms_per_frame": round(serial, 4),
            "overlapped_Mrays_per_s": round(W * H / over / 1e3, 1), "frame_MB": round(W * H * 3 / 1e6, 2),
            "note": "every frame copied to pinned host memory (tight rows, %.0f MB per batch); not the headline value" % mb}


def run_frames(args, env):
    """spp / bounces / lighting workloads (render_ex_kernel + resolve_ex_kernel): one frame per step."""
    g = env
    rank, world, dist_on, rehearsal, dev = g["rank"], g["world"], g["dist_on"], g["rehearsal"], g["dev"]
    W, H, K, D, pitch, scene, cam, stream, hlib, timer = g["W"], g["H"], g["K"], g["D"], g["pitch"], g["scene"], g["cam"], g["stream"], g["hlib"], g["timer"]
    spp, bounces, lighting, base_pose, exchange, comm = g["spp"], g["bounces"], g["lighting"], g["base_pose"], g["exchange"], g["comm"]
    poses = camera_path(base_pose, 8)
    frame = torch.empty((H, pitch), dtype=torch.uint8, device=dev)
    python_path = dist_on and comm is None                      # rehearsal (host-staged) or the torch.distributed fallback (device tensors)
    if python_path:
        max_rows = g["max_rows"]
        local_dev = torch.zeros((max_rows, pitch), dtype=torch.uint8, device=dev)
        gathered_dev = torch.empty((world, max_rows, pitch), dtype=torch.uint8, device=dev) if rank == 0 else None
        local = local_dev if exchange.on_device else torch.zeros_like(local_dev, device="cpu")
        gathered = gathered_dev if exchange.on_device else (torch.empty((world, max_rows, pitch), dtype=torch.uint8) if rank == 0 else None)

    def step(i):
        cam.set_pose(poses[i % len(poses)])
        if not dist_on:
            cam.render_scene(scene, frame.data_ptr(), pitch)                            # Camera::render_scene -> rt_render_ex
        elif not python_path:
            cam.render_scene_tiled(scene, comm, frame.data_ptr(), pitch, stripe_rows=STRIPE_ROWS, root=0)   # stripes + rt_gather + un-stripe
        else:
            cam.render_scene_stripes(scene, local_dev.data_ptr(), pitch, STRIPE_ROWS, rank, world, synchronize=not exchange.on_device)
            if not exchange.on_device:
                local.copy_(local_dev)
            exchange.to_root(local, gathered, 0)
            if rank == 0:
                if not exchange.on_device:
                    gathered_dev.copy_(gathered)
                rt.check(hlib.rt_unstripe(gathered_dev.data_ptr(), pitch, max_rows * pitch, frame.data_ptr(), pitch, W, H, STRIPE_ROWS, world, stream))

    sync = g["sync"]
    for i in range(args.warmup):
        step(i)
    sync()
    g["phase"]("timed loop", 4.0)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync()
    dt = time.perf_counter() - t0
    g["phase"]("check", 4.0)
    if dist_on:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    last_pose = poses[(args.steps - 1) % len(poses)]
    got = frame.cpu().numpy().reshape(H, W, 3) if rank == 0 else None            # the last timed frame (before the probe below reuses the buffer)

    # kernel time of this rank's share of one frame (render_ex_kernel launches + resolve passes), hipEvents
    kernel_ms = None
    if rank == 0:
        n = max(3, min(args.steps, 20))
        local_probe = torch.empty((g["max_rows"] if dist_on else 1, pitch), dtype=torch.uint8, device=dev) if dist_on else None
        cam.set_pose(poses[0])
        torch.cuda.synchronize()
        timer.start(stream)
        for _ in range(n):
            if dist_on:
                cam.render_scene_stripes(scene, local_probe.data_ptr(), pitch, STRIPE_ROWS, rank, world)
            else:
                cam.render_scene(scene, frame.data_ptr(), pitch)
        timer.stop(stream)
        kernel_ms = timer.elapsed_ms() / n
    if dist_on:
        dist.barrier()
    if rank != 0:
        return None

    # the (tiled) frame must equal the frame one GPU renders alone: rank 0 renders it again, whole
    cam.set_pose(last_pose)
    whole = rt.render_ex(scene, cam)
    frame_ok = bool(np.array_equal(got, whole["img"]))
    pops = int(whole["total_pops"].astype(np.int64).sum())
    mesh, wl = g["mesh"], g["wl"]
    name = {"c3": "C3 bunny-class blob OBJ", "c4": "C4 Sponza-class atrium OBJ", "c5": "C5 bunny-class blob OBJ", "c6": "C6 atrium OBJ at 16 x the triangles of C4"}.get(args.workload, args.workload)
    config = {"workload": "%s (%d tris, %d BVH nodes), %dx%d, %d spp, %d specular bounces, lighting %d (material roughness %.2f metallic %.2f), camera '%s' %s"
                          % (name, mesh.num_triangles, mesh.num_nodes, W, H, spp, bounces, lighting, wl.get("roughness", 0.0), wl.get("metallic", 0.0),
                             g["cam_name"], str(tuple(base_pose[:3]))),
              "key": g["key"], "width": W, "height": H, "spp": spp, "bounces": bounces, "lighting": lighting,
              "parallelism": "replicated scene, %d-row stripes round-robin over %d GPU(s)%s"
                             % (STRIPE_ROWS, world, (", one RCCL gather to rank 0 per frame (Camera::render_scene_tiled -> rt_render_tiled)" if comm is not None else ", one gather to rank 0 per frame through torch.distributed") if dist_on else ""),
              "mesh_on_octant_loops": scene.mesh_flags(0) == 0,
              "frames_per_launch": 1, "primary_rays_per_frame": W * H * spp,
              "node_pops_per_pixel": round(pops / (W * H), 1), "G_node_pops_per_s_kernel": round(pops / max(world, 1) / (kernel_ms * 1e-3) / 1e9, 1)}
    roof = roofline("render_ex_kernel", g["key"], kernel_ms, 1, 1.0 / world)
    extra = {"frame_matches_single_gpu_render": frame_ok}
    value = W * H * spp * args.steps / dt / 1e6
    out = base_line(args, g, value, dt, args.warmup, config, roof, extra)
    if not dist_on and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_frame(g["obj"], wl, W, H, K, D, last_pose, (spp, bounces, lighting), got)
    return out


if __name__ == "__main__":
    main()
