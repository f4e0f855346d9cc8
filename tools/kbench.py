#!/usr/bin/env python3
"""Kernel-only timing of the three C2 cameras (hipEvent, no torch): quick A/B tool for kernel work.
   python tools/kbench.py [--iters 100] [--width 1920 --height 1080] [--check]"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("cuda-raytracing_amd")
scenes = importlib.import_module("cuda-raytracing_amd.scenes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="blob70k", choices=["blob70k", "blob5k", "atrium"])
    ap.add_argument("--batch", type=int, default=1, help="frames per launch (rt_render_batch)")
    ap.add_argument("--ex", default=None, help="spp,bounces,lighting: time the extension kernel (rt_render_ex) instead")
    ap.add_argument("--streams", type=int, default=1, help="issue successive launches round-robin on this many HIP streams (torch streams)")
    ap.add_argument("--check", action="store_true", help="compare the frame hash with the debug kernel's")
    ap.add_argument("--path", type=int, default=0, help="single-frame launches cycle through this many poses on bench.py's 4 mm loop around the camera (0: one pose)")
    a = ap.parse_args()
    rt.build()
    cache = os.path.join(ROOT, ".scene_cache")
    os.makedirs(cache, exist_ok=True)
    obj = os.path.join(cache, a.scene + ".obj")
    if not os.path.exists(obj):
        {"blob70k": lambda p: scenes.write_blob_obj(p, 188, 187), "blob5k": lambda p: scenes.write_blob_obj(p, 50, 51),
         "atrium": scenes.write_atrium_obj}[a.scene](obj)
    mesh = rt.Mesh.load_obj(obj)
    scene = rt.Scene()
    scene.add_material(scenes.C4["albedo"] if a.scene == "atrium" else scenes.C2["albedo"], roughness=0.05 if a.ex else 0.0,
                       metallic=0.4 if a.ex else 0.0)
    scene.add_mesh(mesh)
    scene.add_mesh_instance(0, 0)
    scene.upload_to_device()
    print("scene", a.scene, "tris", mesh.num_triangles, "nodes", mesh.num_nodes, scene.info())
    W, H = a.width, a.height
    cams = {"atrium": {"inside": scenes.C4["cam_pose"]}}.get(a.scene, scenes.C2_CAMERAS)
    img = rt.DeviceBuffer(width_bytes=W * 3, height=H)
    extra = [rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(a.batch - 1)]
    ptrs = [img.ptr] + [e.ptr for e in extra]
    t = rt.Timer()
    for name, pose in cams.items():
        cam = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF)
        cam.set_pose(pose)
        if a.ex:
            spp, bounces, lighting = (int(v) for v in a.ex.split(","))
            cam.set_options(spp, bounces, lighting)

        if a.streams > 1:                                  # one camera object (and frame set) per stream
            import torch
            streams = [torch.cuda.Stream() for _ in range(a.streams)]
            cams, sets = [], []
            for st in streams:
                c = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF)
                c.set_pose(pose)
                c.set_stream(st.cuda_stream)
                cams.append(c)
                sets.append([rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(a.batch)])
            calls = [c.prepared_batch(scene, [pose] * a.batch, [b.ptr for b in bs], img.pitch) for c, bs in zip(cams, sets)]
            state = {"i": 0}

            def go():
                calls[state["i"] % a.streams]()
                state["i"] += 1
        else:
            path_calls, state1 = [], {"i": 0}
            if a.path > 0 and not a.ex and a.batch == 1:
                import math
                for k in range(a.path):
                    ang = 2.0 * math.pi * k / a.path
                    q = list(pose)
                    q[0] += 0.004 * math.sin(ang); q[2] += 0.004 * (math.cos(ang) - 1.0); q[3] += 0.002 * math.sin(ang)
                    path_calls.append(cam.prepared_batch(scene, [tuple(q)], [ptrs[0]], img.pitch))

            def go():
                if path_calls:
                    path_calls[state1["i"] % len(path_calls)]()
                    state1["i"] += 1
                elif a.ex or a.batch == 1:
                    cam.render_scene(scene, img.ptr, img.pitch)
                else:
                    cam.render_scene_batch(scene, [pose] * a.batch, ptrs, img.pitch)
        for _ in range(5):                                 # warm-up; single-frame launches also settle their dispatch order
            go()
            rt.check(rt.libs()[0].rt_device_synchronize())
        if a.streams > 1:                                  # events on one stream cannot bracket several: wall clock
            import time
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                go()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / a.iters / a.batch
        else:
            t.start()
            for _ in range(a.iters):
                go()
            t.stop()
            ms = t.elapsed_ms() / a.iters / a.batch
        line = "%-6s %.4f ms  %.1f Mrays/s" % (name, ms, W * H / ms / 1e3)
        if a.ex:
            import hashlib
            import numpy as np
            ex = rt.render_ex(scene, cam)
            tp = int(ex["total_pops"].astype(np.int64).sum())
            line += "  pops/pixel=%.1f  %.1f Gpops/s  img_sha=%s" % (tp / (W * H), tp / ms / 1e6, hashlib.sha1(ex["img"].tobytes()).hexdigest()[:12])
        if a.check and not a.ex:
            import numpy as np
            dbg = rt.render_debug(scene, cam)
            got = img.to_host().reshape(H, W, 3)
            line += "  match_debug=%s pops/ray=%.2f" % (bool(np.array_equal(got, dbg["img"])), dbg["pops"].mean())
        print(line, flush=True)


if __name__ == "__main__":
    main()
