#!/usr/bin/env python3
"""Experiment driver for single-frame launches (RT_SINGLE): block size, priority boost, heavy-first dispatch order.
   python tools/single_exp.py [--iters 200]"""
import argparse, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("cuda-raytracing_amd")
scenes = importlib.import_module("cuda-raytracing_amd.scenes")

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--cams", default="far,mid,near")
a = ap.parse_args()
rt.build()
obj = os.path.join(ROOT, ".scene_cache", "blob70k.obj")
if not os.path.exists(obj):
    os.makedirs(os.path.dirname(obj), exist_ok=True)
    scenes.write_blob_obj(obj, 188, 187)
mesh = rt.Mesh.load_obj(obj)
scene = rt.Scene(); scene.add_material(scenes.C2["albedo"]); scene.add_mesh(mesh); scene.add_mesh_instance(0, 0); scene.upload_to_device()
W, H = 1920, 1080
img = rt.DeviceBuffer(width_bytes=W * 3, height=H)
t = rt.Timer()

def timeit(cam):
    for _ in range(5):
        cam.render_scene(scene, img.ptr, img.pitch)
    t.start()
    for _ in range(a.iters):
        cam.render_scene(scene, img.ptr, img.pitch)
    t.stop()
    return t.elapsed_ms() / a.iters

for name in a.cams.split(","):
    cam = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF)
    cam.set_pose(scenes.C2_CAMERAS[name])
    os.environ.pop("RT_SINGLE", None)
    base = timeit(cam)
    ref = img.to_host().copy()
    print("%-5s production render_kernel F=1: %.4f ms" % (name, base), flush=True)
    for blk in (256, 64):
        cost_f, order_f = "/tmp/cost_%s_%d.bin" % (name, blk), "/tmp/order_%s_%d.bin" % (name, blk)
        os.environ["RT_SINGLE"] = "%d,0,-,%s" % (blk, cost_f)
        cam.render_scene(scene, img.ptr, img.pitch, synchronize=True)
        assert np.array_equal(img.to_host(), ref)
        cost = np.fromfile(cost_f, np.int32)
        np.argsort(-cost, kind="stable").astype(np.int32).tofile(order_f)
        # a pose a little further along (what the next frame of a stream would see): the order is one frame old
        print("      block %3d: tiles %d, cost max %d mean %.1f p99 %d" % (blk, cost.size, cost.max(), cost.mean(), np.percentile(cost, 99)))
        for prio in (0, 48, 96, 160):
            for order in ("-", order_f):
                os.environ["RT_SINGLE"] = "%d,%d,%s" % (blk, prio, order)
                ms = timeit(cam)
                ok = np.array_equal(img.to_host(), ref)
                print("      block %3d prio_after %3d order %-6s: %.4f ms (%+.1f %%) %s" % (blk, prio, "heavy" if order != "-" else "natural", ms, (ms / base - 1) * 100, "" if ok else "MISMATCH"), flush=True)
