#!/usr/bin/env python3
"""Load-to-ready: OBJ file -> scene on the device that a render can use, three ways (70 k-triangle blob, 260 k-triangle atrium):
  host     OBJLoader::load            (parse, host BVH builder)            + Scene::upload_to_device
  gpu      MeshPrimitive(.., true)    (parse, GPU build -> host arrays)    + Scene::upload_to_device
  device   OBJLoader::load_for_device (parse only)                         + Scene::upload_to_device builds the tree in place
   python tools/load_bench.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("cuda-raytracing_amd")
h = rt.libs()[0]
for name in ("blob70k", "atrium"):
    path = os.path.join(ROOT, ".scene_cache", name + ".obj")
    for mode, kw in (("host", {}), ("gpu", {"gpu_build": True}), ("device", {"for_device": True})):
        ts = []
        for _ in range(2 if mode == "host" else 7):
            rt.check(h.rt_device_synchronize())
            t0 = time.perf_counter()
            mesh = rt.Mesh.load_obj(path, **kw)
            t1 = time.perf_counter()
            sp = rt.Scene(); sp.add_material((0.9, 0.5, 0.2)); sp.add_mesh(mesh); sp.add_mesh_instance(0, 0); sp.upload_to_device()
            rt.check(h.rt_device_synchronize())
            t2 = time.perf_counter()
            ts.append((t2 - t0, t1 - t0, t2 - t1))
            sp.close()
        ts.sort()
        tot, load, up = ts[len(ts) // 2]
        print("%-8s %-7s load-to-ready %7.2f ms  (mesh %.2f ms, scene upload %.2f ms)" % (name, mode, tot * 1e3, load * 1e3, up * 1e3), flush=True)
