#!/usr/bin/env python3
"""Time to a new tree for an uploaded mesh whose triangles changed (70 k-triangle blob, 260 k-triangle atrium), host-to-ready:
  device   rt_scene_rebuild_mesh_device: vertices already in device memory, tree built and emitted into the scene's arrays
  host     the route without it: rt_bvh_build (host arrays in and out) -> MeshPrimitive -> Scene::upload_to_device -> rt_scene_upload
   python tools/rebuild_bench.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
rt = importlib.import_module("cuda-raytracing_amd")
h = rt.libs()[0]
for name in ("blob70k", "atrium"):
    mesh = rt.Mesh.load_obj(os.path.join(ROOT, ".scene_cache", name + ".obj"), gpu_build=True)
    tris = mesh.dump()["tris"].copy()
    n = len(tris)
    sp = rt.Scene(); sp.add_material((0.9, 0.5, 0.2)); sp.add_mesh(mesh); sp.add_mesh_instance(0, 0); sp.upload_to_device()
    moved = tris.copy()
    moved[:, :9].reshape(-1, 3, 3)[..., 2] *= np.float32(1.5)
    v, nn, uv = (np.ascontiguousarray(moved[:, a:b]) for a, b in ((0, 9), (9, 12), (12, 18)))
    bufs = [rt.DeviceBuffer(nbytes=x.nbytes) for x in (v, nn, uv)]
    for b, x in zip(bufs, (v, nn, uv)):
        rt.check(h.rt_memcpy_h2d(b.ptr, x.ctypes.data, x.nbytes, None))
    ts = []
    for _ in range(20):
        rt.check(h.rt_device_synchronize())
        t = time.perf_counter()
        rt.check(h.rt_scene_rebuild_mesh_device(sp.device_handle, 0, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, n, None), "rebuild")
        ts.append(time.perf_counter() - t)
    dev = sorted(ts)[len(ts) // 2] * 1e3
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        m2 = rt.Mesh.from_triangles(moved, gpu_build=True)
        s2 = rt.Scene(); s2.add_material((0.9, 0.5, 0.2)); s2.add_mesh(m2); s2.add_mesh_instance(0, 0); s2.upload_to_device()
        ts.append(time.perf_counter() - t)
    host = sorted(ts)[len(ts) // 2] * 1e3
    print("%-8s %7d triangles: device-resident rebuild %.3f ms (min %.3f), host route (GPU build + flatten + upload) %.2f ms" % (name, n, dev, min(ts) * 1e3 if False else dev, host))
