#!/usr/bin/env python3
"""Scene::refit_mesh (host vertices in, refitted device scene out): wall time per call including the final synchronise, for
the benchmark meshes, next to a rebuild (rt_bvh_build + upload) of the same mesh.   python tools/refit_bench.py"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("cuda-raytracing_amd")
h = rt.libs()[0]


def obj_triangles(path):                                  # the generators write `v x y z` and `f a/a b/b c/c` only
    v, f = [], []
    for line in open(path):
        if line.startswith("v "):
            v.append([float(t) for t in line.split()[1:4]])
        elif line.startswith("f "):
            f.append([int(t.split("/")[0]) - 1 for t in line.split()[1:4]])
    tri = np.asarray(v, np.float32)[np.asarray(f)]        # [n][3][3]
    uv = np.zeros((len(tri), 3, 2), np.float32)
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).astype(np.float32)
    return np.concatenate([tri.reshape(-1, 9), n, uv.reshape(-1, 6)], 1).astype(np.float32)   # 18 floats: vertices, normal, uvs


for name in ("blob5k", "blob70k", "atrium"):
    p = os.path.join(ROOT, ".scene_cache", name + ".obj")
    if not os.path.exists(p):
        continue
    tris = obj_triangles(p)
    t0 = time.perf_counter()
    mesh = rt.Mesh.from_triangles(tris, gpu_build=True)
    t_mesh = time.perf_counter() - t0
    sc = rt.Scene()
    sc.add_material((0.8, 0.8, 0.8))
    sc.add_mesh(mesh)
    sc.add_mesh_instance(0, 0)
    sc.upload_to_device()
    rt.check(h.rt_device_synchronize())
    t_build = time.perf_counter() - t0
    moved = tris.copy()
    moved[:, :9] *= np.float32(1.01)
    ts = []
    for k in range(10):
        t0 = time.perf_counter()
        sc.refit_mesh(0, moved if k % 2 == 0 else tris)
        rt.check(h.rt_device_synchronize())
        ts.append(time.perf_counter() - t0)
    # the C-ABI call alone (flat host arrays in, device scene refitted): what Scene::refit_mesh adds is host work -- the copy of
    # the triangle vector, the refit of the host tree, the flattening
    import ctypes as C
    va = [np.ascontiguousarray(a[:, :9]) for a in (moved, tris)]
    na = [np.ascontiguousarray(a[:, 9:12]) for a in (moved, tris)]
    tc = []
    for k in range(10):
        t0 = time.perf_counter()
        rt.check(h.rt_scene_refit_mesh(sc.device_handle, 0, va[k % 2].ctypes.data_as(C.POINTER(C.c_float)), na[k % 2].ctypes.data_as(C.POINTER(C.c_float)), len(tris), None))
        rt.check(h.rt_device_synchronize())
        tc.append(time.perf_counter() - t0)
    d_v, d_n = rt.DeviceBuffer(nbytes=va[0].nbytes), rt.DeviceBuffer(nbytes=na[0].nbytes)
    rt.check(h.rt_memcpy_h2d(d_v.ptr, va[0].ctypes.data, va[0].nbytes, None))
    rt.check(h.rt_memcpy_h2d(d_n.ptr, na[0].ctypes.data, na[0].nbytes, None))
    td = []
    for k in range(10):
        t0 = time.perf_counter()
        rt.check(h.rt_scene_refit_mesh_device(sc.device_handle, 0, d_v.ptr, d_n.ptr, len(tris), None))
        rt.check(h.rt_device_synchronize())
        td.append(time.perf_counter() - t0)
    print("%-8s rt_scene_refit_mesh_device (arrays already on the GPU) %.3f ms" % (name, min(td) * 1e3))
    print("%-8s %7d triangles: Scene::refit_mesh %.3f ms, rt_scene_refit_mesh alone %.3f ms (min of 10 each, synchronised); "
          "mesh from triangles (GPU build) %.1f ms + scene upload %.1f ms" % (name, len(tris), min(ts) * 1e3, min(tc) * 1e3, t_mesh * 1e3, (t_build - t_mesh) * 1e3), flush=True)
