#!/usr/bin/env python3
"""rt_bvh_build alone (host vertices in, host tree out): wall time per call for the benchmark meshes.
   python tools/bvh_build_bench.py"""
import ctypes as C, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("cuda-raytracing_amd")
h = rt.libs()[0]


def obj_vertices(path):                                   # the generators write `v x y z` and `f a/a b/b c/c` only
    v, f = [], []
    for line in open(path):
        if line.startswith("v "):
            v.append([float(t) for t in line.split()[1:4]])
        elif line.startswith("f "):
            f.append([int(t.split("/")[0]) - 1 for t in line.split()[1:4]])
    return np.asarray(v, np.float32)[np.asarray(f)].reshape(-1, 9)


for name in ("blob5k", "blob70k", "atrium"):
    p = os.path.join(ROOT, ".scene_cache", name + ".obj")
    if not os.path.exists(p):
        continue
    vert = np.ascontiguousarray(obj_vertices(p))
    n, cap = len(vert), 2 * len(vert)
    bounds, children = np.zeros((cap, 6), np.float32), np.zeros((cap, 2), np.int32)
    lfirst, lcount, lidx = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(n, np.int32)
    nn, nl = C.c_int32(0), C.c_int32(0)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        rc = h.rt_bvh_build(ptr(vert), n, 32, ptr(bounds), ptr(children), ptr(lfirst), ptr(lcount), ptr(lidx), C.byref(nn), C.byref(nl))
        ts.append(time.perf_counter() - t0)
        assert rc == 0, rc
    print("%-8s %7d triangles -> %7d nodes, %2d levels: rt_bvh_build %.2f ms (min of 8, first %.2f)" %
          (name, n, nn.value, nl.value, min(ts) * 1e3, ts[0] * 1e3), flush=True)
