"""ctypes bindings for the CPU oracle (oracle/librt_oracle.so) and, where it was built, the
reference-probe library (oracle/_ref/libref_probe.so).  TEST INFRASTRUCTURE ONLY: the product
package never imports this module."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int32)
_u8 = C.POINTER(C.c_uint8)


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return None if a is None else a.ctypes.data_as(_i)


def build_oracle():
    """Compile the C restatement (and _ref when /root/reference is mounted)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


class _Lib:
    """Common surface of librt_oracle.so (prefix orc_) and libref_probe.so (prefix ref_)."""

    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        L, p = self.lib, prefix
        g = lambda n: getattr(L, p + n)
        g("q_rsqrt").restype = C.c_float
        g("q_rsqrt").argtypes = [C.c_float]
        g("magnitude").restype = C.c_float
        g("magnitude").argtypes = [_f]
        g("aabb_ray_intersects").restype = C.c_float
        g("aabb_ray_intersects").argtypes = [_f, _f, _f, _f]
        for n in ("mesh_from_triangles", "obj_load"):
            g(n).restype = C.c_void_p
        g("mesh_from_triangles").argtypes = [_f, C.c_int]
        g("obj_load").argtypes = [C.c_char_p]
        for n in ("mesh_num_triangles", "mesh_num_nodes"):
            g(n).restype = C.c_int
            g(n).argtypes = [C.c_void_p]
        g("mesh_get_triangles").argtypes = [C.c_void_p, _f]
        g("mesh_get_nodes").restype = C.c_int
        g("mesh_get_nodes").argtypes = [C.c_void_p, _f, _i, _i]
        g("mesh_get_leaf_indices").argtypes = [C.c_void_p, _i]

    def _v(self, name, *ins, n_out):
        out = np.zeros(n_out, np.float32)
        args = [_fp(np.ascontiguousarray(a, np.float32)) for a in ins]
        fn = getattr(self.lib, self.prefix + name)
        fn.restype = None
        fn(*args, _fp(out))
        return out

    def q_rsqrt(self, x):
        return np.float32(getattr(self.lib, self.prefix + "q_rsqrt")(C.c_float(x)))

    def normalize(self, v): return self._v("normalize", v, n_out=3)
    def euler2quat(self, e): return self._v("euler2quat", e, n_out=4)
    def apply_quat(self, q, v): return self._v("apply_quat", q, v, n_out=3)
    def apply_euler(self, e, v): return self._v("apply_euler", e, v, n_out=3)
    def invert_lre(self, l): return self._v("invert_lre", l, n_out=6)
    def apply_lre(self, l, v): return self._v("apply_lre", l, v, n_out=3)
    def lre2homo(self, l): return self._v("lre2homo", l, n_out=16)
    def invert_intrinsic(self, K): return self._v("invert_intrinsic", K, n_out=9)
    def apply_matrix33(self, K, v): return self._v("apply_matrix33", K, v, n_out=3)
    def ray_ctor(self, o, d): return self._v("ray_ctor", o, d, n_out=13)
    def tri_test(self, tri18, o, d): return self._v("tri_test", tri18, o, d, n_out=5)
    def tri_from_vertices(self, abc9): return self._v("tri_from_vertices", abc9, n_out=18)
    def tri_center(self, tri18): return self._v("tri_center", tri18, n_out=3)

    def magnitude(self, v):
        return np.float32(getattr(self.lib, self.prefix + "magnitude")(_fp(np.ascontiguousarray(v, np.float32))))

    def aabb(self, bmin, bmax, o, d):
        a = [np.ascontiguousarray(x, np.float32) for x in (bmin, bmax, o, d)]
        return np.float32(getattr(self.lib, self.prefix + "aabb_ray_intersects")(*[_fp(x) for x in a]))

    # meshes -----------------------------------------------------------------
    def mesh_from_triangles(self, tris18):
        t = np.ascontiguousarray(tris18, np.float32).reshape(-1, 18)
        return getattr(self.lib, self.prefix + "mesh_from_triangles")(_fp(t), t.shape[0])

    def mesh_refit(self, h, tris18):
        t = np.ascontiguousarray(tris18, np.float32).reshape(-1, 18)
        fn = getattr(self.lib, self.prefix + "mesh_refit")
        fn.argtypes = [C.c_void_p, _f, C.c_int]
        assert fn(h, _fp(t), t.shape[0]) == 0

    def obj_load(self, path):
        h = getattr(self.lib, self.prefix + "obj_load")(path.encode())
        return h

    def mesh_dump(self, h):
        """-> dict(tris[n,18], boxes[m,6], child[m,2], leaf_count[m], leaf_idx[k])"""
        g = lambda n: getattr(self.lib, self.prefix + n)
        nt, nn = g("mesh_num_triangles")(h), g("mesh_num_nodes")(h)
        tris = np.zeros((nt, 18), np.float32)
        g("mesh_get_triangles")(h, _fp(tris))
        boxes = np.zeros((nn, 6), np.float32)
        child = np.zeros((nn, 2), np.int32)
        lc = np.zeros(nn, np.int32)
        total = g("mesh_get_nodes")(h, _fp(boxes), _ip(child), _ip(lc))
        li = np.zeros(max(total, 1), np.int32)
        g("mesh_get_leaf_indices")(h, _ip(li))
        return dict(tris=tris, boxes=boxes, child=child, leaf_count=lc, leaf_idx=li[:total])


class Oracle(_Lib):
    def __init__(self):
        path = os.path.join(ORACLE_DIR, "librt_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        super().__init__(path, "orc_")
        L = self.lib
        L.orc_mesh_single_triangle.restype = C.c_void_p
        L.orc_mesh_single_triangle.argtypes = [_f]
        L.orc_mesh_free.argtypes = [C.c_void_p]
        L.orc_mesh_stats.argtypes = [C.c_void_p, _i]
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_scene_add_material.restype = C.c_int
        L.orc_scene_add_material.argtypes = [C.c_void_p, _f, C.c_void_p, C.c_int, C.c_int, C.c_size_t]
        L.orc_scene_add_mesh.restype = C.c_int
        L.orc_scene_add_mesh.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_scene_add_instance.restype = C.c_int
        L.orc_scene_add_instance.argtypes = [C.c_void_p, C.c_int, C.c_int, _f, _f]
        L.orc_scene_update_instance.restype = C.c_int
        L.orc_scene_update_instance.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _f, _f]
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_int, _f, _f, _f, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_scene_set_material_params.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float]
        L.orc_render_ex.restype = C.c_int
        L.orc_render_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, _f, _f, _f, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                    C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_xorwow.argtypes = [C.c_ulonglong, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_camera_ray.argtypes = [C.c_int, C.c_int, _f, _f, _f, C.c_int, C.c_int, _f]
        L.orc_fnv1a64_image.restype = C.c_uint64
        L.orc_fnv1a64_image.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int]

    def mesh_single_triangle(self, abc9):
        return self.lib.orc_mesh_single_triangle(_fp(np.ascontiguousarray(abc9, np.float32)))

    def mesh_stats(self, h):
        o = np.zeros(5, np.int32)
        self.lib.orc_mesh_stats(h, _ip(o))
        return dict(nodes=int(o[0]), max_tris=int(o[1]), min_tris=int(o[2]), max_depth=int(o[3]), leaves=int(o[4]))

    def camera_ray(self, width, height, K, D, pose, x, y):
        o = np.zeros(3, np.float32)
        self.lib.orc_camera_ray(width, height, _fp(np.asarray(K, np.float32)), _fp(np.asarray(D, np.float32)),
                                _fp(np.asarray(pose, np.float32)), x, y, _fp(o))
        return o


class OracleScene:
    """Scene + render on the oracle; mirrors the reference's Scene/Camera call sequence."""

    def __init__(self, orc):
        self.o = orc
        self.h = orc.lib.orc_scene_create()
        self._keep = []

    def add_material(self, albedo, texture=None, roughness=0.0, metallic=0.0, illumination=0.0):
        a = np.ascontiguousarray(albedo, np.float32)
        if texture is None:
            idx = self.o.lib.orc_scene_add_material(self.h, _fp(a), None, 0, 0, 0)
        else:
            t = np.ascontiguousarray(texture, np.uint8)
            self._keep.append(t)
            idx = self.o.lib.orc_scene_add_material(self.h, _fp(a), t.ctypes.data, t.shape[1], t.shape[0], t.strides[0])
        self.o.lib.orc_scene_set_material_params(self.h, idx, roughness, metallic, illumination)
        return idx

    def add_mesh(self, mesh_handle):
        return self.o.lib.orc_scene_add_mesh(self.h, mesh_handle)

    def add_instance(self, mesh, material, pose=(0, 0, 0, 0, 0, 0), scale=(1, 1, 1)):
        return self.o.lib.orc_scene_add_instance(self.h, mesh, material, _fp(np.asarray(pose, np.float32)),
                                                 _fp(np.asarray(scale, np.float32)))

    def update_instance(self, index, mesh, material, pose, scale=(1, 1, 1)):
        return self.o.lib.orc_scene_update_instance(self.h, index, mesh, material, _fp(np.asarray(pose, np.float32)),
                                                    _fp(np.asarray(scale, np.float32)))

    def render(self, width, height, K, D, cam_pose, y0=0, y1=None, planes=True, threads=1):
        """-> dict(img[h,w,3] u8, hit_inst, hit_tri, pops, aabb, tris, inside (int32 [h,w]), stats)"""
        y1 = height if y1 is None else y1
        img = np.zeros((height, width, 3), np.uint8)
        names = ("hit_inst", "hit_tri", "pops", "aabb", "tris", "inside")
        pl = {n: (np.full((height, width), -1, np.int32) if planes else None) for n in names}
        Kf, Df, Pf = (np.ascontiguousarray(v, np.float32) for v in (K, D, cam_pose))

        def run(a, b):
            st = np.zeros(8, np.int64)
            rc = self.o.lib.orc_render(self.h, width, height, _fp(Kf), _fp(Df), _fp(Pf), img.ctypes.data, width * 3, a, b,
                                       *[(pl[n].ctypes.data if planes else None) for n in names], st.ctypes.data)
            assert rc == 0
            return st

        if threads <= 1:
            stats = run(y0, y1)
        else:
            from concurrent.futures import ThreadPoolExecutor
            step = 4
            bands = [(a, min(a + step, y1)) for a in range(y0, y1, step)]
            with ThreadPoolExecutor(threads) as ex:
                parts = list(ex.map(lambda ab: run(*ab), bands))
            stats = np.sum(parts, axis=0)
            stats[6] = max(p[6] for p in parts)
        out = dict(img=img, stats=dict(rays=int(stats[0]), pops=int(stats[1]), aabb=int(stats[2]), tris=int(stats[3]),
                                       inside=int(stats[4]), hits=int(stats[5]), max_stack=int(stats[6])))
        out.update(pl)
        return out

    def render_ex(self, width, height, K, D, cam_pose, spp=1, bounces=0, lighting=0, threads=1, y0=0, y1=None):
        """Extension semantics (oracle/rt_oracle.c orc_render_ex) -> dict(img, total_pops, stats); rows [y0, y1) only
        (the rest of img / total_pops stays zero)"""
        y1 = height if y1 is None else y1
        img = np.zeros((height, width, 3), np.uint8)
        pops = np.zeros((height, width), np.int32)
        Kf, Df, Pf = (np.ascontiguousarray(v, np.float32) for v in (K, D, cam_pose))

        def run(a, b):
            st = np.zeros(4, np.int64)
            rc = self.o.lib.orc_render_ex(self.h, width, height, _fp(Kf), _fp(Df), _fp(Pf), spp, bounces, lighting,
                                          img.ctypes.data, width * 3, a, b, pops.ctypes.data, st.ctypes.data)
            assert rc == 0
            return st
        if threads <= 1:
            st = run(y0, y1)
        else:
            from concurrent.futures import ThreadPoolExecutor
            step = 4 if (y1 - y0) >= 4 * threads else 1
            bands = [(a, min(a + step, y1)) for a in range(y0, y1, step)]
            with ThreadPoolExecutor(threads) as ex:
                st = np.sum(list(ex.map(lambda ab: run(*ab), bands)), axis=0)
        return dict(img=img, total_pops=pops, stats=dict(rays=int(st[0]), pops=int(st[1]), hits=int(st[2])))

    def close(self):
        if self.h:
            self.o.lib.orc_scene_free(self.h)
            self.h = None


def fnv1a64(img):
    """FNV-1a-64 over the tight RGB bytes."""
    lib = oracle().lib
    a = np.ascontiguousarray(img, np.uint8)
    return "%016x" % lib.orc_fnv1a64_image(a.ctypes.data, a.shape[1] * 3, a.shape[1], a.shape[0])


_ORACLE = None
_REF = None


def oracle():
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = Oracle()
    return _ORACLE


def ref_probe():
    """The reference-probe library, or None when it was not built (no /root/reference)."""
    global _REF
    path = os.path.join(ORACLE_DIR, "_ref", "libref_probe.so")
    if _REF is None and os.path.exists(path):
        _REF = _Lib(path, "ref_")
        _REF.lib.ref_instance_build.argtypes = [C.c_int, C.c_int, _f, _f, C.c_void_p]
    return _REF
