#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz|json from the REFERENCE's own code (oracle/_ref/libref_probe.so, built
by oracle/Makefile from /root/reference/CudaRaytracer/*.hpp|.cpp where they lie) plus the frame hashes
SURVEY.md records from the reference's render().  Run in the build container only:

    make -C oracle && python tests/golden/make_golden.py

Fixtures are data (inputs + expected outputs); no reference source text is stored."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import importlib

import orc

scenes = importlib.import_module("cuda-raytracing_amd.scenes")


def main():
    r = orc.ref_probe()
    assert r is not None, "oracle/_ref/libref_probe.so missing: run `make -C oracle` where /root/reference is mounted"
    rng = np.random.default_rng(20241008)
    g = {}
    # ---- L0 math ------------------------------------------------------------------------------
    x = np.concatenate([np.array([1, 2, 3, 0.5, 14, 1e-3, 12345.678, 0.0, 1e-30, 1e30], np.float32),
                        np.exp(rng.uniform(-40, 40, 500)).astype(np.float32)])
    g["rsqrt_in"] = x
    g["rsqrt_out"] = np.array([r.q_rsqrt(float(v)) for v in x], np.float32)
    v = rng.normal(0, 3, (300, 3)).astype(np.float32)
    g["vec_in"] = v
    g["normalize_out"] = np.stack([r.normalize(a) for a in v])
    g["magnitude_out"] = np.array([r.magnitude(a) for a in v], np.float32)
    poses = np.concatenate([np.array([[1, 2, 3, 0.3, -0.2, 0.1], [-1, -4, 2, 0, 0, 0], [0, 10, 0, 0, 0.5, 0]], np.float32),
                            np.concatenate([rng.uniform(-5, 5, (200, 3)), rng.uniform(-3.2, 3.2, (200, 3))], 1).astype(np.float32)])
    g["pose_in"] = poses
    g["invert_lre_out"] = np.stack([r.invert_lre(p) for p in poses])
    g["lre2homo_out"] = np.stack([r.lre2homo(p) for p in poses])
    g["euler2quat_out"] = np.stack([r.euler2quat(p[3:]) for p in poses])
    pv = rng.normal(0, 2, (poses.shape[0], 3)).astype(np.float32)
    g["pose_vec_in"] = pv
    g["apply_lre_out"] = np.stack([r.apply_lre(p, a) for p, a in zip(poses, pv)])
    g["apply_euler_out"] = np.stack([r.apply_euler(p[3:], a) for p, a in zip(poses, pv)])
    Ks = np.array([scenes.K_1080, scenes.C1["K"], scenes.scaled_K(3840)], np.float32)
    g["K_in"] = Ks
    g["invert_intrinsic_out"] = np.stack([r.invert_intrinsic(k) for k in Ks])
    # apply_matrix(float3x3, float3) (utils.hpp:134-140): the first operation of ray generation, K_inv x (x, y, 1) (raycast.cu:161),
    # on pixel vectors of the three intrinsics above and on random matrices
    mats = np.concatenate([g["invert_intrinsic_out"].reshape(-1, 9)[[0] * 40 + [1] * 40 + [2] * 40], rng.normal(0, 2, (80, 9)).astype(np.float32)])
    px = np.concatenate([np.stack([rng.integers(0, 3840, 120), rng.integers(0, 2160, 120), np.ones(120)], 1), rng.normal(0, 50, (80, 3))]).astype(np.float32)
    g["matrix33_in"] = mats
    g["matrix33_vec_in"] = px
    g["apply_matrix33_out"] = np.stack([r.apply_matrix33(m, a) for m, a in zip(mats, px)])
    # Ray::Ray (Ray.hpp:17-23): direction_inv incl. zero / negative-zero / denormal / huge components
    ro_ = rng.uniform(-5, 5, (120, 3)).astype(np.float32)
    rd_ = rng.normal(0, 1, (120, 3)).astype(np.float32)
    rd_[0] = [0.0, 1.0, -0.0]; rd_[1] = [1e-45, -1e-40, 3e38]; rd_[2] = [-0.0, -0.0, 1.0]; rd_[3::11, 1] = 0.0
    g["ray_in"] = np.concatenate([ro_, rd_], 1)
    with np.errstate(all="ignore"):
        g["ray_ctor_out"] = np.stack([r.ray_ctor(a, b) for a, b in zip(ro_, rd_)])
    # MeshInstance::build_inv: the whole 104-byte struct
    import ctypes as C
    scl = rng.uniform(0.3, 2.0, (poses.shape[0], 3)).astype(np.float32)
    inst = np.zeros((poses.shape[0], 26), np.float32)
    for i, (p, s) in enumerate(zip(poses, scl)):
        buf = np.zeros(26, np.float32)
        r.lib.ref_instance_build(0, 0, p.ctypes.data_as(C.POINTER(C.c_float)), s.ctypes.data_as(C.POINTER(C.c_float)), buf.ctypes.data)
        inst[i] = buf
    g["instance_scale_in"] = scl
    g["instance_build_out"] = inst[:, 2:]           # drop mesh_index / material_index words
    # ---- AABB slab + triangle tests ------------------------------------------------------------
    n = 400
    bmin = rng.uniform(-2, 1, (n, 3)).astype(np.float32)
    bmax = (bmin + rng.uniform(0, 2, (n, 3))).astype(np.float32)
    ro = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    rd = rng.normal(0, 1, (n, 3)).astype(np.float32)
    rd[::17, 0] = 0.0                                # axis-parallel rays -> inf / NaN slabs
    ro[::34, 0] = bmin[::34, 0]
    g["aabb_in"] = np.concatenate([bmin, bmax, ro, rd], 1)
    g["aabb_out"] = np.array([r.aabb(a, b, o, d) for a, b, o, d in zip(bmin, bmax, ro, rd)], np.float32)
    abc = rng.uniform(-1, 1, (n, 9)).astype(np.float32)
    tris = np.stack([r.tri_from_vertices(a) for a in abc])
    tris[:, 12:18] = rng.uniform(0, 1, (n, 6)).astype(np.float32)
    to = (rng.uniform(-1, 1, (n, 3)) + np.array([0, -3, 0])).astype(np.float32)
    td = (tris[:, 0:9].reshape(n, 3, 3).mean(1) - to + rng.normal(0, 0.3, (n, 3))).astype(np.float32)
    g["tri_abc_in"] = abc
    g["tri_from_vertices_out"] = tris.copy()
    g["tri_ray_in"] = np.concatenate([to, td], 1)
    g["tri_test_out"] = np.stack([r.tri_test(t, o, d) for t, o, d in zip(tris, to, td)])
    g["tri_center_out"] = np.stack([r.tri_center(t) for t in tris])
    np.savez_compressed(os.path.join(HERE, "l0_math.npz"), **g)

    # ---- BVH topology + OBJ loading -----------------------------------------------------------
    tmp = os.path.join(HERE, "_tmp_blob5k.obj")
    scenes.write_blob_obj(tmp, 50, 51)
    b = {}
    d = r.mesh_dump(r.obj_load(tmp))
    os.remove(tmp)
    b["blob5k_tris_sha256"] = hashlib.sha256(d["tris"].tobytes()).hexdigest()
    b["blob5k_boxes_sha256"] = hashlib.sha256(d["boxes"].tobytes()).hexdigest()
    b["blob5k_num_nodes"] = int(d["child"].shape[0])
    b["blob5k_num_tris"] = int(d["tris"].shape[0])
    np.savez_compressed(os.path.join(HERE, "blob5k_bvh.npz"), child=d["child"], leaf_count=d["leaf_count"], leaf_idx=d["leaf_idx"],
                        boxes_head=d["boxes"][:64], tris_head=d["tris"][:64])
    # small OBJ exercising v / v/vt / v/vt/vn tokens, polygons (fan), comments, blank lines, vn records
    small = os.path.join(HERE, "small_mixed.obj")
    d2 = r.mesh_dump(r.obj_load(small))
    np.savez_compressed(os.path.join(HERE, "small_mixed_mesh.npz"), **d2)
    # random soup with duplicate triangles (unsplittable leaves)
    base = np.stack([r.tri_from_vertices(a) for a in rng.uniform(-1, 1, (40, 9)).astype(np.float32)])
    soup = np.concatenate([np.repeat(base[:1], 35, 0), base[1:]])
    d3 = r.mesh_dump(r.mesh_from_triangles(soup))
    np.savez_compressed(os.path.join(HERE, "soup_mesh.npz"), tris_in=soup, **{k: v for k, v in d3.items() if k != "tris"})
    # ---- frame hashes recorded by SURVEY.md from the reference's own render() ---------------------
    b["frames"] = {
        "C1_256x256": {"fnv1a64": "6b05ef62c4ffefb7", "hit_pixels": 1870, "source": "SURVEY.md section 4"},
        "C2_far_1920x1080": {"fnv1a64": "1987bc58fc5f9ed0", "source": "SURVEY.md section 8(d)"},
        "C2_mid_1920x1080": {"fnv1a64": "a85de3d252fa5a4f", "source": "SURVEY.md section 8(d)"},
        "C2_near_1920x1080": {"fnv1a64": "c78e8858fb3f7613", "source": "SURVEY.md section 8(d)"},
        "blob70k_obj_sha256": "0a35c3e93b078875a4937333448d4ad32349704c30a37696ed6ab05cd8fa9419",
        "blob5k_obj_sha256": "3c32131d0bf0b714a05cbdaae981deb7b49c4b14ed571ff07b66aedaff24da5d",
        "blob70k_bvh": {"nodes": 130227, "leaves": 65114, "max_tris_per_leaf": 8, "print_stats_max_depth": 17},
    }
    b["rsqrt_kat_hex"] = {"inputs": [1, 2, 3, 0.5, 14, 1e-3, 12345.678],
                          "bits": ["3f7f910f", "3f34f95e", "3f13ac3c", "3fb4f95e", "3e88d049", "41fcae36", "3c13559a"]}
    json.dump(b, open(os.path.join(HERE, "pins.json"), "w"), indent=1)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
