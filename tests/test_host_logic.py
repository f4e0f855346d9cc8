"""Host side of the product (no GPU): the C++ host API (BVH builder, OBJ loader, pose algebra,
MeshInstance::build_inv, camera parameters) against the oracle and the golden vectors; the restated
atanf against libm; the C-ABI libraries load and export every declared symbol; argument validation."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same(a, b):
    return np.array_equal(bits(a), bits(b))


def _mesh_equal(d, e):
    for k in ("boxes", "tris"):
        assert same(d[k], e[k]), k
    for k in ("child", "leaf_count", "leaf_idx"):
        assert np.array_equal(d[k], e[k]), k


def test_libraries_export_every_declared_symbol(rt):
    h, s = rt.libs()
    for header, lib, listed in (("rt_hip.h", h, rt.RT_HIP_SYMBOLS), ("rt_host.h", s, rt.RT_HOST_SYMBOLS)):
        text = open(os.path.join(ROOT, "include", header)).read()
        declared = set(re.findall(r"\b(rth?_[a-z0-9_]+)\s*\(", text))
        assert declared == set(listed), (declared ^ set(listed))
        for name in declared:
            assert getattr(lib, name) is not None
    assert h.rt_abi_version() == 2


def test_struct_layouts_match_header(rt):
    assert C.sizeof(rt.RtCameraParams) == 4 * (2 + 9 + 4 + 6 + 6)
    assert C.sizeof(rt.RtDebugPlanes) == 6 * 8


def test_argument_validation_without_gpu(rt):
    h, _ = rt.libs()
    n = C.c_int32(-1)
    assert h.rt_stripe_rows(1080, 16, 0, 8, C.byref(n)) == 0 and n.value == 144
    assert h.rt_stripe_rows(1080, 16, 7, 8, C.byref(n)) == 0 and n.value == 128
    assert h.rt_stripe_rows(1080, 0, 0, 8, C.byref(n)) == -1
    assert h.rt_stripe_rows(1080, 16, 8, 8, C.byref(n)) == -1
    assert h.rt_render(None, None, None, 0, None, 0) == -1
    assert h.rt_scene_upload(None, None) == -1
    assert h.rt_error_string(-1).decode().startswith("rt:")
    assert h.rt_scene_destroy(None) == 0
    cap = C.c_int32(7)
    assert h.rt_scene_mesh_capacity(None, 0, C.byref(cap)) == -1 and cap.value == 7    # (no scene: refused, the output untouched)
    assert h.rt_scene_rebuild_mesh_device(None, 0, None, None, None, 0, None) == -1
    assert h.rt_scene_view_stats(None, None, None, None, None) == -1


def test_comm_argument_validation_without_gpu(rt):
    """The exchange entry points reject bad arguments before they touch RCCL or a device (no GPU here)."""
    h, _ = rt.libs()
    v = C.c_int32(0)
    rc = h.rt_comm_available(C.byref(v))
    assert rc in (0, -5)                                        # RCCL present in this image; -5 = RT_E_COMM if it were not
    if rc == 0:
        assert v.value > 20000
        out = C.c_void_p()
        assert h.rt_comm_init_rank(None, 0, 1, C.byref(out)) == -1
        buf = (C.c_uint8 * 128)()
        assert h.rt_comm_init_rank(buf, 2, 2, C.byref(out)) == -1      # rank out of range
        assert h.rt_comm_init_all(None, 0, C.byref(out)) == -1
        assert h.rt_gather(None, None, 0, None, 0, None) == -1
        assert h.rt_render_tiled(None, None, None, None, None, 0, 16, 0, None, 0) == -1
    assert h.rt_comm_destroy(None) == 0
    assert h.rt_comm_info(None, None, None, None) == -1
    assert h.rt_error_string(-5).decode().startswith("rt:")
    assert h.rt_scene_refit_mesh(None, 0, None, None, 0, None) == -1


def test_stripe_rows_python_mirror(rt):
    tiling = __import__("importlib").import_module("cuda-raytracing_amd.tiling")
    h, _ = rt.libs()
    for H, stripe, world in [(1080, 16, 8), (203, 7, 3), (16, 16, 2), (5, 16, 4), (2160, 16, 8)]:
        tot = 0
        for r in range(world):
            n = C.c_int32(0)
            assert h.rt_stripe_rows(H, stripe, r, world, C.byref(n)) == 0
            assert n.value == tiling.stripe_rows(H, stripe, r, world) == len(tiling.frame_rows_of(H, stripe, r, world))
            tot += n.value
        assert tot == H


def test_rotating_gather_plan():
    """tiling.rotating_plan: every frame of a group has exactly one assembling rank, blocks are contiguous and as even as
    possible, and a group smaller than the world is padded so that no rank gets an empty message."""
    import importlib
    tiling = importlib.import_module("cuda-raytracing_amd.tiling")
    for world in (1, 2, 3, 4, 8):
        for count in (1, 2, 3, 7, 8, 20, 31, 32):
            slots, counts, offsets, real = tiling.rotating_plan(count, world)
            assert slots == max(count, world) and sum(counts) == slots and min(counts) >= 1
            assert max(counts) - min(counts) <= 1 and offsets == [sum(counts[:d]) for d in range(world)]
            assert sum(real) == count and all(0 <= r <= c for r, c in zip(real, counts))
            owners = [d for f in range(count) for d in range(world) if offsets[d] <= f < offsets[d] + real[d]]
            assert len(owners) == count and owners == sorted(owners)


def test_host_math_golden(rt):
    _, s = rt.libs()
    g = np.load(os.path.join(GOLDEN, "l0_math.npz"))
    f = C.POINTER(C.c_float)

    def call(fn, n_out, *ins):
        out = np.zeros(n_out, np.float32)
        fn(*[np.ascontiguousarray(a, np.float32).ctypes.data_as(f) for a in ins], out.ctypes.data_as(f))
        return out
    assert same([s.rth_q_rsqrt(float(v)) for v in g["rsqrt_in"]], g["rsqrt_out"])
    assert same(np.stack([call(s.rth_normalize, 3, v) for v in g["vec_in"]]), g["normalize_out"])
    P, V = g["pose_in"], g["pose_vec_in"]
    assert same(np.stack([call(s.rth_invert_lre, 6, p) for p in P]), g["invert_lre_out"])
    assert same(np.stack([call(s.rth_euler2quat, 4, p[3:]) for p in P]), g["euler2quat_out"])
    assert same(np.stack([call(s.rth_apply_lre, 3, p, v) for p, v in zip(P, V)]), g["apply_lre_out"])
    assert same(np.stack([call(s.rth_invert_intrinsic, 9, k) for k in g["K_in"]]), g["invert_intrinsic_out"])
    # MeshInstance::build_inv: pose, inv_pose, rotation, inv_rotation, scale, inv_scale
    got = np.stack([call(s.rth_instance_build, 24, p, sc) for p, sc in zip(P, g["instance_scale_in"])])
    assert same(got, g["instance_build_out"])


def test_atanf_restatement_matches_libm(rt):
    """raycast.cu:170 calls libm atan(float); the kernels evaluate rt::atanf_fdlibm instead.  Check it against
    this machine's libm over every float in [0, 8) (all radii a <= 150-degree fisheye can produce) in strides,
    plus dense windows around the argument-reduction breakpoints and the large/tiny ranges."""
    _, s = rt.libs()
    libm = C.CDLL("libm.so.6")
    libm.atanf.restype = C.c_float
    libm.atanf.argtypes = [C.c_float]
    hi = int(np.float32(8.0).view(np.uint32))
    idx = list(range(0, hi, 4099))
    for centre in (0x31000000, 0x3ee00000, 0x3f300000, 0x3f980000, 0x401c0000, 0x4c000000):
        idx += list(range(centre - 300, centre + 300))
    idx += list(range(0x7f7ffff0, 0x7f800001)) + [0, 1, 0x00800000]
    xs = np.array(idx, np.uint32).view(np.float32)
    mine = np.array([s.rth_atanf(float(x)) for x in xs], np.float32)
    ref = np.array([libm.atanf(float(x)) for x in xs], np.float32)
    assert same(mine, ref)
    assert same([s.rth_atanf(-float(x)) for x in xs[::50]], [libm.atanf(-float(x)) for x in xs[::50]])


def test_plane_epsilon_float_threshold():
    """TrianglePrimitive.hpp:66 compares abs(denom) < 1e-6 in double; the kernel compares in float against
    0x358637be.  Both select exactly the same floats."""
    thr = np.array([0x358637be], np.uint32).view(np.float32)[0]
    u = np.arange(0x358637be - 5000, 0x358637be + 5000, dtype=np.uint32)
    x = u.view(np.float32)
    assert np.array_equal(x.astype(np.float64) < 1e-6, x < thr)
    wide = np.linspace(0, 0x7f800000, 200001).astype(np.uint32).view(np.float32)
    assert np.array_equal(wide.astype(np.float64) < 1e-6, wide < thr)
    src = open(os.path.join(ROOT, "cuda-raytracing_amd", "csrc", "rt_kernels.hip")).read()
    assert "0x358637be" in src


def test_xorwow_matches_oracle(rt, oracle):
    """The extension's generator: product (rt_math.h) == oracle, and the seeding constants of curand_init(seed, 0, 0)."""
    _, s = rt.libs()
    for seed in (0, 1, 1000, 123456789, (1 << 63) + 12345, 0xFFFFFFFFFFFFF830):       # incl. a sign-extended negative seed
        a_bits, a_u = np.zeros(64, np.uint32), np.zeros(64, np.float32)
        b_bits, b_u = np.zeros(64, np.uint32), np.zeros(64, np.float32)
        s.rth_xorwow(seed, 64, a_bits.ctypes.data, a_u.ctypes.data)
        oracle.lib.orc_xorwow(seed, 64, b_bits.ctypes.data, b_u.ctypes.data)
        assert np.array_equal(a_bits, b_bits) and same(a_u, b_u)
        assert (a_u > 0).all() and (a_u <= 1).all()
    # seed 0: state = the five Marsaglia constants mixed with t0 = 1099087573 * 0xaad26b49, t1 = 2591861531 * 0xf7dcefdd
    t0 = (1099087573 * 0xaad26b49) & 0xFFFFFFFF
    t1 = (2591861531 * 0xf7dcefdd) & 0xFFFFFFFF
    v = [(123456789 + t0) & 0xFFFFFFFF, 362436069 ^ t0, (521288629 + t1) & 0xFFFFFFFF, 88675123 ^ t1, (5783321 + t0) & 0xFFFFFFFF]
    d = (6615241 + t1 + t0) & 0xFFFFFFFF
    t = v[0] ^ (v[0] >> 2)
    v4 = ((v[4] ^ ((v[4] << 4) & 0xFFFFFFFF)) ^ (t ^ ((t << 1) & 0xFFFFFFFF))) & 0xFFFFFFFF
    first = (v4 + d + 362437) & 0xFFFFFFFF
    got = np.zeros(1, np.uint32)
    s.rth_xorwow(0, 1, got.ctypes.data, None)
    assert int(got[0]) == first


def test_host_bvh_and_obj_vs_oracle(rt, oracle, blob5k):
    _mesh_equal(rt.Mesh.load_obj(blob5k).dump(), oracle.mesh_dump(oracle.obj_load(blob5k)))
    p = os.path.join(GOLDEN, "small_mixed.obj")
    d = rt.Mesh.load_obj(p).dump()
    _mesh_equal(d, oracle.mesh_dump(oracle.obj_load(p)))
    _mesh_equal(d, np.load(os.path.join(GOLDEN, "small_mixed_mesh.npz")))
    e = np.load(os.path.join(GOLDEN, "soup_mesh.npz"))
    m = rt.Mesh.from_triangles(e["tris_in"]).dump()
    for k in ("child", "leaf_count", "leaf_idx"):
        assert np.array_equal(m[k], e[k])
    assert same(m["boxes"], e["boxes"])


def test_host_bvh_70k_vs_oracle(rt, oracle, blob70k):
    m = rt.Mesh.load_obj(blob70k)
    assert (m.num_triangles, m.num_nodes) == (69936, 130227)
    _mesh_equal(m.dump(), oracle.mesh_dump(oracle.obj_load(blob70k)))


def test_host_bvh_random_soups(rt, oracle):
    import scene_defs as sd
    for seed, n in [(1, 0), (2, 1), (3, 2), (4, 3), (5, 64), (6, 1500)]:
        tris = sd.random_triangles(n, seed=seed) if n else np.zeros((0, 18), np.float32)
        _mesh_equal(rt.Mesh.from_triangles(tris).dump(), oracle.mesh_dump(oracle.mesh_from_triangles(tris)))
    single = rt.Mesh.single_triangle([-1, 0, -1, 1, 0, -1, 0, 0, 1]).dump()
    assert same(single["tris"], oracle.mesh_dump(oracle.mesh_single_triangle([-1, 0, -1, 1, 0, -1, 0, 0, 1]))["tris"])


def _same_or_both_nan(a, b):
    a = np.ascontiguousarray(a, np.float32).ravel()
    b = np.ascontiguousarray(b, np.float32).ravel()
    return a.shape == b.shape and bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


def test_host_bvh_awkward_meshes_vs_oracle(rt, oracle):
    """The product's host builder against the oracle's (which test_oracle_pins.py holds against the reference's own BVHTree::fill on the
    same meshes): lattice vertices, zero-area triangles, piles of coincident triangles, 1e18 / 1e-20 coordinates, slivers, NaN and
    infinite vertices (NaN boxes compared as NaN) -- the mesh kinds of the adversarial GPU fuzz."""
    import test_gpu_parity as gp
    for seed in range(400):
        kind, tris = gp._adversarial_mesh(oracle, np.random.default_rng(52000 + seed))
        d, e = rt.Mesh.from_triangles(tris).dump(), oracle.mesh_dump(oracle.mesh_from_triangles(tris))
        assert _same_or_both_nan(d["boxes"], e["boxes"]) and _same_or_both_nan(d["tris"], e["tris"]), (seed, kind)
        for k in ("child", "leaf_count", "leaf_idx"):
            assert np.array_equal(d[k], e[k]), (seed, kind, k)


def test_host_math_on_special_values_vs_oracle(rt, oracle):
    """Pose algebra and MeshInstance::build_inv of the host library against the oracle on zeros of both signs, denormals, FLT_MAX,
    infinities, NaN, 1e-20 / 1e18 mixed with ordinary values (the oracle against the reference's own code on such values:
    test_oracle_pins.py::test_oracle_vs_reference_live_on_special_values)."""
    import test_oracle_pins as op
    _, s = rt.libs()
    f = C.POINTER(C.c_float)

    def call(fn, n_out, *ins):
        out = np.zeros(n_out, np.float32)
        fn(*[np.ascontiguousarray(a, np.float32).ctypes.data_as(f) for a in ins], out.ctypes.data_as(f))
        return out
    import orc as orc_mod
    ref = orc_mod.ref_probe()
    rng = np.random.default_rng(777)
    with np.errstate(all="ignore"):
        for it in range(4000):
            ps = (0.0, 0.15, 0.5, 0.5)[it % 4]
            v, p = op._draw(rng, 3, ps), np.concatenate([op._draw(rng, 3, ps), op._draw(rng, 3, ps, 3.0)])
            m9 = op._draw(rng, 9, ps)
            assert _same_or_both_nan(call(s.rth_normalize, 3, v), oracle.normalize(v)), ("normalize", v)
            assert _same_or_both_nan(call(s.rth_invert_lre, 6, p), oracle.invert_lre(p)), ("invert_lre", p)
            assert _same_or_both_nan(call(s.rth_euler2quat, 4, p[3:]), oracle.euler2quat(p[3:])), ("euler2quat", p)
            assert _same_or_both_nan(call(s.rth_apply_lre, 3, p, v), oracle.apply_lre(p, v)), ("apply_lre", p, v)
            assert _same_or_both_nan(call(s.rth_invert_intrinsic, 9, m9), oracle.invert_intrinsic(m9)), ("invert_intrinsic", m9)
            x = float(op._draw(rng, 1, ps)[0])
            assert _same_or_both_nan(s.rth_q_rsqrt(x), oracle.q_rsqrt(x)), ("q_rsqrt", x)
            # MeshInstance::build_inv (MeshInstance.hpp): against the reference's own constructor where oracle/_ref was built
            if ref is not None:
                sc = op._draw(rng, 3, ps, 1.0)
                buf = np.zeros(26, np.float32)
                ref.lib.ref_instance_build(0, 0, np.ascontiguousarray(p, np.float32).ctypes.data_as(f), sc.ctypes.data_as(f), buf.ctypes.data)
                assert _same_or_both_nan(call(s.rth_instance_build, 24, p, sc), buf[2:]), ("instance_build", p, sc)


def test_obj_loader_error_behaviour(rt, tmp_path):
    with pytest.raises(rt.RtError, match="Could not open file"):
        rt.Mesh.load_obj(str(tmp_path / "missing.obj"))
    p = tmp_path / "vn_only.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    with pytest.raises(rt.RtError):
        rt.Mesh.load_obj(str(p))
    q = tmp_path / "range.obj"
    q.write_text("v 0 0 0\nv 1 0 0\nf 1 2 3\n")
    with pytest.raises(rt.RtError, match="out of range"):
        rt.Mesh.load_obj(str(q))


def test_camera_params(rt, oracle, scenes):
    cam = rt.Camera(1920, 1080, scenes.K_1080, scenes.D_REF)
    pose = (-1.0, -4.0, 2.0, 0.3, -0.1, 0.2)
    cam.set_pose(pose)
    p = cam.params()
    assert (p.width, p.height) == (1920, 1080)
    assert same(list(p.K_inv), oracle.invert_intrinsic(scenes.K_1080))
    assert same(list(p.inv_camera_pose), oracle.invert_lre(pose))
    assert same(list(p.camera_pose), np.asarray(pose, np.float32))
    assert same(list(p.D), np.asarray(scenes.D_REF, np.float32))


def test_no_gpu_fails_loudly(rt, scenes):
    """Without a device the product refuses to render; it never falls back to a CPU path."""
    if rt.device_count() > 0:
        pytest.skip("GPU present")
    s = rt.Scene()
    s.add_material((1, 1, 1))
    s.add_mesh(rt.Mesh.single_triangle([-1, 0, -1, 1, 0, -1, 0, 0, 1]))
    s.add_mesh_instance(0, 0)
    with pytest.raises(rt.RtError):
        s.upload_to_device()


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ (only tests/, smoke() and bench.py's cpu_baseline may)."""
    pkg = os.path.join(ROOT, "cuda-raytracing_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                text = open(os.path.join(dp, f), errors="ignore").read()
                assert "rt_oracle" not in text and "librt_oracle" not in text and "import orc" not in text, f


def test_png_writer_roundtrip(rt, tmp_path):
    """write_png_bgr (display_image's out.png without OpenCV): decodes with PIL to the same pixels, B,G,R -> R,G,B."""
    from PIL import Image
    rng = np.random.default_rng(0)
    for w, h in [(1, 1), (7, 5), (301, 223)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        p = str(tmp_path / ("t_%d_%d.png" % (w, h)))
        rt.write_png(p, img)
        back = np.asarray(Image.open(p).convert("RGB"))
        assert back.shape == (h, w, 3) and np.array_equal(back, img[:, :, ::-1])
    _, s = rt.libs()
    assert s.rth_write_png_bgr(b"/nonexistent_dir/x.png", img.ctypes.data, w, h, w * 3) == -1


def test_obj_lenient_mode(rt, tmp_path):
    """OBJLoader::load_lenient: `v//vn` tokens and negative (relative) indices give the same mesh as the equivalent
    strict file; strict mode still rejects them like the reference (H11)."""
    strict = tmp_path / "strict.obj"
    strict.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nf 1/1 2/2 3/3\nv 0 1 0.5\nvt 0 1\nf 1/1 3/3 4/4\nf 1 2 4\n")
    loose = tmp_path / "loose.obj"
    loose.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nvn 0 0 1\nvt 0 0\nvt 1 0\nvt 1 1\nf -3/-3 -2/-2 -1/-1\nv 0 1 0.5\nvt 0 1\n"
                     "f 1/1/1 3/3/1 -1/-1/1\nf 1//1 2//1 4//1\n")
    a = rt.Mesh.load_obj(str(strict)).dump()
    b = rt.Mesh.load_obj(str(loose), lenient=True).dump()
    assert a["tris"].shape == (3, 18)
    _mesh_equal(a, b)
    with pytest.raises(rt.RtError):
        rt.Mesh.load_obj(str(loose))


def _desc_from_mesh(rt, dump):
    """RtMeshDesc (plus the arrays that back it) from a Mesh.dump()."""
    t = dump["tris"]
    keep = dict(v=np.ascontiguousarray(t[:, 0:9]), n=np.ascontiguousarray(t[:, 9:12]), uv=np.ascontiguousarray(t[:, 12:18]),
                b=np.ascontiguousarray(dump["boxes"]), c=np.ascontiguousarray(dump["child"]),
                lc=np.ascontiguousarray(dump["leaf_count"]), li=np.ascontiguousarray(dump["leaf_idx"]))
    first = np.zeros(len(keep["lc"]), np.int32)
    first[1:] = np.cumsum(keep["lc"])[:-1]
    keep["lf"] = first
    f, i = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    d = rt.RtMeshDesc(t.shape[0], keep["v"].ctypes.data_as(f), keep["n"].ctypes.data_as(f), keep["uv"].ctypes.data_as(f),
                      len(keep["lc"]), keep["b"].ctypes.data_as(f), keep["c"].ctypes.data_as(i), keep["lf"].ctypes.data_as(i),
                      keep["lc"].ctypes.data_as(i), len(keep["li"]), keep["li"].ctypes.data_as(i))
    return d, keep


def test_scene_upload_validates_its_input(rt, blob5k):
    """rt_scene_upload rejects malformed trees / indices with RT_E_INVALID before it touches the device (the traversal
    kernel trusts the uploaded layout, so a bad tree must never reach it)."""
    import scene_defs as sd
    h, _ = rt.libs()
    dump = rt.Mesh.from_triangles(sd.random_triangles(40, seed=9)).dump()
    mat = rt.RtMaterialDesc(0.0, (C.c_float * 3)(1, 1, 1), 0.0, 0.0, None, 0, 0, 0)
    inst = rt.RtInstanceDesc(0, 0, (C.c_float * 6)(), (C.c_float * 6)(), (C.c_float * 3)(), (C.c_float * 3)(),
                             (C.c_float * 3)(1, 1, 1), (C.c_float * 3)(1, 1, 1))

    def upload(mesh_desc, instance=inst, material=mat):
        sd_ = rt.RtSceneDesc(1, C.pointer(mesh_desc), 1, C.pointer(material), 1, C.pointer(instance))
        out = C.c_void_p()
        rc = h.rt_scene_upload(C.byref(sd_), C.byref(out))
        if rc == 0:
            h.rt_scene_destroy(out)
        return rc

    d, keep = _desc_from_mesh(rt, dump)
    ok = upload(d)
    assert ok != -1                                        # valid: succeeds on a GPU box, "no device"-type error here
    if rt.device_count() == 0:
        assert ok != 0

    def broken(mutate):
        d2, k2 = _desc_from_mesh(rt, dump)
        mutate(d2, k2)
        return upload(d2)
    interior = int(np.nonzero(keep["c"][:, 0] > 0)[0][0])
    leaf = int(np.nonzero(keep["c"][:, 0] < 0)[0][0])
    assert broken(lambda d2, k: k["c"].__setitem__((interior, 0), len(k["lc"]) + 5)) == -1      # child out of range
    assert broken(lambda d2, k: k["c"].__setitem__((interior, 1), interior)) == -1              # child not after parent (cycle)
    assert broken(lambda d2, k: k["c"].__setitem__((interior, 1), k["c"][interior, 0])) == -1   # both children the same node
    assert broken(lambda d2, k: k["lc"].__setitem__(leaf, 10 ** 6)) == -1                       # leaf range past the list
    assert broken(lambda d2, k: k["li"].__setitem__(0, 10 ** 6)) == -1                          # triangle index out of range
    assert broken(lambda d2, k: setattr(d2, "num_nodes", -1)) == -1
    assert broken(lambda d2, k: setattr(d2, "num_nodes", 0)) != -1                                 # 0 = "build the tree on the device": valid
    bad_inst = rt.RtInstanceDesc(3, 0, (C.c_float * 6)(), (C.c_float * 6)(), (C.c_float * 3)(), (C.c_float * 3)(),
                                 (C.c_float * 3)(1, 1, 1), (C.c_float * 3)(1, 1, 1))
    assert upload(d, instance=bad_inst) == -1                                                   # mesh_index out of range
    tex = np.zeros((4, 4, 3), np.uint8)
    bad_mat = rt.RtMaterialDesc(0.0, (C.c_float * 3)(1, 1, 1), 0.0, 0.0, tex.ctypes.data, 4, 4, 5)   # pitch < width * 3
    assert upload(d, material=bad_mat) == -1


# ---------------------------------------------------------------- image files, overlay, interaction

def test_image_decoders_match_committed_fixtures(rt):
    """PNG (all colour types, 1/2/4/8/16 bit, plain and Adam7-interlaced) and JPEG (baseline and progressive; 4:4:4, 4:2:2,
    4:2:0, grey, custom Huffman tables, restart intervals, 1x1) decode to exactly the bytes libpng / libjpeg-turbo produce
    (tests/golden/make_image_fixtures.py); a progressive file cut before its last scan is refused."""
    d = os.path.join(GOLDEN, "images")
    exp = np.load(os.path.join(d, "images_expected.npz"))
    assert len(exp.files) >= 29 and sum(n.startswith("jpg_prog") for n in exp.files) >= 7 and sum(n.startswith("png_adam7") for n in exp.files) >= 8
    for name in exp.files:
        got = rt.read_image(os.path.join(d, name))
        assert got.shape == exp[name].shape and np.array_equal(got, exp[name]), name
    with pytest.raises(rt.RtError, match="incomplete progressive"):
        rt.read_image(os.path.join(d, "refused_progressive_cut.jpg"))
    with pytest.raises(rt.RtError):
        rt.read_image(os.path.join(d, "images_expected.npz"))


def test_image_decoders_against_pillow_live(rt, tmp_path):
    """The same comparison on freshly written files of odd sizes, when Pillow is importable (it is in this image)."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(11)
    for k, (w, h) in enumerate([(23, 40), (96, 8), (8, 96), (129, 67)]):
        rgb = np.clip(rng.normal(128, 60, (h, w, 3)) * 0.3 + np.linspace(0, 255, w)[None, :, None] * 0.7, 0, 255).astype(np.uint8)
        for ss in (0, 1, 2):
            p = str(tmp_path / ("a%d_%d.jpg" % (k, ss)))
            Image.fromarray(rgb).save(p, quality=40 + 20 * ss, subsampling=ss)
            assert np.array_equal(rt.read_image(p), np.asarray(Image.open(p).convert("RGB"))[..., ::-1]), (w, h, ss)
            p = str(tmp_path / ("p%d_%d.jpg" % (k, ss)))
            Image.fromarray(rgb).save(p, quality=30 + 30 * ss, subsampling=ss, progressive=True, optimize=bool(k & 1))
            assert np.array_equal(rt.read_image(p), np.asarray(Image.open(p).convert("RGB"))[..., ::-1]), ("progressive", w, h, ss)
        p = str(tmp_path / ("a%d.png" % k))
        Image.fromarray(rgb).save(p, compress_level=k * 3)               # level 0 = stored blocks, 9 = dynamic Huffman
        assert np.array_equal(rt.read_image(p), rgb[..., ::-1])


def test_corrupted_image_files_never_crash(rt, tmp_path):
    """Byte flips, truncations and spliced garbage in every fixture: the decoders either decode something of the announced
    size or refuse with an RtError -- never a crash, a hang or an allocation beyond the caps."""
    d = os.path.join(GOLDEN, "images")
    rng = np.random.default_rng(77)
    names = [n for n in sorted(os.listdir(d)) if n.endswith((".png", ".jpg"))]
    tried = refused = 0
    for name in names:
        data = open(os.path.join(d, name), "rb").read()
        for k in range(12):
            b = bytearray(data)
            mode = k % 3
            if mode == 0:                                              # a few flipped bytes past the signature
                for _ in range(1 + k // 3):
                    b[int(rng.integers(2, len(b)))] ^= int(rng.integers(1, 256))
            elif mode == 1:                                            # truncated
                b = b[:int(rng.integers(4, len(b)))]
            else:                                                      # a run of bytes overwritten with 0xFF / 0x00 / noise
                i = int(rng.integers(2, len(b) - 1))
                n = int(rng.integers(1, 24))
                b[i:i + n] = bytes(rng.integers(0, 256, n, dtype=np.uint8)) if k % 2 else b"\xff" * n
            p = str(tmp_path / ("c_%d_%s" % (k, name)))
            open(p, "wb").write(bytes(b))
            tried += 1
            try:
                img = rt.read_image(p)
                assert img.ndim == 3 and img.shape[2] == 3 and img.size <= 3 << 28
            except rt.RtError:
                refused += 1
            os.remove(p)
    assert tried >= 300 and refused > tried // 4


def test_zlib_inflate_against_zlib(rt):
    import zlib
    host = rt.libs()[1]
    rng = np.random.default_rng(3)
    for data in [b"", b"a", b"abc" * 1000, rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(), bytes(range(256)) * 300,
                 rng.integers(0, 4, 50000, dtype=np.uint8).tobytes()]:
        for level in (0, 1, 6, 9):
            z = zlib.compress(data, level)
            src = np.frombuffer(z, np.uint8)
            out = np.zeros(max(len(data), 1), np.uint8)
            n = C.c_size_t(0)
            rt.check(host.rth_zlib_inflate(src.ctypes.data, len(z), out.ctypes.data, out.nbytes, C.byref(n)))
            assert n.value == len(data) and out[:n.value].tobytes() == data
    bad = bytearray(zlib.compress(b"hello world, hello world, hello world", 6))
    bad[-1] ^= 1                                                       # checksum
    n = C.c_size_t(0)
    assert host.rth_zlib_inflate(np.frombuffer(bytes(bad), np.uint8).ctypes.data, len(bad), None, 0, C.byref(n)) != 0


def test_hostile_image_files_are_refused(rt, tmp_path):
    """Structurally valid files built to mislead the decoders: a JPEG with a second frame header that changes the
    sampling factors (the planes would be sized for the first, read for the second), an SOS segment with no payload, and
    a PNG whose IDAT inflates to far more than IHDR announces (a decompression bomb).  All must be refused, not decoded."""
    import struct
    import zlib
    d = os.path.join(GOLDEN, "images")
    jpg = open(os.path.join(d, "jpg_420_q75.jpg"), "rb").read()
    i = jpg.index(b"\xff\xc0")
    seg_len = struct.unpack(">H", jpg[i + 2:i + 4])[0]
    sof = bytearray(jpg[i:i + 2 + seg_len])
    assert sof[9] == 3 and sof[11] == 0x22                              # three components, luma 2x2
    sof[11] = 0x11                                                      # second frame header says 4:4:4
    p = str(tmp_path / "double_sof.jpg")
    open(p, "wb").write(jpg[:i + 2 + seg_len] + bytes(sof) + jpg[i + 2 + seg_len:])
    with pytest.raises(rt.RtError, match="more than one frame header"):
        rt.read_image(p)
    j = jpg.index(b"\xff\xda")
    p = str(tmp_path / "empty_sos.jpg")
    open(p, "wb").write(jpg[:j] + b"\xff\xda\x00\x02" + jpg[j:])
    with pytest.raises(rt.RtError):
        rt.read_image(p)

    # a sequential file that codes its components twice: the second scan would be decoded on top of the first one's coefficients
    eoi = jpg.rindex(b"\xff\xd9")
    p = str(tmp_path / "scan_twice.jpg")
    open(p, "wb").write(jpg[:eoi] + jpg[j:eoi] + jpg[eoi:])
    with pytest.raises(rt.RtError, match="scanned twice"):
        rt.read_image(p)

    # DC differences that run the predictor out of the 16-bit coefficient range (32 blocks of +2047 each): corrupt data, not an
    # integer overflow.  The same file with alternating signs is a valid image and decodes.
    def grey_jpeg(diff_bits):
        seg = lambda m, body: b"\xff" + bytes([m]) + struct.pack(">H", len(body) + 2) + body
        bits = "".join("0" + b + "0" for b in diff_bits)            # DC code '0' = category 11, the 11 value bits, AC code '0' = end of block
        bits += "1" * (-len(bits) % 8)
        data = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)).replace(b"\xff", b"\xff\x00")
        return (b"\xff\xd8" + seg(0xDB, b"\x00" + bytes([1] * 64)) + seg(0xC0, struct.pack(">BHHB", 8, 8, 8 * len(diff_bits), 1) + b"\x01\x11\x00")
                + seg(0xC4, b"\x00" + bytes([1] + [0] * 15) + b"\x0b") + seg(0xC4, b"\x10" + bytes([1] + [0] * 15) + b"\x00")
                + seg(0xDA, b"\x01\x01\x00\x00\x3f\x00") + data + b"\xff\xd9")
    up, down = "1" * 11, "0" * 11                                       # +2047, -2047
    p = str(tmp_path / "dc_ok.jpg")
    open(p, "wb").write(grey_jpeg([up, down] * 16))
    assert rt.read_image(p).shape == (8, 256, 3)
    p = str(tmp_path / "dc_runaway.jpg")
    open(p, "wb").write(grey_jpeg([up] * 32))
    with pytest.raises(rt.RtError, match="corrupt JPEG data"):
        rt.read_image(p)

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))
    ihdr = struct.pack(">IIBBBBB", 4, 4, 8, 2, 0, 0, 0)                # 4 x 4 RGB: 52 bytes of image data
    for payload, ok in ((b"\0" * 52, True), (b"\0" * (64 << 20), False)):
        p = str(tmp_path / ("bomb_%d.png" % ok))
        open(p, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(payload, 9)) + chunk(b"IEND", b""))
        if ok:
            assert rt.read_image(p).shape == (4, 4, 3)
        else:
            assert os.path.getsize(p) < 100000
            with pytest.raises(rt.RtError):
                rt.read_image(p)
    # rth_zlib_inflate with a destination buffer stops at its capacity
    host = rt.libs()[1]
    z = np.frombuffer(zlib.compress(b"x" * 100000, 9), np.uint8)
    out = np.zeros(1000, np.uint8)
    n = C.c_size_t(0)
    assert host.rth_zlib_inflate(z.ctypes.data, z.size, out.ctypes.data, out.nbytes, C.byref(n)) != 0


def test_texture_from_png_and_jpeg(rt, tmp_path):
    """Material::upload_texture takes PNG / JPEG / PPM by signature (the reference: cv::imread, Material.hpp:29-43)."""
    d = os.path.join(GOLDEN, "images")
    sc = rt.Scene()
    for name in ("png_RGB.png", "jpg_420_q75.jpg"):
        sc.add_material((1, 1, 1), texture_path=os.path.join(d, name))
    for name in ("jpg_prog_420_q75.jpg", "png_adam7_RGB.png"):
        sc.add_material((1, 1, 1), texture_path=os.path.join(d, name))
    with pytest.raises(rt.RtError):
        sc.add_material((1, 1, 1), texture_path=os.path.join(d, "refused_progressive_cut.jpg"))


def test_overlay_text(rt):
    host = rt.libs()[1]
    img = np.zeros((40, 200, 3), np.uint8)
    host.rth_overlay_text_bgr(img.ctypes.data, 200, 40, img.strides[0], b"FPS: 12.5", 10, 30, 3, 0, 255, 0)
    on = (img == (0, 255, 0)).all(-1)
    assert on.any() and not img[..., 0].any() and not img[..., 2].any()
    ys, xs = np.nonzero(on)
    assert ys.min() == 30 - 21 and ys.max() == 29 and xs.min() == 10 and xs.max() < 10 + 9 * 18   # 7 rows x scale 3 above y = 30
    cell = on[9:30, 10:25]                                              # 'F': full top row, full left column
    assert cell[0].all() and cell[:, 0].all() and not cell[20, 6:].any()
    clipped = np.zeros((10, 10, 3), np.uint8)
    host.rth_overlay_text_bgr(clipped.ctypes.data, 10, 10, clipped.strides[0], b"W?", -4, 5, 2, 9, 9, 9)  # partly outside: no fault
    assert clipped.any()


def test_interaction_handlers_match_oracle(rt, oracle):
    """on_mouse (kernel.cu:112-139) and the WASD handling (kernel.cu:51-103): same pose bits as the oracle."""
    host, olib = rt.libs()[1], oracle.lib
    olib.orc_on_mouse.restype = None
    olib.orc_on_key.restype = C.c_int
    rng = np.random.default_rng(8)
    pose_a = np.array([0.3, -2.0, 0.7, 0.4, -0.2, 0.1], np.float32)
    pose_b = pose_a.copy()
    st_a, st_b = np.zeros(4, np.int32), np.zeros(4, np.int32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    events = [(0, 100, 100), (1, 100, 100)] + [(0, int(x), int(y)) for x, y in rng.integers(0, 800, (40, 2))] + [(4, 0, 0), (0, 5, 5), (0, 700, 20)]
    for ev, x, y in events:
        host.rth_on_mouse(fp(pose_a), ip(st_a), ev, x, y)
        olib.orc_on_mouse(fp(pose_b), ip(st_b), ev, x, y)
        assert pose_a.tobytes() == pose_b.tobytes() and np.array_equal(st_a, st_b)
    assert pose_a[3] != np.float32(0.4) and st_a[3] == 0
    for key in "wwadsdwq x":
        ra = host.rth_on_key(fp(pose_a), ord(key))
        rb = olib.orc_on_key(fp(pose_b), ord(key))
        assert ra == rb == (0 if key == "q" else 1) and pose_a.tobytes() == pose_b.tobytes()


def test_obj_float_scanner_matches_strtof(rt):
    """OBJLoader's own float scanner (the fp32 / double fast paths and the strtof fallback) returns, for every token, the
    float glibc's strtof returns -- the function std::stof calls in the reference (OBJLoader.hpp:47-49): correctly rounded,
    prefix rule, inf / nan / hexadecimal forms, no conversion for the same tokens."""
    import ctypes as C
    s = rt.libs()[1]
    s.rth_scan_float.restype = C.c_int
    s.rth_scan_float.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_float)]
    libc = C.CDLL("libc.so.6")
    libc.strtof.restype = C.c_float
    libc.strtof.argtypes = [C.c_char_p, C.POINTER(C.c_char_p)]
    rng = np.random.default_rng(5)
    toks = ["0", "-0", "0.0", "-0.000000", "1", "-1", ".5", "-.5", "5.", "+3.25", "1e5", "1E-5", "1e", "1e+", "1.5abc", "12.5e3x", "1e-40", "1e-46",
            "3.4028235e38", "3.4028236e38", "3.5e38", "1e39", "-1e39", "16777216", "16777217", "16777218", "16777219", "33554434", "33554435",
            "0.1", "0.2", "0.3", "123456789", "1234567890123456789", "12345678901234567890123", "0.000000000000000000000000000001",
            "9007199254740993", "4503599627370497.5", "1.00000005960464477539", "1.0000000596046447753906250", "1.00000005960464477539062500001",
            "inf", "-inf", "infinity", "nan", "NAN", "0x1.8p1", "0x10", "0x", ".", "-", "+", "e5", "abc", "", "--1", "1..2", "1e5e5", "00012.50", "1e0010", "1e-0010",
            "8.5070592e37", "1.17549435e-38", "1.17549428e-38", "5.877472e-39", "1.4e-45", "0.7e-45", "0.70064923e-45", "0.8e-45"]
    for _ in range(4000):
        kind = rng.integers(7)
        x = float(rng.normal(0, 1) * 10.0 ** rng.integers(-12, 12))
        if kind == 0: toks.append("%.6f" % x)
        elif kind == 1: toks.append("%.9g" % x)
        elif kind == 2: toks.append("%.17g" % x)
        elif kind == 3: toks.append("%e" % x)
        elif kind == 4: toks.append("%d" % int(rng.integers(-2 ** 40, 2 ** 40)))
        elif kind == 5:                                                  # exactly between two floats, and one double to either side
            f = np.float32(x)
            mid = (float(f) + float(np.nextafter(f, np.float32(np.inf)))) / 2.0
            toks += [repr(mid), repr(float(np.nextafter(mid, np.inf))), repr(float(np.nextafter(mid, -np.inf))), "%.25f" % mid, "%.30e" % mid]
        else: toks.append("".join(rng.choice(list("0123456789"), int(rng.integers(1, 30)))) + "." + "".join(rng.choice(list("0123456789"), int(rng.integers(0, 30)))))
    bad = []
    for t in toks:
        b = t.encode()
        e2 = C.c_char_p()
        buf = C.create_string_buffer(b)
        want = np.float32(libc.strtof(buf, C.byref(e2)))
        converted = C.cast(e2, C.c_void_p).value != C.addressof(buf)
        got = C.c_float(0)
        ok = s.rth_scan_float(b, len(b), C.byref(got))
        if bool(ok) != bool(converted):
            bad.append((t, "conversion", ok, converted))
        elif ok and not (np.array_equal(np.float32(got.value).view(np.uint32), want.view(np.uint32)) or (np.isnan(want) and np.isnan(got.value))):
            bad.append((t, float(got.value), float(want)))
    assert not bad, bad[:10]


def test_obj_parse_is_the_same_on_one_thread_and_on_many(rt, tmp_path, blob70k, monkeypatch):
    """Files above a megabyte are parsed in pieces on several threads: same triangles, in the same order, bit for bit; a face
    may name vertices defined later in the file, relative indices count the records before the face line across piece
    boundaries, and the first error of the file is the one reported."""
    import ctypes as C
    s = rt.libs()[1]
    s.rth_obj_parse.restype = C.c_int32
    s.rth_obj_parse.argtypes = [C.c_char_p, C.c_int32, C.c_void_p, C.c_int32]
    s.rth_last_error.restype = C.c_char_p

    def parse(path, lenient, threads):
        monkeypatch.setenv("RT_OBJ_THREADS", str(threads))
        n = s.rth_obj_parse(str(path).encode(), lenient, None, 0)
        if n < 0:
            return s.rth_last_error().decode()
        out = np.zeros((n, 18), np.float32)
        assert s.rth_obj_parse(str(path).encode(), lenient, out.ctypes.data, n) == n
        return out

    a = parse(blob70k, 0, 1)
    assert a.shape[0] == 69936
    for t in (2, 5, 8):
        assert np.array_equal(parse(blob70k, 0, t).view(np.uint32), a.view(np.uint32)), t
    # a file with everything the lenient mode accepts, long enough to be cut into pieces: faces before their vertices,
    # polygons, v//vn tokens, negative indices, comments, blank and vn lines
    rng = np.random.default_rng(11)
    lines = ["# mixed"]
    nv = 0
    for k in range(30000):
        lines.append("v %.6f %.6f %.6f" % tuple(rng.normal(0, 1, 3)))
        lines.append("vt %.6f %.6f" % tuple(rng.uniform(0, 1, 2)))
        nv += 1
        if k % 3 == 0:
            lines.append("vn 0 0 1")
        if k >= 4:
            r = k % 5
            if r == 0: lines.append("f -1/-1 -2/-2 -3/-3 -4/-4")
            elif r == 1: lines.append("f %d//1 %d//1 %d//1" % (nv, nv - 1, nv - 2))
            elif r == 2: lines.append("f %d/%d/1 %d/%d/1 %d/%d/1" % (nv, nv, nv - 2, nv - 2, nv - 3, nv - 3))
            elif r == 3: lines.append("   f   %d %d %d   " % (1, nv, nv - 1))
            else: lines.append("")
    lines.insert(1, "f 30000/30000 29999/29999 29998/29998")                      # names vertices defined at the end of the file
    path = tmp_path / "mixed.obj"
    path.write_text("\n".join(lines))
    assert os.path.getsize(path) > (1 << 20)
    one = parse(path, 1, 1)
    assert not isinstance(one, str) and one.shape[0] > 25000
    for t in (3, 7):
        assert np.array_equal(parse(path, 1, t).view(np.uint32), one.view(np.uint32)), t
    assert parse(path, 0, 1) == parse(path, 0, 4) and isinstance(parse(path, 0, 4), str)        # strict mode: the same (first) error
    # first error in file order, whatever piece finds it
    broken = lines[:]
    broken[40001] = "v 1.0 oops 2.0"
    broken[50000] = "vt x y"
    (tmp_path / "broken.obj").write_text("\n".join(broken))
    assert parse(tmp_path / "broken.obj", 1, 1) == parse(tmp_path / "broken.obj", 1, 6) == "malformed v record"


def test_forced_rccl_library_that_cannot_be_loaded_is_an_error(rt, tmp_path):
    """RT_RCCL_LIBRARY names the library to use: if it cannot be loaded, or lacks an entry point, rt_comm_* fail with RT_E_COMM and
    the loader's message -- never a silent switch to the system's RCCL (no GPU involved: loading is all that happens)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import importlib, sys; rt = importlib.import_module('cuda-raytracing_amd'); import ctypes as C\n"
            "h = rt.libs()[0]; v = C.c_int32(0); rc = h.rt_comm_available(C.byref(v)); print(rc, rt.Comm.last_error())")
    bad = str(tmp_path / "not_a_library.so")
    open(bad, "w").write("this is not an ELF file")
    for path, needle in ((str(tmp_path / "missing.so"), "could not be loaded"), (bad, "could not be loaded"), ("libm.so.6", "lacks ncclGetUniqueId")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root, env=dict(os.environ, RT_RCCL_LIBRARY=path))
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.split()[0] == "-5" and needle in r.stdout, r.stdout


def test_comm_error_text_is_readable_from_another_thread(rt, tmp_path):
    """rt_comm_last_error() is the CALLING thread's last RT_E_COMM.  bench.py's phase watchdog is another thread: it reads
    rt_comm_last_error_any(), the most recent error of any thread (it used to print the per-thread text, which is
    always empty there)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import importlib, threading; rt = importlib.import_module('cuda-raytracing_amd'); import ctypes as C\n"
            "h = rt.libs()[0]; v = C.c_int32(0); rc = h.rt_comm_available(C.byref(v))\n"
            "box = {}\n"
            "t = threading.Thread(target=lambda: box.update(own=rt.Comm.last_error(), any=rt.Comm.last_error_any())); t.start(); t.join()\n"
            "print(rc); print('OWN=' + box['own']); print('ANY=' + box['any']); print('MAIN=' + rt.Comm.last_error())")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root,
                       env=dict(os.environ, RT_RCCL_LIBRARY=str(tmp_path / "missing.so")))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "-5"
    assert lines[1] == "OWN="                                           # the other thread made no comm call
    assert "could not be loaded" in lines[2] and lines[2].startswith("ANY=")
    assert lines[3] == "MAIN=" + lines[2][4:]


def test_damaged_obj_files_never_crash_and_parse_the_same_on_any_thread_count(rt, tmp_path, blob5k, monkeypatch):
    """Truncated (also at page-size multiples: the file is memory-mapped, nothing may be read past its end), bit-flipped and
    spliced OBJ files: every one is either parsed or refused with a message -- the same triangles or the same message on one
    thread and on five (a 413-file campaign of the same generator ran clean under AddressSanitizer + UBSan on the CPU build)."""
    import random
    s = rt.libs()[1]
    s.rth_last_error.restype = C.c_char_p
    random.seed(7)
    base = open(blob5k, "rb").read()
    files = []
    for cut in (0, 1, 2, 4095, 4096, 4097, 8192, len(base) - 1):
        files.append(base[:cut])
    inserts = [b"/", b"//", b"-", b"1e999", b"\n", b" ", b"f ", b"v ", b"vt ", b"\r", b"\x00", b"999999999999", b".", b"e", b"+", b"0x1p3", b"nan"]
    for k in range(120):
        src = bytearray(base[:random.randint(1, 20000)] if k % 2 else base[100000:100000 + random.randint(10, 9000)])
        for _ in range(random.randint(1, 10)):
            if not src:
                break
            i, op = random.randrange(len(src)), random.randint(0, 4)
            if op == 0: src[i] = random.randrange(256)
            elif op == 1: del src[i:i + random.randint(1, 20)]
            elif op == 2: src[i:i] = random.choice(inserts)
            elif op == 3: src = src[:i]
            else: src[i:i] = b"f 1/1/1 2/2/2 3/3/3 4/4/4 5\n"
        files.append(bytes(src))
    big = bytearray(base * 5)                                             # above 1 MB: the threaded path, damaged
    for _ in range(40):
        big[random.randrange(len(big))] = random.randrange(256)
    files.append(bytes(big))

    def parse(path, lenient, threads):
        monkeypatch.setenv("RT_OBJ_THREADS", str(threads))
        n = s.rth_obj_parse(path, lenient, None, 0)
        if n < 0:
            return s.rth_last_error().decode(errors="replace")
        out = np.zeros((n, 18), np.float32)
        assert s.rth_obj_parse(path, lenient, out.ctypes.data, n) == n
        return out.view(np.uint32).tobytes()
    parsed = 0
    for k, data in enumerate(files):
        path = str(tmp_path / ("d%03d.obj" % k))
        open(path, "wb").write(data)
        for lenient in (0, 1):
            a, b = parse(path.encode(), lenient, 1), parse(path.encode(), lenient, 5)
            assert a == b, (k, lenient)
            parsed += not isinstance(a, str)
    assert 20 < parsed < 2 * len(files)
