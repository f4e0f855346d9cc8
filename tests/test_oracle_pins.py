"""The oracle (oracle/rt_oracle.c) against everything that pins it:
  * golden vectors generated from the REFERENCE's own headers (tests/golden/make_golden.py, via
    oracle/_ref/libref_probe.so) -- L0 math, AABB slab, triangle tests, BVH topology, OBJ loading;
  * the known-answer values and full-frame hashes SURVEY.md recorded from the reference's render();
  * when oracle/_ref is present (build container), the reference code itself on fresh random inputs."""
import hashlib
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same(a, b):
    return np.array_equal(bits(a), bits(b))


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "l0_math.npz"))


@pytest.fixture(scope="module")
def pins():
    return json.load(open(os.path.join(GOLDEN, "pins.json")))


def test_rsqrt_kat(oracle, pins):
    k = pins["rsqrt_kat_hex"]
    got = ["%08x" % int(bits([oracle.q_rsqrt(x)])[0]) for x in k["inputs"]]
    assert got == k["bits"]


def test_l0_math_golden(oracle, g):
    o = oracle
    assert same([o.q_rsqrt(float(v)) for v in g["rsqrt_in"]], g["rsqrt_out"])
    assert same(np.stack([o.normalize(v) for v in g["vec_in"]]), g["normalize_out"])
    assert same([o.magnitude(v) for v in g["vec_in"]], g["magnitude_out"])
    P, V = g["pose_in"], g["pose_vec_in"]
    assert same(np.stack([o.invert_lre(p) for p in P]), g["invert_lre_out"])
    assert same(np.stack([o.lre2homo(p) for p in P]), g["lre2homo_out"])
    assert same(np.stack([o.euler2quat(p[3:]) for p in P]), g["euler2quat_out"])
    assert same(np.stack([o.apply_lre(p, v) for p, v in zip(P, V)]), g["apply_lre_out"])
    assert same(np.stack([o.apply_euler(p[3:], v) for p, v in zip(P, V)]), g["apply_euler_out"])
    assert same(np.stack([o.invert_intrinsic(k) for k in g["K_in"]]), g["invert_intrinsic_out"])


def test_ray_generation_pins(oracle, g):
    """The two reference operations that open the path and were only pinned indirectly before: apply_matrix(float3x3, float3)
    (utils.hpp:134-140; K_inv x (x, y, 1), raycast.cu:161) and the Ray constructor (Ray.hpp:17-23: direction_inv = 1 / d,
    also for zero, negative-zero, denormal and huge components; color (1,1,1), illumination 0)."""
    o = oracle
    assert same(np.stack([o.apply_matrix33(m, v) for m, v in zip(g["matrix33_in"], g["matrix33_vec_in"])]), g["apply_matrix33_out"])
    R = g["ray_in"]
    got = np.stack([o.ray_ctor(r[:3], r[3:]) for r in R])
    assert same(got, g["ray_ctor_out"])
    assert np.isinf(g["ray_ctor_out"][0, 6]) and np.signbit(g["ray_ctor_out"][0, 8])       # 1 / 0 = +inf, 1 / -0 = -inf


def test_survey_kats(oracle):
    """SURVEY.md section 4 known-answer rows."""
    o = oracle
    np.testing.assert_allclose(o.normalize([3, 4, 12]), [0.230371147, 0.30716154, 0.92148459], rtol=0, atol=1e-9)
    np.testing.assert_allclose(o.invert_lre([1, 2, 3, 0.3, -0.2, 0.1]),
                               [-0.0251887739, -1.56620836, -3.3979938, -0.322609663, 0.160027221, -0.156419501], rtol=2e-7)
    np.testing.assert_allclose(o.apply_lre([1, 2, 3, 0.3, -0.2, 0.1], [0.5, -1.5, 2.5]), [0.672042012, -3.3225069, -1.12218881], rtol=2e-7)
    np.testing.assert_allclose(o.invert_lre([-1, -4, 2, 0, 0, 0]), [1, 4, -2, 0, 0, 0], atol=0)


def test_aabb_and_triangle_golden(oracle, g):
    o = oracle
    A = g["aabb_in"]
    got = np.array([o.aabb(a[0:3], a[3:6], a[6:9], a[9:12]) for a in A], np.float32)
    assert same(got, g["aabb_out"])
    assert (got == np.float32(3.4028234663852886e38)).sum() > 20          # misses are present
    # 3-vertex constructor: vertices + normal (the reference leaves uv_coords uninitialised, so only 12 floats are pinned)
    built = np.stack([o.tri_from_vertices(a) for a in g["tri_abc_in"]])
    assert same(built[:, :12], g["tri_from_vertices_out"][:, :12])
    T, R = g["tri_from_vertices_out"], g["tri_ray_in"]
    tt = np.stack([o.tri_test(t, r[:3], r[3:]) for t, r in zip(T, R)])
    assert same(tt, g["tri_test_out"])
    inside = (tt[:, 3] != np.float32(3.4028234663852886e38)).sum()
    assert 20 < inside < len(T)                                             # both outcomes exercised
    assert same(np.stack([o.tri_center(t) for t in T]), g["tri_center_out"])


def _mesh_equal(d, e):
    for k in ("boxes", "tris"):
        if k in e:
            assert same(d[k], e[k]), k
    for k in ("child", "leaf_count", "leaf_idx"):
        assert np.array_equal(d[k], e[k]), k


def test_bvh_topology_golden(oracle, scenes, blob5k, pins):
    assert hashlib.sha256(open(blob5k, "rb").read()).hexdigest() == pins["frames"]["blob5k_obj_sha256"]
    d = oracle.mesh_dump(oracle.obj_load(blob5k))
    e = np.load(os.path.join(GOLDEN, "blob5k_bvh.npz"))
    assert d["child"].shape[0] == pins["blob5k_num_nodes"] and d["tris"].shape[0] == pins["blob5k_num_tris"]
    _mesh_equal(d, {k: e[k] for k in ("child", "leaf_count", "leaf_idx")})
    assert hashlib.sha256(d["tris"].tobytes()).hexdigest() == pins["blob5k_tris_sha256"]
    assert hashlib.sha256(d["boxes"].tobytes()).hexdigest() == pins["blob5k_boxes_sha256"]
    assert same(d["boxes"][:64], e["boxes_head"]) and same(d["tris"][:64], e["tris_head"])


def test_obj_loader_golden(oracle):
    d = oracle.mesh_dump(oracle.obj_load(os.path.join(GOLDEN, "small_mixed.obj")))
    e = np.load(os.path.join(GOLDEN, "small_mixed_mesh.npz"))
    assert d["tris"].shape[0] == 8                                           # quad + tri + tri + pentagon + tri
    _mesh_equal(d, e)


def test_unsplittable_leaf_golden(oracle):
    e = np.load(os.path.join(GOLDEN, "soup_mesh.npz"))
    d = oracle.mesh_dump(oracle.mesh_from_triangles(e["tris_in"]))
    _mesh_equal(d, {k: e[k] for k in ("boxes", "child", "leaf_count", "leaf_idx")})
    assert d["leaf_count"].max() >= 35


def test_obj_errors(oracle, tmp_path):
    assert not oracle.obj_load(str(tmp_path / "missing.obj"))
    p = tmp_path / "vn_only.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    assert not oracle.obj_load(str(p))                                       # stoi("/1") throws in the reference (H11)
    q = tmp_path / "range.obj"
    q.write_text("v 0 0 0\nv 1 0 0\nf 1 2 3\n")
    assert not oracle.obj_load(str(q))


def test_c1_frame_pin(orc, scenes, pins):
    """BASELINE.json configs[0]: the reference's own CPU-runnable case; hash from SURVEY.md section 4."""
    import scene_defs as sd
    c = scenes.C1
    s = sd.c1_scene(scenes).build_oracle(orc)
    out = s.render(c["width"], c["height"], c["K"], c["D"], c["cam_pose"])
    pin = pins["frames"]["C1_256x256"]
    assert out["stats"]["hits"] == pin["hit_pixels"]
    assert orc.fnv1a64(out["img"]) == pin["fnv1a64"]
    hit = out["hit_tri"] >= 0
    ys, xs = np.nonzero(hit)
    assert (xs.min(), xs.max(), ys.min(), ys.max()) == (97, 159, 98, 159)
    assert (out["img"][hit] == np.array([25, 51, 229], np.uint8)).all()
    assert tuple(out["img"][128, 128]) == (255, 204, 153)
    s.close()


def test_extension_degenerates_to_pinned_frames(orc, scenes, blob5k, pins):
    """orc_render_ex(spp=1, bounces=0, lighting=0) must be the reference frame: C1 by its SURVEY hash, a blob scene by
    equality with orc_render (RGB and node pops).  This is the only pin the extension semantics have."""
    import scene_defs as sd
    c = scenes.C1
    s = sd.c1_scene(scenes).build_oracle(orc)
    ex = s.render_ex(c["width"], c["height"], c["K"], c["D"], c["cam_pose"], 1, 0, 0)
    assert orc.fnv1a64(ex["img"]) == pins["frames"]["C1_256x256"]["fnv1a64"]
    s.close()
    m = sd.SHINY_CAMERA
    s2 = sd.shiny_scene(scenes, blob5k).build_oracle(orc)
    K = scenes.scaled_K(m["width"])
    a = s2.render(m["width"], m["height"], K, scenes.D_REF, m["pose"], threads=4)
    b = s2.render_ex(m["width"], m["height"], K, scenes.D_REF, m["pose"], 1, 0, 0, threads=4)
    assert np.array_equal(a["img"], b["img"]) and np.array_equal(a["pops"], b["total_pops"])
    lit = s2.render_ex(m["width"], m["height"], K, scenes.D_REF, m["pose"], 4, 2, 1, threads=4)
    assert lit["stats"]["rays"] > 4 * m["width"] * m["height"] and not np.array_equal(lit["img"], a["img"])
    s2.close()


@pytest.mark.parametrize("cam", ["far", "mid", "near"])
def test_c2_frame_pins(orc, scenes, blob70k, pins, cam):
    """BASELINE.json configs[1] at full size: oracle frame hash == the hash SURVEY.md 8(d) recorded from the reference's
    render() on the byte-identical OBJ, the frame's visit statistics == the ones SURVEY.md section 6 printed from the same
    reference run (node pops, AABB tests and triangle tests per ray, deepest stack, coverage -- everything the survey wrote
    down about cast_ray's control flow), and the oracle's own planes == their frozen hashes (self-regression: the oracle
    and the kernels are compared with each other on every run, this keeps the pair from drifting together)."""
    import scene_defs as sd
    assert hashlib.sha256(open(blob70k, "rb").read()).hexdigest() == pins["frames"]["blob70k_obj_sha256"]
    s = sd.blob_scene(scenes, blob70k).build_oracle(orc)
    c = scenes.C2
    out = s.render(c["width"], c["height"], scenes.scaled_K(c["width"]), c["D"], scenes.C2_CAMERAS[cam], planes=True, threads=8)
    assert orc.fnv1a64(out["img"]) == pins["frames"]["C2_%s_1920x1080" % cam]["fnv1a64"]
    st, want = out["stats"], pins["survey_c2_stats"][cam]
    rays = st["rays"]
    assert rays == c["width"] * c["height"]
    assert round(st["pops"] / rays, 2) == want["pops_per_ray"]
    assert round(st["aabb"] / rays, 2) == want["aabb_per_ray"]
    assert round(st["tris"] / rays, 2) == want["tris_per_ray"]
    assert st["max_stack"] == want["max_stack"]
    assert round(100.0 * st["hits"] / rays, 1) == want["coverage_pct"]
    # bytes per ray as the survey derived them from the same run: the reference's AoS layout (section 6, to the byte) and the
    # SoA accounting of section 8(d) that bench.py's `hbm_algorithmic` uses ("approx": two to three significant digits)
    aos = (48 * (st["pops"] + st["aabb"]) + 76 * st["tris"]) / rays
    assert round(aos) == want["aos_bytes_per_ray"]
    interior = st["aabb"] // 2
    soa = (24 * st["aabb"] + 8 * interior + 8 * (st["pops"] - interior) + 52 * st["tris"] + 24 * st["inside"] + 3 * rays) / rays
    assert abs(soa - want["soa_bytes_per_ray_approx"]) <= 0.015 * want["soa_bytes_per_ray_approx"]
    # the planes add up to the frame totals, and equal their frozen hashes
    assert int(out["pops"].sum()) == st["pops"] and int(out["aabb"].sum()) == st["aabb"] and int(out["tris"].sum()) == st["tris"]
    _check_frozen_planes(out, pins, "C2_%s_1920x1080" % cam)
    s.close()


def _check_frozen_planes(out, pins, name):
    frozen = pins["oracle_planes_self_regression"]["frames"][name]
    for k in ("hit_inst", "hit_tri", "pops", "aabb", "tris", "inside"):
        assert hashlib.sha256(np.ascontiguousarray(out[k], np.int32).tobytes()).hexdigest() == frozen[k], \
            "%s: the oracle's %s plane changed (tests/golden/make_plane_pins.py re-freezes it -- only after the change is understood)" % (name, k)
    assert hashlib.sha256(out["img"].tobytes()).hexdigest() == frozen["img_sha256"], name


def test_frozen_oracle_planes_small_scenes(orc, scenes, blob70k, blob5k, pins):
    """Self-regression pins (NOT reference-derived) of the oracle's hit-id and visit-count planes: C1 and the multi-instance
    textured scene (rotated, non-uniformly scaled instances, two meshes, a texture)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_plane_pins", os.path.join(GOLDEN, "make_plane_pins.py"))
    mpp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mpp)
    import scene_defs as sd
    for name, (desc, W, H, K, pose) in mpp.cases(scenes, sd, blob70k, blob5k).items():
        if name.startswith("C2_"):
            continue                                                    # (full size: test_c2_frame_pins)
        s = desc.build_oracle(orc)
        _check_frozen_planes(s.render(W, H, K, scenes.D_REF, pose, threads=4), pins, name)
        s.close()


def test_blob5k_bvh_stats_survey(oracle, blob5k, pins):
    """SURVEY.md line 255: the reference builder on the 5 000-triangle mesh gives 9 197 nodes, 4 599 leaves, depth 12."""
    st = oracle.mesh_stats(oracle.obj_load(blob5k))
    p = pins["survey_c2_stats"]["blob5k_bvh"]
    assert (st["nodes"], st["leaves"], st["max_depth"]) == (p["nodes"], p["leaves"], p["print_stats_max_depth"])


def test_blob70k_bvh_stats(orc, oracle, blob70k, pins):
    st = oracle.mesh_stats(oracle.obj_load(blob70k))
    p = pins["frames"]["blob70k_bvh"]
    assert (st["nodes"], st["leaves"], st["max_tris"], st["max_depth"]) == (p["nodes"], p["leaves"], p["max_tris_per_leaf"], p["print_stats_max_depth"])


# ---- live comparison with the reference's own code (only where oracle/_ref was built) -----------------

def test_oracle_vs_reference_live(orc, oracle):
    r = orc.ref_probe()
    if r is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    rng = np.random.default_rng(99)
    for _ in range(300):
        p = np.concatenate([rng.uniform(-5, 5, 3), rng.uniform(-3.2, 3.2, 3)]).astype(np.float32)
        v = rng.normal(0, 2, 3).astype(np.float32)
        assert same(oracle.invert_lre(p), r.invert_lre(p))
        assert same(oracle.apply_lre(p, v), r.apply_lre(p, v))
        assert same(oracle.normalize(v), r.normalize(v))
        bmin = rng.uniform(-2, 1, 3).astype(np.float32)
        bmax = (bmin + rng.uniform(0, 2, 3)).astype(np.float32)
        o3, d3 = rng.uniform(-4, 4, 3).astype(np.float32), rng.normal(0, 1, 3).astype(np.float32)
        assert same(oracle.aabb(bmin, bmax, o3, d3), r.aabb(bmin, bmax, o3, d3))
        t = oracle.tri_from_vertices(rng.uniform(-1, 1, 9).astype(np.float32))
        assert same(t[:12], r.tri_from_vertices(t[:9])[:12])
        assert same(oracle.tri_test(t, o3, d3), r.tri_test(t, o3, d3))
        m9 = rng.normal(0, 2, 9).astype(np.float32)
        assert same(oracle.apply_matrix33(m9, v), r.apply_matrix33(m9, v))
        assert same(oracle.ray_ctor(o3, d3), r.ray_ctor(o3, d3))


def test_oracle_bvh_vs_reference_live(orc, oracle):
    r = orc.ref_probe()
    if r is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    import scene_defs as sd
    for seed, n in [(1, 1), (2, 2), (3, 17), (4, 900)]:
        tris = sd.random_triangles(n, seed=seed)
        _mesh_equal(oracle.mesh_dump(oracle.mesh_from_triangles(tris)), r.mesh_dump(r.mesh_from_triangles(tris)))


# ---- the same live comparison where the arithmetic is least comfortable (round 6, third session) ---------------------------
# NaN results are compared as NaN: the reference's functions and the restatement may differ in a NaN's sign / payload bits (x87-free SSE
# arithmetic propagates the FIRST operand's payload, and the two are compiled by different front ends), which nothing downstream can
# observe -- every consumer compares or multiplies the value.

def same_or_both_nan(a, b):
    a = np.ascontiguousarray(a, np.float32).ravel()
    b = np.ascontiguousarray(b, np.float32).ravel()
    return a.shape == b.shape and bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


_FI = np.finfo(np.float32)
_SPECIAL = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, _FI.tiny, -_FI.tiny, 1e-45, -1e-45, _FI.max, -_FI.max, np.inf, -np.inf, np.nan,
                     1e-20, -1e-20, 1e18, -1e18, 1e30, 3.0], np.float32)


def _draw(rng, n, p_special, sigma=2.0):
    v = rng.normal(0, sigma, n).astype(np.float32)
    m = rng.random(n) < p_special
    v[m] = rng.choice(_SPECIAL, int(m.sum()))
    return v


def test_oracle_vs_reference_live_on_special_values(orc, oracle):
    """Every L0 / L1 function cast_ray and render call (BVHTree.hpp:40-54 slab test, TrianglePrimitive.hpp:62-79 / :151-185 plane and
    barycentric tests, the Ray constructor, utils.hpp / transforms.hpp), restatement against the reference's own compiled code, on
    inputs drawn from zeros of both signs, denormals, FLT_MIN / FLT_MAX, infinities, NaN, 1e-20 / 1e18 / 1e30 mixed with ordinary
    values: flat and inverted boxes, origins on box faces, rays parallel to axes, zero-area triangles, rays aimed at a triangle from
    on or near its plane."""
    r = orc.ref_probe()
    if r is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    rng = np.random.default_rng(4321)
    F = np.float32
    with np.errstate(all="ignore"):
        for it in range(6000):
            ps = (0.0, 0.15, 0.5, 0.5)[it % 4]
            bmin, bmax = _draw(rng, 3, ps), _draw(rng, 3, ps)
            if rng.random() < 0.6:
                bmax = np.maximum(bmin, bmax)
            if rng.random() < 0.3:
                k = int(rng.integers(3)); bmax[k] = bmin[k]
            o3, d3 = _draw(rng, 3, ps), _draw(rng, 3, ps, 1.0)
            if rng.random() < 0.3:
                o3[int(rng.integers(3))] = bmin[int(rng.integers(3))]
            if rng.random() < 0.3:
                d3[int(rng.integers(3))] = F(0.0) if rng.random() < 0.5 else F(-0.0)
            assert same_or_both_nan(oracle.aabb(bmin, bmax, o3, d3), r.aabb(bmin, bmax, o3, d3)), ("aabb", bmin, bmax, o3, d3)
            v9 = _draw(rng, 9, ps, 1.0)
            if rng.random() < 0.3:
                v9[3:6] = v9[0:3]
            if rng.random() < 0.2:
                v9[6:9] = v9[0:3] + F(2) * (v9[3:6] - v9[0:3])
            t = oracle.tri_from_vertices(v9)
            assert same_or_both_nan(t[:12], r.tri_from_vertices(v9)[:12]), ("tri_from_vertices", v9)
            assert same_or_both_nan(oracle.tri_test(t, o3, d3), r.tri_test(t, o3, d3)), ("tri_test", t, o3, d3)
            w = rng.dirichlet((1, 1, 1)).astype(np.float32)
            o4 = _draw(rng, 3, ps * 0.3)
            d4 = ((w[0] * t[0:3] + w[1] * t[3:6] + w[2] * t[6:9]).astype(np.float32) - o4).astype(np.float32)
            assert same_or_both_nan(oracle.tri_test(t, o4, d4), r.tri_test(t, o4, d4)), ("tri_test, aimed", t, o4, d4)
            assert same_or_both_nan(oracle.tri_center(t), r.tri_center(t))
            v = _draw(rng, 3, ps)
            p = np.concatenate([_draw(rng, 3, ps), _draw(rng, 3, ps, 3.0)])
            e, q, m9 = _draw(rng, 3, ps, 3.0), _draw(rng, 4, ps, 1.0), _draw(rng, 9, ps)
            for name, args in (("normalize", (v,)), ("invert_lre", (p,)), ("apply_lre", (p, v)), ("euler2quat", (e,)), ("apply_quat", (q, v)),
                               ("apply_euler", (e, v)), ("apply_matrix33", (m9, v)), ("invert_intrinsic", (m9,)), ("ray_ctor", (o3, d3)), ("lre2homo", (p,))):
                assert same_or_both_nan(getattr(oracle, name)(*args), getattr(r, name)(*args)), (name, args)
            assert same_or_both_nan(oracle.magnitude(v), r.magnitude(v)), ("magnitude", v)
            x = float(_draw(rng, 1, ps)[0])
            assert same_or_both_nan(oracle.q_rsqrt(x), r.q_rsqrt(x)), ("q_rsqrt", x)


def test_oracle_bvh_vs_reference_live_on_awkward_meshes(orc, oracle):
    """BVHTree::fill (BVHTree.hpp:203-292), restatement against the reference's own builder, on the mesh kinds of the adversarial GPU
    fuzz (tests/test_gpu_parity.py _adversarial_mesh): lattice vertices (flat boxes, coincident centroids), zero-area triangles, piles
    of coincident triangles (unsplittable leaves above 30), coordinates at 1e6..1e18 and 1e-6..1e-20, slivers, non-finite vertices."""
    r = orc.ref_probe()
    if r is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    import collections
    import test_gpu_parity as gp
    seen = collections.Counter()
    for seed in range(600):
        kind, tris = gp._adversarial_mesh(oracle, np.random.default_rng(52000 + seed))
        d, e = oracle.mesh_dump(oracle.mesh_from_triangles(tris)), r.mesh_dump(r.mesh_from_triangles(tris))
        assert same_or_both_nan(d["boxes"], e["boxes"]) and same_or_both_nan(d["tris"], e["tris"]), (seed, kind)
        for k in ("child", "leaf_count", "leaf_idx"):
            assert np.array_equal(d[k], e[k]), (seed, kind, k)
        seen[kind] += 1
    assert len(seen) == 8 and min(seen.values()) >= 30, seen
