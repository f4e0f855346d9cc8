"""GPU parity: the HIP path (through the C-ABI, driven by the host C++ API) against the CPU oracle
on the same scene, camera and pixels.  Bit-exact: RGB bytes, hit (instance, triangle) ids,
node-pop / AABB-test / triangle-test / inside-hit counts per pixel."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import scene_defs as sd

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
PLANES = ("hit_inst", "hit_tri", "pops", "aabb", "tris", "inside")


def _camera(rt, scenes, width, height, K, pose):
    cam = rt.Camera(width, height, K, scenes.D_REF)
    cam.set_pose(pose)
    return cam


def _compare(rt, orc, desc, width, height, K, D, pose, threads=8, gpu_build=False):
    so = desc.build_oracle(orc)
    ref = so.render(width, height, K, D, pose, threads=threads)
    sp = desc.build_product(rt, gpu_build=gpu_build)
    sp.upload_to_device()
    cam = rt.Camera(width, height, K, D)
    cam.set_pose(pose)
    dbg = rt.render_debug(sp, cam)
    img = rt.render(sp, cam)
    assert np.array_equal(img, dbg["img"]), "debug and production kernels disagree"
    ids = rt.render_ids(sp, cam)                                # the production (timed) kernel's own hit ids, raycast.cu:107-127
    assert np.array_equal(ids["img"], img)
    for n in ("hit_inst", "hit_tri"):
        bad = int((ids[n] != ref[n]).sum())
        assert bad == 0, "production kernel %s: %d pixels differ from the oracle" % (n, bad)
    nbad = int((img != ref["img"]).any(axis=2).sum())
    assert nbad == 0, "%d pixels differ from the oracle" % nbad
    for n in PLANES:
        bad = int((dbg[n] != ref[n]).sum())
        assert bad == 0, "%s: %d pixels differ from the oracle" % (n, bad)
    so.close()
    # a launch of four frames renders through VIEW records where the scene allows them (at most eight instances; tests/conftest.py
    # lifts the rays-per-record threshold, so that frames of any size do): all four must be the frame above
    bufs = [rt.DeviceBuffer(width_bytes=width * 3, height=height) for _ in range(4)]
    cam.render_scene_batch(sp, [pose] * 4, [b.ptr for b in bufs], bufs[0].pitch, synchronize=True)
    for k, b in enumerate(bufs):
        assert np.array_equal(b.to_host().reshape(height, width, 3), img), "frame %d of a batch of four differs (view records %s)" % (k, sp.view_stats())
        b.free()
    return img, ref


def test_c1_single_triangle(rt, orc, scenes):
    """BASELINE.json configs[0]; frame hash and hit count recorded in SURVEY.md section 4."""
    c = scenes.C1
    img, ref = _compare(rt, orc, sd.c1_scene(scenes), c["width"], c["height"], c["K"], c["D"], c["cam_pose"])
    assert ref["stats"]["hits"] == 1870
    assert orc.fnv1a64(img) == "6b05ef62c4ffefb7"
    assert tuple(img[128, 128]) == (255, 204, 153)          # NaN ray at the principal point renders sky (H5)


@pytest.mark.parametrize("cam", ["far", "mid", "near"])
def test_blob5k_small(rt, orc, scenes, blob5k, cam):
    _compare(rt, orc, sd.blob_scene(scenes, blob5k), 480, 270, scenes.scaled_K(480), scenes.D_REF, scenes.C2_CAMERAS[cam])


@pytest.mark.parametrize("cam,fnv", [("far", "1987bc58fc5f9ed0"), ("mid", "a85de3d252fa5a4f"), ("near", "c78e8858fb3f7613")])
def test_c2_blob70k_1080p(rt, orc, scenes, blob70k, cam, fnv):
    """BASELINE.json configs[1] at full size: every pixel against the oracle, and the frame hash
    against the one SURVEY.md 8(d) recorded from the reference's own render()."""
    c = scenes.C2
    img, _ = _compare(rt, orc, sd.blob_scene(scenes, blob70k), c["width"], c["height"], scenes.scaled_K(c["width"]),
                      c["D"], scenes.C2_CAMERAS[cam], threads=16)
    assert orc.fnv1a64(img) == fnv


def test_multi_instance_textured(rt, orc, scenes, blob5k):
    m = sd.MULTI_CAMERA
    _compare(rt, orc, sd.multi_instance_scene(scenes, blob5k), m["width"], m["height"], scenes.scaled_K(m["width"]),
             scenes.D_REF, m["pose"])


def test_texture_files_and_display_image(rt, orc, scenes, blob5k, tmp_path):
    """Material::upload_texture from a JPEG / PNG file (the reference: cv::imread) renders the same frame as the oracle
    given the decoded pixels; display_image (kernel.cu:30-43) writes that frame with the FPS overlay as a PNG."""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "images")
    W, H = 320, 200
    K, pose = scenes.scaled_K(W), sd.MULTI_CAMERA["pose"]
    for name in ("jpg_420_q35_opt.jpg", "png_P.png"):
        tex = rt.read_image(os.path.join(gold, name))
        so = sd.SceneDesc([((1.0, 1.0, 1.0), tex)], [("obj", blob5k)], [(0, 0, (0,) * 6, (1, 1, 1))]).build_oracle(orc)
        ref = so.render(W, H, K, scenes.D_REF, pose, threads=8)["img"]
        sp = rt.Scene()
        sp.add_material((1.0, 1.0, 1.0), texture_path=os.path.join(gold, name))
        sp.add_mesh(rt.Mesh.load_obj(blob5k))
        sp.add_mesh_instance(0, 0)
        sp.upload_to_device()
        cam = _camera(rt, scenes, W, H, K, pose)
        img = rt.DeviceBuffer(width_bytes=W * 3, height=H)
        cam.render_scene(sp, img.ptr, img.pitch, synchronize=True)
        got = img.to_host().reshape(H, W, 3)
        assert np.array_equal(got, ref), name
        assert len(np.unique(got.reshape(-1, 3), axis=0)) > 20                  # the texture really is on screen
    out = str(tmp_path / "out.png")
    rt.check(rt.libs()[1].rth_display_image(img.ptr, W, H, img.pitch, 123.456789, os.fsencode(out)))
    shown = rt.read_image(out)
    expect = got.copy()
    rt.libs()[1].rth_overlay_text_bgr(expect.ctypes.data, W, H, expect.strides[0], b"FPS: 123.456789", 10, 30, 3, 0, 255, 0)
    assert np.array_equal(shown, expect) and not np.array_equal(shown, got)


def test_identity_instance_with_negative_zeros(rt, orc, scenes, blob5k):
    """The production kernel skips the mesh -> world transform of raycast.cu:98-104 for instances whose inverse pose is
    the identity (rt_kernels.hip, triangle_test).  -0.0 components compare equal to 0 and keep that shortcut on, while the
    oracle runs the full scale / translate / rotate sequence: hit ids, RGB (textured, so uv matters) and counts must agree."""
    nz = np.float32(-0.0)
    tex = sd.checker_texture(40, 24, seed=9)
    for pose in ((nz, 0, nz, 0, 0, 0), (nz, nz, nz, nz, nz, nz), (0, nz, 0, nz, 0, nz)):
        d = sd.SceneDesc([((1.0, 1.0, 1.0), tex)], [("obj", blob5k)], [(0, 0, pose, (1, 1, 1))])
        _compare(rt, orc, d, 320, 180, scenes.scaled_K(320), scenes.D_REF, scenes.C2_CAMERAS["mid"])
        _compare(rt, orc, d, 160, 90, scenes.scaled_K(160), scenes.D_REF, (0.3, -1.4, 0.1, 0.2, -0.1, 0.05))


def test_ragged_sizes(rt, orc, scenes, blob5k):
    """Widths/heights that are not multiples of the 16x16 tile, down to 1x1."""
    for w, h in [(1, 1), (17, 9), (250, 131)]:
        _compare(rt, orc, sd.blob_scene(scenes, blob5k), w, h, scenes.scaled_K(w), scenes.D_REF, scenes.C2_CAMERAS["mid"])


def test_large_leaves_and_duplicates(rt, orc, scenes):
    """Coincident triangles cannot be split by centroid: leaves with > 30 triangles (leaf_count lookup path)
    and equal-distance candidates (first visited wins, raycast.cu:109)."""
    base = sd.random_triangles(6, seed=3, spread=0.5, size=0.6)
    tris = np.concatenate([np.repeat(base[:1], 40, axis=0), np.repeat(base[1:2], 33, axis=0), base[2:]])
    d = sd.SceneDesc([((0.3, 0.6, 0.9), None)], [("tris", tris)], [(0, 0, (0,) * 6, (1, 1, 1))])
    _compare(rt, orc, d, 160, 120, scenes.scaled_K(160), scenes.D_REF, (0.0, -2.5, 0.0, 0, 0, 0))


@pytest.mark.parametrize("kind", ["degenerate", "axis_aligned", "huge", "tiny", "mixed"])
def test_degenerate_and_extreme_geometry(rt, orc, scenes, kind):
    """Inputs on which the arithmetic leaves the comfortable range -- zero-area triangles (NaN normals), triangles and rays
    exactly parallel to the axes (0 * inf in the slab test), coordinates near 1e18 (squares overflow) and near 1e-20
    (products underflow to denormals): every plane and the RGB must still equal the oracle's bit for bit."""
    import orc as orc_mod
    o = orc_mod.oracle()
    rng = np.random.default_rng({"degenerate": 1, "axis_aligned": 2, "huge": 3, "tiny": 4, "mixed": 5}[kind])

    def tris_from(v):
        out = np.stack([o.tri_from_vertices(np.asarray(t, np.float32).ravel()) for t in v])
        out[:, 12:18] = rng.uniform(0, 1, (len(out), 6)).astype(np.float32)
        return out

    normal = sd.random_triangles(40, seed=77, spread=0.8, size=0.4)
    pose, scale = (0.0, -3.0, 0.0, 0, 0, 0), (1, 1, 1)
    if kind == "degenerate":
        v = rng.uniform(-1, 1, (30, 3, 3)).astype(np.float32)
        v[:10, 1] = v[:10, 0]                                            # two equal vertices
        v[10:20, 2] = v[10:20, 0] + 2 * (v[10:20, 1] - v[10:20, 0])      # collinear
        v[20:25] = v[20:25, :1]                                          # a point
        tris = np.concatenate([tris_from(v), normal])
    elif kind == "axis_aligned":
        v = np.zeros((24, 3, 3), np.float32)
        for i in range(24):
            ax = i % 3
            p = rng.integers(-2, 3, (3, 3)).astype(np.float32) * 0.5
            p[:, ax] = np.float32(rng.integers(-1, 2)) * 0.5             # the triangle lies in a plane x / y / z = const: flat boxes
            v[i] = p
        tris = tris_from(v)
        pose = (0.0, -3.0, 0.0, 0, 0, 0)
    elif kind == "huge":
        v = (rng.uniform(-1, 1, (30, 3, 3)) * 1e18).astype(np.float32)
        v[:, :, 1] += np.float32(2e18)
        tris = np.concatenate([tris_from(v), normal])
    elif kind == "tiny":
        v = (rng.uniform(-1, 1, (30, 3, 3)) * 1e-20).astype(np.float32)
        v[:, :, 1] += np.float32(1e-19)
        tris = tris_from(v)
        pose = (0.0, -1e-19, 0.0, 0, 0, 0)
    else:
        v = rng.uniform(-1, 1, (40, 3, 3)).astype(np.float32)
        v[:8] *= np.float32(1e12); v[8:16] *= np.float32(1e-12); v[16:20, 1] = v[16:20, 0]
        tris = np.concatenate([tris_from(v), normal])
        scale = (0.5, 1.0, 2.0)
    d = sd.SceneDesc([((0.3, 0.6, 0.9), sd.checker_texture(8, 8, seed=1))], [("tris", tris)], [(0, 0, (0,) * 6, scale)])
    img, ref = _compare(rt, orc, d, 96, 64, scenes.scaled_K(96), scenes.D_REF, pose)
    hits = int((ref["hit_tri"] >= 0).sum())
    print(kind, "hits", hits, "tri tests", int(ref["tris"].sum()), "inside", int(ref["inside"].sum()))
    assert ref["tris"].sum() > 0 and (hits > 0 or kind == "tiny")


@pytest.mark.parametrize("gpu_build", [False, True])
def test_non_finite_vertices_and_the_octant_preconditions(rt, orc, scenes, gpu_build):
    """The octant-specialised slab test (rt_kernels.hip slab_oct) is only valid for boxes with min <= max and no NaN, rays with
    a finite origin and finite non-zero direction inverses of one sign pattern per wave; everything else must take the generic
    loop.  Triangles with NaN and infinite coordinates (NaN-ignoring min / max leave such boxes partly infinite, a node whose
    triangles are all NaN on an axis keeps (FLT_MAX, -FLT_MAX): min > max -- the mesh is flagged by whoever writes its interior
    records: upload, device build, refit); a camera whose central rays are exactly parallel to two axes (dinv = inf) and whose
    8x8 tiles straddle all sign octants; a refit that moves NaNs into a mesh that had none.  All planes against the oracle."""
    import orc as orc_mod
    o = orc_mod.oracle()
    rng = np.random.default_rng(41)
    clean = sd.random_triangles(120, seed=21, spread=0.9, size=0.35)
    bad = clean.copy()
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    bad[3, 0] = nan                                              # one NaN coordinate
    bad[10, 0:9] = nan                                           # a triangle of NaNs
    bad[17, 1] = inf; bad[18, 5] = -inf                          # infinite coordinates
    bad[25:31, 2] = nan; bad[25:31, 5] = nan; bad[25:31, 8] = nan   # six triangles with no z at all
    for i in (3, 10, 17, 18, 25, 26, 27, 28, 29, 30):
        bad[i, :12] = o.tri_from_vertices(bad[i, :9])[:12]
    W, H = 128, 96
    K = scenes.scaled_K(W)
    # looking along +y from a point with x = z = 0: the centre pixels' rays have direction components that are exactly zero
    K_centred = (K[0], 0.0, W / 2.0, 0.0, K[4], H / 2.0, 0.0, 0.0, 1.0)
    for tris, name in ((bad, "non-finite"), (clean, "clean")):
        d = sd.SceneDesc([((0.3, 0.6, 0.9), sd.checker_texture(8, 8, seed=2))], [("tris", tris)], [(0, 0, (0,) * 6, (1.0, 0.7, 1.3))])
        for pose, Kc in (((0.0, -3.0, 0.0, 0, 0, 0), K_centred), ((0.2, -2.5, 0.1, 0.3, -0.2, 0.1), K)):
            _compare(rt, orc, d, W, H, Kc, scenes.D_REF, pose, gpu_build=gpu_build)
    # a refit that brings NaNs into a mesh uploaded clean: the flag is set by the refit kernels
    so = orc_mod.OracleScene(o)
    so.add_material((0.3, 0.6, 0.9))
    om = o.mesh_from_triangles(clean)
    so.add_mesh(om)
    so.add_instance(0, 0)
    sp = rt.Scene()
    sp.add_material((0.3, 0.6, 0.9))
    sp.add_mesh(rt.Mesh.from_triangles(clean, gpu_build=gpu_build))
    sp.add_mesh_instance(0, 0)
    sp.upload_to_device()
    cam = rt.Camera(W, H, K, scenes.D_REF)
    cam.set_pose((0.0, -3.0, 0.0, 0, 0, 0))
    moved = clean.copy()
    moved[:, 2] += np.float32(0.05); moved[:, 5] += np.float32(0.05); moved[:, 8] += np.float32(0.05)
    moved[40:44, 0:9] = nan
    for i in range(len(moved)):
        moved[i, :12] = o.tri_from_vertices(moved[i, :9])[:12]
    sp.refit_mesh(0, moved)
    o.mesh_refit(om, moved)
    ref = so.render(W, H, K, scenes.D_REF, (0.0, -3.0, 0.0, 0, 0, 0), threads=4)
    dbg = rt.render_debug(sp, cam)
    for n in ("img",) + PLANES:
        assert np.array_equal(dbg[n], ref[n]), ("refit with NaNs", n)
    assert np.array_equal(rt.render(sp, cam), ref["img"])
    # ... and the flag is not sticky: whether the four NaN triangles left a child box unordered or not, a refit back to
    # finite vertices decides it anew -- the mesh is on the octant loops again, and still equal to the oracle
    flagged = sp.mesh_flags(0)
    moved2 = clean.copy()
    moved2[:, 0] += np.float32(0.03); moved2[:, 3] += np.float32(0.03); moved2[:, 6] += np.float32(0.03)
    for i in range(len(moved2)):
        moved2[i, :12] = o.tri_from_vertices(moved2[i, :9])[:12]
    sp.refit_mesh(0, moved2)
    o.mesh_refit(om, moved2)
    assert sp.mesh_flags(0) == 0, flagged
    ref = so.render(W, H, K, scenes.D_REF, (0.0, -3.0, 0.0, 0, 0, 0), threads=4)
    dbg = rt.render_debug(sp, cam)
    for n in ("img",) + PLANES:
        assert np.array_equal(dbg[n], ref[n]), ("clean refit after the NaNs", n)
    # a mesh whose every triangle lacks an axis does leave unordered boxes: flagged at upload and by the refit
    allnan = clean.copy()
    allnan[:, 2] = nan; allnan[:, 5] = nan; allnan[:, 8] = nan
    for i in range(len(allnan)):
        allnan[i, :12] = o.tri_from_vertices(allnan[i, :9])[:12]
    sp.refit_mesh(0, allnan)
    assert sp.mesh_flags(0) == 1
    sp.refit_mesh(0, clean)
    assert sp.mesh_flags(0) == 0
    so.close()


def test_exact_uv_path(rt, orc, scenes):
    """uv values near FLT_MAX switch the kernel to the per-candidate uv test of raycast.cu:96."""
    tris = sd.random_triangles(50, seed=5, spread=0.6, size=0.5)
    tris[::3, 12] = 3.0e38
    tris[1::7, 14] = np.float32(np.finfo(np.float32).max)
    d = sd.SceneDesc([((0.8, 0.8, 0.1), None)], [("tris", tris)], [(0, 0, (0,) * 6, (1, 1, 1))])
    _compare(rt, orc, d, 160, 120, scenes.scaled_K(160), scenes.D_REF, (0.0, -2.5, 0.0, 0, 0, 0))


def test_empty_and_missing(rt, orc, scenes):
    """Empty mesh, and a scene with no instances: all sky."""
    d = sd.SceneDesc([((1, 1, 1), None)], [("tris", np.zeros((0, 18), np.float32))], [(0, 0, (0,) * 6, (1, 1, 1))])
    img, _ = _compare(rt, orc, d, 64, 48, scenes.scaled_K(64), scenes.D_REF, (0, -3, 0, 0, 0, 0))
    assert (img == np.array([255, 204, 153], np.uint8)).all()
    d2 = sd.SceneDesc([((1, 1, 1), None)], [("tris", np.zeros((0, 18), np.float32))], [])
    img, _ = _compare(rt, orc, d2, 64, 48, scenes.scaled_K(64), scenes.D_REF, (0, -3, 0, 0, 0, 0))
    assert (img == np.array([255, 204, 153], np.uint8)).all()


def test_update_mesh_instance(rt, orc, scenes, blob5k):
    """Scene::update_mesh_instance (Scene.cpp:67-74): re-pose one instance, render again."""
    desc = sd.multi_instance_scene(scenes, blob5k)
    m = sd.MULTI_CAMERA
    K = scenes.scaled_K(m["width"])
    so = desc.build_oracle(orc)
    sp = desc.build_product(rt)
    sp.upload_to_device()
    cam = _camera(rt, scenes, m["width"], m["height"], K, m["pose"])
    new_pose, new_scale = (0.4, 0.2, 0.0, -0.3, 0.2, 0.5), (0.9, 0.8, 1.2)
    so.update_instance(0, 0, 2, new_pose, new_scale)
    sp.update_mesh_instance(0, 0, 2, new_pose, new_scale)
    ref = so.render(m["width"], m["height"], K, scenes.D_REF, m["pose"], threads=8)
    dbg = rt.render_debug(sp, cam)
    assert np.array_equal(dbg["img"], ref["img"])
    for n in PLANES:
        assert np.array_equal(dbg[n], ref[n]), n


def test_refit_of_a_deforming_mesh(rt, orc, scenes, blob5k):
    """Scene::refit_mesh / rt_scene_refit_mesh: the triangles of a mesh move (same count, same order), the tree keeps its
    topology and every node gets the exact bounds of its triangles.  All planes must equal the oracle's, which refits the
    same tree on its side (orc_mesh_refit); the deformed frame differs from the rest pose; a second refit back to the rest
    pose restores the original frame; a host-built and a GPU-built mesh behave alike."""
    import orc as orc_mod
    o = orc_mod.oracle()
    W, H = 320, 180
    K, pose = scenes.scaled_K(W), scenes.C2_CAMERAS["mid"]
    tex = sd.checker_texture(32, 32, seed=5)
    for gpu_build in (False, True):
        mesh = rt.Mesh.load_obj(blob5k, gpu_build=gpu_build)
        rest = mesh.dump()["tris"].copy()
        sp = rt.Scene()
        sp.add_material((1.0, 1.0, 1.0), texture_bgr=tex)
        sp.add_mesh(mesh)
        sp.add_mesh_instance(0, 0, (0.1, 0.0, 0.0, 0.2, 0.0, 0.0), (1.0, 0.9, 1.1))
        sp.upload_to_device()
        cam = rt.Camera(W, H, K, scenes.D_REF)
        cam.set_pose(pose)
        so = orc_mod.OracleScene(o)
        so.add_material((1.0, 1.0, 1.0), tex)
        om = o.mesh_from_triangles(rest)
        so.add_mesh(om)
        so.add_instance(0, 0, (0.1, 0.0, 0.0, 0.2, 0.0, 0.0), (1.0, 0.9, 1.1))
        frames = []
        for step in (1, 2, 0):                                  # two deformations, then back to the rest pose
            moved = rest.copy()
            v = moved[:, :9].reshape(-1, 3, 3)
            v[..., 2] += np.float32(0.15 * step) * np.sin(3.0 * v[..., 0] + step)           # a travelling bulge
            v[..., 1] *= np.float32(1.0 + 0.1 * step)
            for i in range(len(moved)):                         # normals as the 3-vertex constructor computes them
                moved[i, :12] = o.tri_from_vertices(moved[i, :9])[:12]
            if step == 2:
                # the deformation arrives in device memory (rt_scene_refit_mesh_device): same result, no host arrays
                hl = rt.libs()[0]
                vert, nrm = np.ascontiguousarray(moved[:, :9]), np.ascontiguousarray(moved[:, 9:12])
                d_v, d_n = rt.DeviceBuffer(nbytes=vert.nbytes), rt.DeviceBuffer(nbytes=nrm.nbytes)
                rt.check(hl.rt_memcpy_h2d(d_v.ptr, vert.ctypes.data, vert.nbytes, None))
                rt.check(hl.rt_memcpy_h2d(d_n.ptr, nrm.ctypes.data, nrm.nbytes, None))
                rt.check(hl.rt_scene_refit_mesh_device(sp.device_handle, 0, d_v.ptr, d_n.ptr, len(moved), None), "rt_scene_refit_mesh_device")
                rt.check(hl.rt_device_synchronize())
                from_device = rt.render_debug(sp, cam)
            sp.refit_mesh(0, moved)
            o.mesh_refit(om, moved)
            ref = so.render(W, H, K, scenes.D_REF, pose, threads=8)
            dbg = rt.render_debug(sp, cam)
            ids = rt.render_ids(sp, cam)
            assert np.array_equal(dbg["img"], ref["img"]) and np.array_equal(ids["img"], ref["img"]), step
            if step == 2:
                for n in ("img",) + PLANES:
                    assert np.array_equal(from_device[n], ref[n]), ("device arrays", n)
            for n in PLANES:
                assert np.array_equal(dbg[n], ref[n]), (n, step)
            assert np.array_equal(ids["hit_tri"], ref["hit_tri"])
            frames.append(ref["img"])
            if step == 2:                                       # a re-upload sends the host copy, which was refitted too
                sp.upload_to_device()
                again = rt.render_debug(sp, cam)
                assert np.array_equal(again["img"], ref["img"]) and np.array_equal(again["pops"], ref["pops"])
        assert not np.array_equal(frames[0], frames[2]) and not np.array_equal(frames[0], frames[1])
        so.close()
    h = rt.libs()[0]
    import ctypes as C
    assert h.rt_scene_refit_mesh(sp.device_handle, 3, rest.ctypes.data_as(C.POINTER(C.c_float)), rest.ctypes.data_as(C.POINTER(C.c_float)), len(rest), None) == -1
    assert h.rt_scene_refit_mesh(sp.device_handle, 0, rest.ctypes.data_as(C.POINTER(C.c_float)), rest.ctypes.data_as(C.POINTER(C.c_float)), len(rest) - 1, None) == -1


def test_animated_instance_without_host_waits(rt, orc, scenes, blob5k):
    """The animation loop of kernel.cu:272-277 (re-pose an instance, render, repeat) issued on one stream with no
    synchronisation in between: update k must be seen by render k and by no earlier one (rt_scene_update_instance_async)."""
    desc = sd.multi_instance_scene(scenes, blob5k)
    m = sd.MULTI_CAMERA
    W, H, K = m["width"], m["height"], scenes.scaled_K(m["width"])
    so = desc.build_oracle(orc)
    sp = desc.build_product(rt)
    sp.upload_to_device()
    cam = _camera(rt, scenes, W, H, K, m["pose"])
    nframes = 6
    poses = [(0.4 - 0.1 * k, 0.2, 0.05 * k, -0.3 + 0.2 * k, 0.2, 0.5) for k in range(nframes)]
    imgs = [rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(nframes)]
    for k in range(nframes):
        sp.update_mesh_instance(0, 0, 2, poses[k], (0.9, 0.8, 1.2), stream=None)      # ordered on the (default) stream
        cam.render_scene(sp, imgs[k].ptr, imgs[k].pitch)                                # asynchronous
    rt.check(rt.libs()[0].rt_device_synchronize())
    frames = [b.to_host().reshape(H, W, 3) for b in imgs]
    for k in range(nframes):
        so.update_instance(0, 0, 2, poses[k], (0.9, 0.8, 1.2))
        ref = so.render(W, H, K, scenes.D_REF, m["pose"], threads=8, planes=False)["img"]
        assert np.array_equal(frames[k], ref), "frame %d" % k
    assert not np.array_equal(frames[0], frames[1])


def test_stripes_equal_full_frame(rt, orc, scenes, blob5k):
    """Frame tiling: for 1, 2, 3, 4, 8 virtual ranks the un-striped gather equals the 1-GPU frame byte for byte."""
    import ctypes as C
    h = rt.libs()[0]
    W, H = 322, 203
    sp = sd.blob_scene(scenes, blob5k).build_product(rt)
    sp.upload_to_device()
    cam = _camera(rt, scenes, W, H, scenes.scaled_K(W), scenes.C2_CAMERAS["mid"])
    full = rt.render(sp, cam)
    for nr, stripe in [(1, 16), (2, 16), (3, 8), (4, 16), (8, 16), (8, 7)]:
        rows = []
        for r in range(nr):
            n = C.c_int32(0)
            rt.check(h.rt_stripe_rows(H, stripe, r, nr, C.byref(n)))
            rows.append(n.value)
        assert sum(rows) == H
        maxr = max(rows)
        pitch = W * 3
        gathered = rt.DeviceBuffer(nbytes=nr * maxr * pitch)
        for r in range(nr):
            cam.render_scene_stripes(sp, gathered.ptr.value + r * maxr * pitch, pitch, stripe, r, nr, synchronize=True)
        out = rt.DeviceBuffer(width_bytes=W * 3, height=H)
        rt.check(h.rt_unstripe(gathered.ptr, pitch, maxr * pitch, out.ptr, out.pitch, W, H, stripe, nr, None))
        rt.check(h.rt_device_synchronize())
        got = out.to_host().reshape(H, W, 3)
        assert np.array_equal(got, full), "stripes nr=%d stripe=%d" % (nr, stripe)
        gathered.free()
        out.free()


def test_batched_frames_equal_single_frames(rt, orc, scenes, blob5k):
    """rt_render_batch: a camera path of 1..8 frames in one launch == the same frames rendered one by one
    (and frame 0 == the oracle)."""
    W, H = 200, 120
    desc = sd.blob_scene(scenes, blob5k)
    sp = desc.build_product(rt)
    sp.upload_to_device()
    K = scenes.scaled_K(W)
    cam = rt.Camera(W, H, K, scenes.D_REF)
    poses = [(0.05 * i, -1.5 - 0.15 * i, 0.2 + 0.02 * i, 0.02 * i, -0.01 * i, 0.0) for i in range(8)]
    singles = []
    for ps in poses:
        cam.set_pose(ps)
        singles.append(rt.render(sp, cam))
    so = desc.build_oracle(orc)
    assert np.array_equal(singles[0], so.render(W, H, K, scenes.D_REF, poses[0], planes=False)["img"])
    for n in (1, 3, 8):
        bufs = [rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(n)]
        cam.render_scene_batch(sp, poses[:n], [b.ptr for b in bufs], bufs[0].pitch, synchronize=True)
        for i, b in enumerate(bufs):
            assert np.array_equal(b.to_host().reshape(H, W, 3), singles[i]), "batch %d frame %d" % (n, i)
            b.free()
    with pytest.raises(rt.RtError):
        cam.render_scene_batch(sp, poses * 5, [1] * 40, W * 3)              # more than RT_MAX_BATCH (32) frames
    bufs = [rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(32)]
    cam.render_scene_batch(sp, poses * 4, [b.ptr for b in bufs], bufs[0].pitch, synchronize=True)
    for i in (0, 13, 31):
        assert np.array_equal(bufs[i].to_host().reshape(H, W, 3), singles[i % 8])
    for b in bufs:
        b.free()


def test_view_records(rt, orc, scenes, blob5k):
    """Round 5: launches of four and more frames render through VIEW records -- per frame and instance the interior records with
    `box - origin` in place of the boxes, written once by a pre-pass into a pool behind the scene's records (rt_scene_view_stats).
    Frames must not change: batches against single-frame launches (which take no views) and the oracle; the pool grows from 4 to
    16 frames per slot (the record array moves; round 6: to the batch sizes seen, not straight to 32); an instance moved between two batches; a mesh refitted after the array has moved;
    more launches in flight on different streams than the pool has slots (the extra ones render without views)."""
    import torch
    W, H = 400, 240                             # (a frame must bring eight rays per view record: 10 297 records in the second scene)
    K = scenes.scaled_K(W)
    poses = [(0.05 * i, -1.5 - 0.1 * i, 0.2 + 0.02 * i, 0.02 * i, -0.01 * i, 0.0) for i in range(32)]
    for desc in (sd.blob_scene(scenes, blob5k), sd.multi_instance_scene(scenes, blob5k)):
        sp = desc.build_product(rt)
        sp.upload_to_device()
        cam = rt.Camera(W, H, K, scenes.D_REF)
        singles = []
        for ps in poses[:9]:
            cam.set_pose(ps)
            singles.append(rt.render_ids(sp, cam)["img"])                # one frame per launch: no views
        assert sp.view_stats()["launches"] == 0
        so = desc.build_oracle(orc)
        for k in (0, 5):
            assert np.array_equal(singles[k], so.render(W, H, K, scenes.D_REF, poses[k], planes=False)["img"])
        so.close()
        bufs = [rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(32)]
        for n, slot_frames, grows in ((4, 4, 1), (3, 4, 1), (9, 16, 2), (4, 16, 2)):
            cam.render_scene_batch(sp, poses[:n], [b.ptr for b in bufs[:n]], bufs[0].pitch, synchronize=True)
            st = sp.view_stats()
            if n >= 4:
                assert (st["slot_frames"], st["grows"], st["fallbacks"]) == (slot_frames, grows, 0), (n, st)
            for i in range(n):
                assert np.array_equal(bufs[i].to_host().reshape(H, W, 3), singles[i]), "batch of %d, frame %d" % (n, i)
        assert sp.view_stats()["launches"] == 3
        # an instance moved between two batches: the views are of the pose at launch time
        mesh0, mat0, pose0, scale0 = desc.instances[0]
        sp.update_mesh_instance(0, mesh0, mat0, (0.3, 0.2, -0.1, 0.2, 0.0, 0.1), scale0)
        cam.set_pose(poses[2])
        moved = rt.render_debug(sp, cam)["img"]
        cam.render_scene_batch(sp, [poses[2]] * 4, [b.ptr for b in bufs[:4]], bufs[0].pitch, synchronize=True)
        assert np.array_equal(bufs[3].to_host().reshape(H, W, 3), moved) and not np.array_equal(moved, singles[2])
        sp.update_mesh_instance(0, mesh0, mat0, pose0, scale0)
        # six streams, three slots: some launches find no slot and render without views -- every frame is still right
        streams = [torch.cuda.Stream() for _ in range(6)]
        cams = [rt.Camera(W, H, K, scenes.D_REF) for _ in streams]
        sets = [[rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(4)] for _ in streams]
        for rep in range(3):
            for c, st_, bs in zip(cams, streams, sets):
                c.set_stream(st_.cuda_stream)
                c.render_scene_batch(sp, poses[rep:rep + 4], [b.ptr for b in bs], bs[0].pitch)
            torch.cuda.synchronize()
            for bs in sets:
                for i, b in enumerate(bs):
                    assert np.array_equal(b.to_host().reshape(H, W, 3), singles[rep + i]), (rep, i)
        assert sp.view_stats()["launches"] == 4 + 18
        for bs in sets:
            for b in bs:
                b.free()
        for b in bufs:
            b.free()
    # a refit after the record array has moved behind the pool (single mesh: the blob, squeezed and back)
    import orc as orc_mod
    o = orc_mod.oracle()
    mesh = rt.Mesh.load_obj(blob5k)
    tris = mesh.dump()["tris"].copy()
    sp = rt.Scene()
    sp.add_material((0.9, 0.5, 0.2))
    sp.add_mesh(mesh)
    sp.add_mesh_instance(0, 0)
    sp.upload_to_device()
    cam = rt.Camera(W, H, K, scenes.D_REF)
    bufs = [rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(4)]
    cam.render_scene_batch(sp, poses[:4], [b.ptr for b in bufs], bufs[0].pitch, synchronize=True)
    tris[:, :9].reshape(-1, 3, 3)[..., 0] *= np.float32(0.7)
    for i in range(len(tris)):
        tris[i, :12] = o.tri_from_vertices(tris[i, :9])[:12]
    sp.refit_mesh(0, tris)
    cam.set_pose(poses[1])
    want = rt.render_debug(sp, cam)["img"]
    cam.render_scene_batch(sp, poses[:4], [b.ptr for b in bufs], bufs[0].pitch, synchronize=True)
    assert np.array_equal(bufs[1].to_host().reshape(H, W, 3), want)
    assert sp.view_stats()["launches"] == 2 and sp.view_stats()["fallbacks"] == 0


def test_view_pool_reserved_bounded_and_accounted(rt, scenes, blob5k):
    """Round 6: rt_scene_reserve_views sizes the pool once, after upload -- launches then never grow it (no render
    call blocks), a launch of more frames than reserved renders without views; handing sizing back lets the pool grow, and the block
    it leaves is freed: the scene's device bytes are records + pool again.  Two streams batch while a third renders single frames."""
    import torch
    W, H = 400, 240
    K = scenes.scaled_K(W)
    poses = [(0.05 * i, -1.5 - 0.1 * i, 0.2 + 0.02 * i, 0.02 * i, -0.01 * i, 0.0) for i in range(16)]
    sp = sd.blob_scene(scenes, blob5k).build_product(rt)
    sp.upload_to_device()
    cam = rt.Camera(W, H, K, scenes.D_REF)
    singles = []
    for ps in poses:
        cam.set_pose(ps)
        singles.append(rt.render_ids(sp, cam)["img"])
    m0 = sp.memory()
    assert m0["view_pool_bytes"] == 0 and m0["view_slots"] == 0 and m0["device_bytes"] == sp.info()["device_bytes"]
    sp.reserve_views(8)
    m1 = sp.memory()
    frame_bytes = m1["view_pool_bytes"] // (3 * 8)
    assert (m1["view_slots"], m1["view_slot_frames"]) == (3, 8) and frame_bytes * 24 == m1["view_pool_bytes"] and frame_bytes % 64 == 0
    assert 0 <= m1["device_bytes"] - m0["device_bytes"] - m1["view_pool_bytes"] < 256, (m0, m1)          # (the pool starts on a 256-byte boundary)
    assert sp.view_stats()["grows"] == 1
    streams = [torch.cuda.Stream() for _ in range(3)]
    cams = [rt.Camera(W, H, K, scenes.D_REF) for _ in streams]
    for c, st in zip(cams, streams):
        c.set_stream(st.cuda_stream)
    sets = [[rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(8)] for _ in range(2)]
    one = rt.DeviceBuffer(width_bytes=W * 3, height=H)
    batches = 0
    for rep in range(6):
        for k in range(2):
            cams[k].render_scene_batch(sp, poses[rep:rep + 8], [b.ptr for b in sets[k]], sets[k][0].pitch)
            batches += 1
        for j in range(3):                                      # single frames on the third stream, beside the batches
            cams[2].set_pose(poses[rep + j])
            cams[2].render_scene(sp, one.ptr, one.pitch)
        torch.cuda.synchronize()
        assert np.array_equal(one.to_host().reshape(H, W, 3), singles[rep + 2])
        for bs in sets:
            for i, b in enumerate(bs):
                assert np.array_equal(b.to_host().reshape(H, W, 3), singles[rep + i]), (rep, i)
    st = sp.view_stats()
    assert (st["launches"], st["fallbacks"], st["grows"], st["slot_frames"]) == (batches, 0, 1, 8), st
    # more frames than reserved: no growth, no views, the right frames
    nine = [rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(9)]
    cam.render_scene_batch(sp, poses[:9], [b.ptr for b in nine], nine[0].pitch, synchronize=True)
    st = sp.view_stats()
    assert (st["fallbacks"], st["grows"], st["slot_frames"]) == (1, 1, 8), st
    assert all(np.array_equal(b.to_host().reshape(H, W, 3), singles[i]) for i, b in enumerate(nine))
    # sizing handed back to the launches: the pool grows to 16 frames per slot and the block it leaves is freed
    sp.reserve_views(0)
    cam.render_scene_batch(sp, poses[:9], [b.ptr for b in nine], nine[0].pitch, synchronize=True)
    m2 = sp.memory()
    assert (m2["view_slots"], m2["view_slot_frames"], m2["view_pool_bytes"]) == (3, 16, frame_bytes * 48), m2
    assert 0 <= m2["device_bytes"] - m0["device_bytes"] - m2["view_pool_bytes"] < 256, (m0, m2)
    assert sp.view_stats()["grows"] == 2 and sp.info()["device_bytes"] == m2["device_bytes"]
    assert all(np.array_equal(b.to_host().reshape(H, W, 3), singles[i]) for i, b in enumerate(nine))
    h = rt.libs()[0]
    assert h.rt_scene_reserve_views(sp.device_handle, 33) == -1 and h.rt_scene_reserve_views(None, 4) == -1
    for b in nine + sets[0] + sets[1] + [one]:
        b.free()


def test_view_pool_budget_in_a_child_process():
    """RT_VIEW_MAX_BYTES (read once per process): a pool of three slots that does not fit drops to two; a reservation beyond the budget is refused."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys, importlib, numpy as np; sys.path.insert(0, 'tests'); import scene_defs as sd\n"
            "rt = importlib.import_module('cuda-raytracing_amd'); scenes = importlib.import_module('cuda-raytracing_amd.scenes'); rt.build()\n"
            "blob = os.path.join('.scene_cache', 'blob5k.obj')\n"
            "os.makedirs('.scene_cache', exist_ok=True)\n"
            "if not os.path.exists(blob): scenes.write_blob_obj(blob, 50, 51)\n"
            "W, H = 400, 240; K = scenes.scaled_K(W); sp = sd.blob_scene(scenes, blob).build_product(rt); sp.upload_to_device()\n"
            "poses = [(0.05 * i, -1.5 - 0.1 * i, 0.2, 0.02 * i, 0.0, 0.0) for i in range(4)]\n"
            "cam = rt.Camera(W, H, K, scenes.D_REF); bufs = [rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(4)]\n"
            "cam.render_scene_batch(sp, poses, [b.ptr for b in bufs], bufs[0].pitch, synchronize=True)\n"
            "m = sp.memory(); print(m); assert (m['view_slots'], m['view_slot_frames']) == (2, 4) and m['view_pool_bytes'] <= 3000000, m\n"
            "cam.set_pose(poses[3]); assert np.array_equal(bufs[3].to_host().reshape(H, W, 3), rt.render_ids(sp, cam)['img'])\n"
            "assert rt.libs()[0].rt_scene_reserve_views(sp.device_handle, 32) == -2\n"
            "assert sp.memory() == m; print('budget ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, RT_VIEW_MAX_BYTES="3000000"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "budget ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_view_records_switched_off_in_a_child_process():
    """RT_VIEW_RECORDS=0 (read once per process): the batch launches of the smoke scenes without view records, against the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys, importlib, numpy as np; sys.path.insert(0, 'tests'); import orc, scene_defs as sd\n"
            "rt = importlib.import_module('cuda-raytracing_amd'); scenes = importlib.import_module('cuda-raytracing_amd.scenes'); rt.build(); orc.build_oracle()\n"
            "blob = os.path.join('.scene_cache', 'blob5k.obj')\n"
            "os.makedirs('.scene_cache', exist_ok=True)\n"
            "if not os.path.exists(blob): scenes.write_blob_obj(blob, 50, 51)\n"
            "W, H = 320, 200; K = scenes.scaled_K(W); desc = sd.blob_scene(scenes, blob); sp = desc.build_product(rt); sp.upload_to_device()\n"
            "poses = [(0.05 * i, -1.5 - 0.1 * i, 0.2, 0.02 * i, 0.0, 0.0) for i in range(4)]\n"
            "cam = rt.Camera(W, H, K, scenes.D_REF); bufs = [rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(4)]\n"
            "cam.render_scene_batch(sp, poses, [b.ptr for b in bufs], bufs[0].pitch, synchronize=True)\n"
            "so = desc.build_oracle(orc)\n"
            "assert np.array_equal(bufs[3].to_host().reshape(H, W, 3), so.render(W, H, K, scenes.D_REF, poses[3], planes=False)['img'])\n"
            "assert sp.view_stats()['launches'] == 0; print('no views ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, RT_VIEW_RECORDS="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "no views ok" in r.stdout, r.stderr[-2000:]


def test_scene_upload_builds_the_tree_on_the_device(rt, orc, scenes, blob5k, tmp_path):
    """A mesh that arrives without a tree (MeshPrimitive::for_device_build / OBJLoader::load_for_device; RtMeshDesc.num_nodes = 0):
    rt_scene_upload reserves its part of the arrays and the GPU builds the reference's tree in place.  The device arrays are byte
    for byte those of a scene whose mesh was built first (GPU build -> host arrays -> upload), all planes equal the oracle's, a
    second mesh with a host-built tree lives next to it, and refit / rebuild / re-upload work afterwards."""
    import orc as orc_mod
    o = orc_mod.oracle()
    W, H = 320, 180
    K, pose = scenes.scaled_K(W), scenes.C2_CAMERAS["mid"]
    soup = sd.random_triangles(300, seed=8, spread=0.6, size=0.3)
    inst0, inst1 = ((0.1, 0.0, 0.0, 0.2, 0.0, 0.0), (1.0, 0.9, 1.1)), ((1.2, 0.5, 0.3, 0.0, 0.4, 0.0), (0.8, 0.8, 0.8))

    def build(deferred):
        sp = rt.Scene()
        sp.add_material((0.9, 0.5, 0.2))
        sp.add_mesh(rt.Mesh.load_obj(blob5k, for_device=True) if deferred else rt.Mesh.load_obj(blob5k, gpu_build=True))
        sp.add_mesh(rt.Mesh.from_triangles(soup))                # a host-built tree in the same scene
        sp.add_mesh(rt.Mesh.from_triangles(soup[:1], for_device=deferred, gpu_build=not deferred))      # one triangle
        sp.add_mesh_instance(0, 0, *inst0)
        sp.add_mesh_instance(1, 0, *inst1)
        sp.add_mesh_instance(2, 0, (0, 0, 1.5, 0, 0, 0), (1, 1, 1))
        sp.upload_to_device()
        return sp

    a, b = build(True), build(False)
    for which, dt in ((0, np.uint32), (1, np.uint32), (2, np.int32), (3, np.int32), (4, np.uint32)):
        x, y = a.debug_read(which, dt), b.debug_read(which, dt)
        assert x.shape == y.shape and np.array_equal(x, y), ("array", which)
    assert a.info() == b.info()
    so = orc_mod.OracleScene(o)
    so.add_material((0.9, 0.5, 0.2))
    blob = o.obj_load(blob5k)
    so.add_mesh(blob)
    so.add_mesh(o.mesh_from_triangles(soup))
    so.add_mesh(o.mesh_from_triangles(soup[:1]))
    so.add_instance(0, 0, *inst0)
    so.add_instance(1, 0, *inst1)
    so.add_instance(2, 0, (0, 0, 1.5, 0, 0, 0), (1, 1, 1))
    cam = rt.Camera(W, H, K, scenes.D_REF)
    cam.set_pose(pose)
    ref = so.render(W, H, K, scenes.D_REF, pose, threads=8)
    for sp in (a, b):
        dbg = rt.render_debug(sp, cam)
        for n in ("img",) + PLANES:
            assert np.array_equal(dbg[n], ref[n]), n
    # the deferred mesh deforms (refit), then changes (rebuild), then the scene is uploaded again
    tris = o.mesh_dump(blob)["tris"].copy()
    tris[:, :9].reshape(-1, 3, 3)[..., 2] *= np.float32(1.1)
    for i in range(len(tris)):
        tris[i, :12] = o.tri_from_vertices(tris[i, :9])[:12]
    a.refit_mesh(0, tris)
    o.mesh_refit(blob, tris)
    ref = so.render(W, H, K, scenes.D_REF, pose, threads=8)
    dbg = rt.render_debug(a, cam)
    for n in ("img",) + PLANES:
        assert np.array_equal(dbg[n], ref[n]), ("refit", n)
    # An upload never changes the tree a mesh is rendered with (round 6, third session; found by test_fuzz_adversarial_api_sequences): the
    # deferred mesh was refitted since its device build, so its host tree -- built now, at last -- is the tree of that build refitted, not
    # a new one over the moved triangles (which this test used to expect): the frame and every count stay what they were before the upload
    a.upload_to_device()
    dbg = rt.render_debug(a, cam)
    for n in ("img",) + PLANES:
        assert np.array_equal(dbg[n], ref[n]), ("re-upload", n)
    fresh = orc_mod.OracleScene(o)                              # (a new tree over the moved triangles would have shown: its visit counts differ)
    fresh.add_material((0.9, 0.5, 0.2))
    for m in (o.mesh_from_triangles(tris), o.mesh_from_triangles(soup), o.mesh_from_triangles(soup[:1])):
        fresh.add_mesh(m)
    fresh.add_instance(0, 0, *inst0)
    fresh.add_instance(1, 0, *inst1)
    fresh.add_instance(2, 0, (0, 0, 1.5, 0, 0, 0), (1, 1, 1))
    assert not np.array_equal(fresh.render(W, H, K, scenes.D_REF, pose, threads=8)["pops"], ref["pops"])
    fresh.close()
    # a mesh in the middle of the record array (index 1, host-built at upload) is rebuilt in place; its new triangles carry a uv
    # value that is not an ordinary number, which switches the mesh to per-candidate uv interpolation (raycast.cu:96)
    soup2 = sd.random_triangles(300, seed=9, spread=0.7, size=0.35)
    soup2[5, 12] = np.inf
    a.rebuild_mesh(1, soup2)
    so3 = orc_mod.OracleScene(o)
    so3.add_material((0.9, 0.5, 0.2))
    so3.add_mesh(blob)                                          # (mesh 0 as it is rendered: the first tree, refitted)
    so3.add_mesh(o.mesh_from_triangles(soup2))
    so3.add_mesh(o.mesh_from_triangles(soup[:1]))
    so3.add_instance(0, 0, *inst0)
    so3.add_instance(1, 0, *inst1)
    so3.add_instance(2, 0, (0, 0, 1.5, 0, 0, 0), (1, 1, 1))
    ref = so3.render(W, H, K, scenes.D_REF, pose, threads=8)
    dbg = rt.render_debug(a, cam)
    for n in ("img",) + PLANES:
        assert np.array_equal(dbg[n], ref[n]), ("rebuild of mesh 1", n)
    so.close()
    so3.close()


def test_device_resident_rebuild_of_a_mesh(rt, orc, scenes, blob5k):
    """Scene::rebuild_mesh / rt_scene_rebuild_mesh_device: the triangles of an uploaded mesh are replaced and the device copy
    gets a NEW tree, built by the GPU build kernels and emitted straight into the scene's record arrays.  (a) the arrays are,
    byte for byte, what the host route produces for the same triangles (GPU build -> host arrays -> Scene::upload_to_device
    -> rt_scene_upload); (b) all planes equal the oracle's, whose builder is the reference's; (c) a refit, an instance update
    and a re-upload after the rebuild behave; (d) fewer triangles than at upload, a single leaf, one triangle, none."""
    import orc as orc_mod
    o = orc_mod.oracle()
    W, H = 320, 180
    K, pose = scenes.scaled_K(W), scenes.C2_CAMERAS["mid"]
    tex = sd.checker_texture(32, 32, seed=5)
    inst = ((0.1, 0.0, 0.0, 0.2, 0.0, 0.0), (1.0, 0.9, 1.1))
    rest = rt.Mesh.load_obj(blob5k).dump()["tris"].copy()

    def with_normals(t):
        t = t.copy()
        for i in range(len(t)):
            t[i, :12] = o.tri_from_vertices(t[i, :9])[:12]
        return t

    def scrambled(step):                                        # far more than a refit can follow: the blob folded and stretched
        m = rest.copy()
        v = m[:, :9].reshape(-1, 3, 3)
        v[..., 2] += np.float32(0.6 * step) * np.sin(5.0 * v[..., 0] + step)
        v[..., 0] = np.where(v[..., 1] > 0, v[..., 0], -v[..., 0] * np.float32(1.3))
        return with_normals(m)

    def oracle_scene(tris):
        so = orc_mod.OracleScene(o)
        so.add_material((1.0, 1.0, 1.0), tex)
        om = o.mesh_from_triangles(tris)
        so.add_mesh(om)
        so.add_instance(0, 0, *inst)
        return so, om

    def product_scene(tris):
        sp = rt.Scene()
        sp.add_material((1.0, 1.0, 1.0), texture_bgr=tex)
        sp.add_mesh(rt.Mesh.from_triangles(tris, gpu_build=True))
        sp.add_mesh_instance(0, 0, *inst)
        sp.upload_to_device()
        return sp

    def check(sp, so, what):
        cam = rt.Camera(W, H, K, scenes.D_REF)
        cam.set_pose(pose)
        ref = so.render(W, H, K, scenes.D_REF, pose, threads=8)
        dbg = rt.render_debug(sp, cam)
        for n in ("img",) + PLANES:
            assert np.array_equal(dbg[n], ref[n]), (what, n)
        assert np.array_equal(rt.render_ids(sp, cam)["hit_tri"], ref["hit_tri"]), what
        return ref["img"]

    sp = product_scene(rest)
    first = None
    for step in (1, 2):
        moved = scrambled(step)
        sp.rebuild_mesh(0, moved)
        fresh = product_scene(moved)                            # the host route for the same triangles
        for which, dt in ((0, np.uint32), (1, np.uint32), (2, np.int32), (3, np.int32), (4, np.uint32)):
            a, b = sp.debug_read(which, dt), fresh.debug_read(which, dt)
            assert a.shape == b.shape and np.array_equal(a, b), ("array", which, step)
        so, om = oracle_scene(moved)
        img = check(sp, so, "rebuilt %d" % step)
        first = img if first is None else first
        # a small deformation afterwards is a refit of the NEW tree
        m2 = moved.copy()
        m2[:, :9].reshape(-1, 3, 3)[..., 2] += np.float32(0.02) * np.cos(4.0 * m2[:, :9].reshape(-1, 3, 3)[..., 1])
        m2 = with_normals(m2)
        sp.refit_mesh(0, m2)
        o.mesh_refit(om, m2)
        check(sp, so, "refit after rebuild %d" % step)
        sp.update_mesh_instance(0, 0, 0, (0.0, 0.1, 0.0, 0.1, 0.0, 0.0), (1.0, 1.0, 1.0))
        so.update_instance(0, 0, 0, (0.0, 0.1, 0.0, 0.1, 0.0, 0.0), (1.0, 1.0, 1.0))
        check(sp, so, "instance update after rebuild %d" % step)
        sp.update_mesh_instance(0, 0, 0, *inst)
        so.close()
    # a re-upload sends the host copy, whose tree is built when it is needed: the tree of the last rebuild, refitted as the mesh was since
    so, om = oracle_scene(moved)
    o.mesh_refit(om, m2)
    sp.upload_to_device()
    check(sp, so, "re-upload")
    so.close()
    # fewer triangles; one leaf that cannot be split (duplicates); one triangle; none
    for name, tris in (("a third", scrambled(1)[::3]), ("duplicates", np.repeat(rest[:1], 40, 0)), ("one", rest[7:8]), ("none", rest[:0])):
        sp.rebuild_mesh(0, tris)
        so, om = oracle_scene(tris)
        check(sp, so, name)
        so.close()
    # more triangles than the mesh was uploaded with do not fit its part of the device arrays: the C-ABI call refuses them
    # untouched, and Scene::rebuild_mesh uploads the scene again -- host and device describe the same mesh afterwards
    # (the host mesh used to change first and a refused device call left the two apart)
    bigger = np.concatenate([rest, scrambled(2)[:10]])
    h = rt.libs()[0]
    cap = C.c_int32(0)
    assert h.rt_scene_mesh_capacity(sp.device_handle, 0, C.byref(cap)) == 0 and cap.value == len(rest)
    dummy = rt.DeviceBuffer(width_bytes=64, height=1)           # (never read: the count is refused first)
    assert h.rt_scene_rebuild_mesh_device(sp.device_handle, 0, dummy.ptr, dummy.ptr, None, len(bigger), None) == -1      # RT_E_INVALID
    sp.rebuild_mesh(0, rest)
    sp.rebuild_mesh(0, bigger)
    so, om = oracle_scene(bigger)
    check(sp, so, "more triangles than at upload")
    assert h.rt_scene_mesh_capacity(sp.device_handle, 0, C.byref(cap)) == 0 and cap.value == len(bigger)
    b2 = bigger.copy()
    b2[:, :9].reshape(-1, 3, 3)[..., 0] += np.float32(0.01)
    b2 = with_normals(b2)
    sp.refit_mesh(0, b2)                                        # (the count the host holds is the count the device holds)
    o.mesh_refit(om, b2)
    check(sp, so, "refit after growing")
    so.close()
    sp.rebuild_mesh(0, rest)                                    # and back: the first frame again
    so, om = oracle_scene(rest)
    check(sp, so, "back to rest")
    # A re-upload that FAILS (the growing path used to replace the host mesh and destroy the device scene before
    # rt_scene_upload had run): host and device must still describe the old mesh -- it still renders, a refit with the OLD
    # triangle count is still accepted, and a later attempt with the same triangles succeeds.
    cap_before = C.c_int32(0)
    assert h.rt_scene_mesh_capacity(sp.device_handle, 0, C.byref(cap_before)) == 0
    even_bigger = np.concatenate([bigger, scrambled(1)[:500]])
    assert len(even_bigger) > cap_before.value
    handle_before = sp.device_handle
    os.environ["RT_TEST_FAIL_UPLOAD"] = "1"
    try:
        with pytest.raises(rt.RtError):
            sp.rebuild_mesh(0, even_bigger)
    finally:
        del os.environ["RT_TEST_FAIL_UPLOAD"]
    assert sp.device_handle == handle_before                    # the old device scene is still the scene
    check(sp, so, "after a failed re-upload")
    sp.refit_mesh(0, rest)                                      # the host still holds the old mesh: its count is accepted
    check(sp, so, "refit after a failed re-upload")
    so.close()
    sp.rebuild_mesh(0, even_bigger)                             # the same request without the injected failure
    so, om = oracle_scene(even_bigger)
    check(sp, so, "grown after the failed attempt")
    assert h.rt_scene_mesh_capacity(sp.device_handle, 0, C.byref(cap)) == 0 and cap.value == len(even_bigger)
    # the grown mesh is still a device-built one: growing it again needs no host tree (a re-upload through upload_to_device works too)
    sp.upload_to_device()
    check(sp, so, "re-upload of the grown mesh")
    so.close()


def test_heavy_first_tile_order_keeps_frames_identical(rt, scenes, blob70k):
    """Single-frame launches dispatch their tiles heaviest-first, in an order sorted from an earlier frame's per-tile costs
    on a side stream (rt_kernels.hip, launch_ordered).  Whatever order a launch happens to read -- none yet, one frame
    old, several frames old, sorted for another camera -- every frame must equal the frame the natural-order kernel
    renders (rt_render_ids does not use the order)."""
    W, H = 1920, 1080
    sp = sd.blob_scene(scenes, blob70k).build_product(rt)
    sp.upload_to_device()
    cam = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF)
    img = rt.DeviceBuffer(width_bytes=W * 3, height=H)
    poses = [scenes.C2_CAMERAS["far"]] * 3 + [scenes.C2_CAMERAS["near"], scenes.C2_CAMERAS["mid"]] + \
            [(0.1 * k, -1.5 - 0.1 * k, 0.2, 0.05 * k, 0.0, 0.02 * k) for k in range(6)]
    want = {}
    for k, pose in enumerate(poses):
        cam.set_pose(pose)
        if pose not in want:
            want[pose] = rt.render_ids(sp, cam)["img"]
        # odd frames are issued back to back with the previous one (no host wait in between: the sort of the previous
        # frame is then usually still pending), even ones after a synchronise
        cam.render_scene(sp, img.ptr, img.pitch, synchronize=(k % 2 == 0))
        rt.check(rt.libs()[0].rt_device_synchronize())
        assert np.array_equal(img.to_host().reshape(H, W, 3), want[pose]), "frame %d" % k
    # a rank's stripes of a batch of frames take the same path (all frames' costly tiles first), here on two streams
    import importlib
    import torch
    tiling = importlib.import_module("cuda-raytracing_amd.tiling")
    h = rt.libs()[0]
    F, world, stripe, pitch = 4, 8, 16, W * 3
    bposes = poses[-F:]
    max_rows = max(tiling.stripe_rows(H, stripe, r, world) for r in range(world))
    gathered = rt.DeviceBuffer(nbytes=world * F * max_rows * pitch)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    cams = []
    for st in streams:
        c = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF)
        c.set_stream(st.cuda_stream)
        cams.append(c)
    for rep in range(3):
        for r in range(world):
            base = gathered.ptr.value + r * F * max_rows * pitch
            cams[r & 1].render_scene_stripes_batch(sp, bposes, tiling.batch_local_ptrs(base, F, max_rows, pitch), pitch, stripe, r, world)
        torch.cuda.synchronize()
        outs = rt.DeviceBuffer(nbytes=F * H * pitch)
        rt.check(h.rt_unstripe_batch(gathered.ptr, pitch, F * max_rows * pitch, max_rows * pitch, outs.ptr, pitch, H * pitch, F, W, H, stripe, world, None))
        rt.check(h.rt_device_synchronize())
        got = outs.to_host().reshape(F, H, W, 3)
        for f in range(F):
            assert np.array_equal(got[f], want[bposes[f]]), "stripes batch rep %d frame %d" % (rep, f)
        outs.free()
    gathered.free()
    # another frame size on the same scene: the order state starts over
    cam2 = rt.Camera(1280, 720, scenes.scaled_K(1280), scenes.D_REF)
    cam2.set_pose(scenes.C2_CAMERAS["mid"])
    ref2 = rt.render_ids(sp, cam2)["img"]
    for _ in range(3):
        assert np.array_equal(rt.render(sp, cam2), ref2)
    cam.set_pose(scenes.C2_CAMERAS["mid"])
    assert np.array_equal(rt.render(sp, cam), want[scenes.C2_CAMERAS["mid"]])


def test_batched_stripes_layout(rt, scenes, blob5k):
    """The buffer layout bench.py uses for N ranks x F frames per gather: every virtual rank renders its F stripe
    buffers straight into its slice of the gathered buffer, then each frame is un-striped and compared."""
    import importlib
    tiling = importlib.import_module("cuda-raytracing_amd.tiling")
    h = rt.libs()[0]
    W, H, F, stripe = 210, 133, 3, 16
    sp = sd.blob_scene(scenes, blob5k).build_product(rt)
    sp.upload_to_device()
    cam = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF)
    poses = [(0.03 * i, -1.6 - 0.2 * i, 0.2, 0.0, 0.01 * i, 0.0) for i in range(F)]
    full = []
    for ps in poses:
        cam.set_pose(ps)
        full.append(rt.render(sp, cam))
    pitch = W * 3
    for world in (2, 8):
        max_rows = max(tiling.stripe_rows(H, stripe, r, world) for r in range(world))
        gathered = rt.DeviceBuffer(nbytes=world * F * max_rows * pitch)
        g0 = gathered.ptr.value
        for r in range(world):
            local_base = g0 + r * F * max_rows * pitch          # what the gather would place there
            cam.render_scene_stripes_batch(sp, poses, tiling.batch_local_ptrs(local_base, F, max_rows, pitch), pitch, stripe, r, world,
                                           synchronize=True)
        for f in range(F):
            out = rt.DeviceBuffer(width_bytes=W * 3, height=H)
            src, rank_stride = tiling.batch_unstripe_args(g0, f, F, max_rows, pitch)
            rt.check(h.rt_unstripe(src, pitch, rank_stride, out.ptr, out.pitch, W, H, stripe, world, None))
            rt.check(h.rt_device_synchronize())
            assert np.array_equal(out.to_host().reshape(H, W, 3), full[f]), "world %d frame %d" % (world, f)
            out.free()
        # all F frames in one rt_unstripe_batch call, as bench.py does
        outs = rt.DeviceBuffer(nbytes=F * H * pitch)
        src, rank_stride = tiling.batch_unstripe_args(g0, 0, F, max_rows, pitch)
        rt.check(h.rt_unstripe_batch(src, pitch, rank_stride, max_rows * pitch, outs.ptr, pitch, H * pitch, F, W, H, stripe, world, None))
        rt.check(h.rt_device_synchronize())
        got = outs.to_host().reshape(F, H, W, 3)
        for f in range(F):
            assert np.array_equal(got[f], full[f]), "batched unstripe world %d frame %d" % (world, f)
        outs.free()
        gathered.free()


def test_rotating_stripe_owner(rt, scenes, blob5k):
    """rt_render_stripes_batch_rotating / rt_unstripe_batch_rotating: frame i of a launch is rendered as owner
    (rank + first_frame + i) % world, so every rank renders every owner's share in turn.  Every virtual rank renders its F buffers
    into its slice of the gathered buffer; the un-stripe pass -- whole group, and a sub-range of it as a rank of the rotating-root
    exchange sees it (first_frame > 0) -- must give the frames a single launch renders; the host mirror (tiling.unstripe_host)
    agrees; the rows every rank renders over a group of `world` frames are the same number."""
    import importlib
    tiling = importlib.import_module("cuda-raytracing_amd.tiling")
    h = rt.libs()[0]
    sp = sd.blob_scene(scenes, blob5k).build_product(rt)
    sp.upload_to_device()
    rng = np.random.default_rng(77)
    shapes = [(210, 133, 16, 8, 11), (200, 120, 16, 3, 7), (128, 77, 8, 4, 8), (96, 50, 16, 5, 3)]
    for _ in range(12):                                         # and a dozen drawn ones: odd sizes, stripes of 1 .. 33 rows, up to 9 ranks
        shapes.append((int(rng.integers(17, 230)), int(rng.integers(9, 170)), int(rng.choice([1, 2, 3, 4, 7, 8, 16, 33])),
                       int(rng.integers(2, 10)), int(rng.integers(3, 13))))
    for (W, H, stripe, world, F) in shapes:
        cam = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF)
        poses = [(0.03 * i, -1.6 - 0.1 * i, 0.2, 0.0, 0.01 * i, 0.0) for i in range(F)]
        full = []
        for ps in poses:
            cam.set_pose(ps)
            full.append(rt.render(sp, cam))
        pitch = W * 3
        max_rows = max(tiling.stripe_rows(H, stripe, r, world) for r in range(world))
        gathered = rt.DeviceBuffer(nbytes=world * F * max_rows * pitch)
        g0 = gathered.ptr.value
        rt.check(h.rt_memcpy_h2d(gathered.ptr, np.full(world * F * max_rows * pitch, 9, np.uint8).ctypes.data, world * F * max_rows * pitch, None))
        for r in range(world):
            # two launches per rank (frames [0, cut) and [cut, F)): the second starts at first_frame = cut
            cut = F // 2
            base = g0 + r * F * max_rows * pitch
            ptrs = tiling.batch_local_ptrs(base, F, max_rows, pitch)
            cam.render_scene_stripes_batch(sp, poses[:cut], ptrs[:cut], pitch, stripe, r, world, synchronize=True, rotate_first=0)
            cam.render_scene_stripes_batch(sp, poses[cut:], ptrs[cut:], pitch, stripe, r, world, synchronize=True, rotate_first=cut)
        outs = rt.DeviceBuffer(nbytes=F * H * pitch)
        rt.check(h.rt_unstripe_batch_rotating(g0, pitch, F * max_rows * pitch, max_rows * pitch, outs.ptr, pitch, H * pitch, F, W, H, stripe, world, 0, None))
        rt.check(h.rt_device_synchronize())
        got = outs.to_host().reshape(F, H, W, 3)
        for f in range(F):
            assert np.array_equal(got[f], full[f]), "rotating owner: world %d frame %d" % (world, f)
        # frames [2, F) only, as the rank that assembles them would un-stripe them
        rt.check(h.rt_memcpy_h2d(outs.ptr, np.zeros(F * H * pitch, np.uint8).ctypes.data, F * H * pitch, None))
        rt.check(h.rt_unstripe_batch_rotating(g0 + 2 * max_rows * pitch, pitch, F * max_rows * pitch, max_rows * pitch, outs.ptr, pitch, H * pitch,
                                              F - 2, W, H, stripe, world, 2, None))
        rt.check(h.rt_device_synchronize())
        got = outs.to_host().reshape(F, H, W, 3)
        for f in range(2, F):
            assert np.array_equal(got[f - 2], full[f]), "rotating owner, sub-range: world %d frame %d" % (world, f)
        # host mirror
        host = gathered.to_host().reshape(world, F, max_rows, pitch)
        for f in range(F):
            assert np.array_equal(tiling.unstripe_host(host[:, f], H, stripe, world, frame_index=f).reshape(H, W, 3), full[f])
        # equal shares: over `world` consecutive frames every rank renders H rows in all
        for r in range(world):
            assert sum(tiling.stripe_rows(H, stripe, tiling.owner_of(r, f, world), world) for f in range(world)) == H
        # without rotation the same buffers would not un-stripe to the frames (the test can tell the two apart)
        rt.check(h.rt_unstripe_batch(g0, pitch, F * max_rows * pitch, max_rows * pitch, outs.ptr, pitch, H * pitch, F, W, H, stripe, world, None))
        rt.check(h.rt_device_synchronize())
        if (W, H, stripe, world, F) in shapes[:4]:
            assert not np.array_equal(outs.to_host().reshape(F, H, W, 3)[1], full[1])
        outs.free()
        gathered.free()


def test_c4_atrium_quarter_res(rt, orc, scenes, atrium):
    """configs[3] scene (deep BVH, leaves up to 96 triangles, camera inside) at 960x540: all planes vs the oracle."""
    W, H = 960, 540
    _compare(rt, orc, sd.atrium_scene(scenes, atrium), W, H, scenes.scaled_K(W), scenes.D_REF, scenes.C4["cam_pose"], threads=16)


def test_c4_atrium_4k(rt, orc, scenes, atrium):
    """configs[3] at its full 3840x2160 (1 primary ray per pixel -- the reference has no spp loop): RGB and hit ids
    against the oracle, plus the size-independent property that a centred 960x540 crop of the K-shifted camera
    reproduces the same pixels (ray generation depends on (x, y) only through K_inv * (x, y, 1))."""
    W, H = 3840, 2160
    c = scenes.C4
    desc = sd.atrium_scene(scenes, atrium)
    so = desc.build_oracle(orc)
    sp = desc.build_product(rt)
    sp.upload_to_device()
    K = scenes.scaled_K(W)
    cam = rt.Camera(W, H, K, scenes.D_REF)
    cam.set_pose(c["cam_pose"])
    dbg = rt.render_debug(sp, cam)
    ref = so.render(W, H, K, scenes.D_REF, c["cam_pose"], threads=32)
    assert np.array_equal(dbg["img"], ref["img"])
    assert np.array_equal(dbg["hit_tri"], ref["hit_tri"]) and np.array_equal(dbg["pops"], ref["pops"])
    assert np.array_equal(rt.render(sp, cam), ref["img"])
    ids = rt.render_ids(sp, cam)                               # production kernel
    assert np.array_equal(ids["hit_tri"], ref["hit_tri"]) and np.array_equal(ids["img"], ref["img"])
    # crop property: shifting the principal point by (-x0, -y0) renders the window [x0, x0+w) x [y0, y0+h)
    x0, y0, w, h = 1440, 810, 960, 540
    Kc = list(K)
    Kc[2] -= x0
    Kc[5] -= y0
    camc = rt.Camera(w, h, Kc, scenes.D_REF)
    camc.set_pose(c["cam_pose"])
    crop = rt.render(sp, camc)
    same = (crop == ref["img"][y0:y0 + h, x0:x0 + w]).all(axis=2).mean()
    assert same > 0.999, same                      # K_inv*(x,y,1) rounds differently after the shift: a handful of edge pixels may move
    so.close()


def test_heavy_first_order_with_alternating_frame_sizes_and_many_streams(rt, scenes, blob70k):
    """Single-frame launches of two sizes alternate on one scene (two cameras), each issued while a long batch is still
    running on another stream: every size has its own order state, so no call tears anything down or synchronises the device
    -- the calls return while the long batch is still in flight -- and all frames equal the natural-order kernel's.  Then more
    than four streams take turns on one size (a stream slot whose launch has finished is recycled), then a fifth and sixth
    size arrive (an idle state is evicted, or the launch falls back to natural order): same frames throughout."""
    import time
    import torch
    sp = sd.blob_scene(scenes, blob70k).build_product(rt)
    sp.upload_to_device()
    h = rt.libs()[0]
    sizes = [(1920, 1080), (1920, 544)]
    pose = scenes.C2_CAMERAS["mid"]
    cams, imgs, want = [], [], []
    for W, H in sizes:
        K = scenes.scaled_K(W)
        c = rt.Camera(W, H, K, scenes.D_REF)
        c.set_pose(pose)
        cams.append(c)
        imgs.append(rt.DeviceBuffer(width_bytes=W * 3, height=H))
        want.append(rt.render_ids(sp, c)["img"])
    # warm both states up (first use allocates)
    for c, im in zip(cams, imgs):
        c.render_scene(sp, im.ptr, im.pitch, synchronize=True)
    long_stream, side = torch.cuda.Stream(), torch.cuda.Stream()
    big = rt.Camera(1920, 1080, scenes.scaled_K(1920), scenes.D_REF)
    big.set_stream(long_stream.cuda_stream)
    F = 32
    bufs = rt.DeviceBuffer(nbytes=F * 1080 * 1920 * 3)
    ptrs = [bufs.ptr.value + k * 1080 * 1920 * 3 for k in range(F)]
    done = torch.cuda.Event()
    for _ in range(12):                                              # ~12 x 4.2 ms of work queued on long_stream
        big.render_scene_batch(sp, [pose] * F, ptrs, 1920 * 3)
    done.record(long_stream)
    for c in cams:
        c.set_stream(side.cuda_stream)
    t0 = time.perf_counter()
    for k in range(10):
        cams[k & 1].render_scene(sp, imgs[k & 1].ptr, imgs[k & 1].pitch)
    host_ms = (time.perf_counter() - t0) * 1e3
    still_running = not done.query()
    torch.cuda.synchronize()
    assert still_running and host_ms < 25.0, (still_running, host_ms)        # a device-wide synchronise would have waited ~50 ms for the batch
    for k in range(2):
        assert np.array_equal(imgs[k].to_host().reshape(sizes[k][1], sizes[k][0], 3), want[k])
    # six streams on one size
    streams = [torch.cuda.Stream() for _ in range(6)]
    outs = [rt.DeviceBuffer(width_bytes=1920 * 3, height=1080) for _ in streams]
    for rep in range(3):
        for st, o in zip(streams, outs):
            cams[0].set_stream(st.cuda_stream)
            cams[0].render_scene(sp, o.ptr, o.pitch)
        torch.cuda.synchronize()
        for o in outs:
            assert np.array_equal(o.to_host().reshape(1080, 1920, 3), want[0])
    # six sizes in rotation: more than the four cached states
    more = [(1920, 1080), (1920, 544), (1600, 912), (1280, 1024), (2048, 640), (1760, 800)]
    for rep in range(2):
        for W, H in more:
            c = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF)
            c.set_pose(pose)
            ref = rt.render_ids(sp, c)["img"]
            im = rt.DeviceBuffer(width_bytes=W * 3, height=H)
            c.render_scene(sp, im.ptr, im.pitch)
            c.render_scene(sp, im.ptr, im.pitch, synchronize=True)
            assert np.array_equal(im.to_host().reshape(H, W, 3), ref), (W, H)
    rt.check(h.rt_device_synchronize())


def test_deep_traversal_stack_spills(rt, orc, scenes):
    """Stack entries beyond the 16 kept in LDS go to the private spill array; counts and hits must not change."""
    desc = sd.deep_stack_scene(28)
    W, H = 96, 64
    img, ref = _compare(rt, orc, desc, W, H, scenes.scaled_K(W), scenes.D_REF, (0.0, -1.0, 0.0, 0, 0, 0))
    assert ref["stats"]["max_stack"] >= 24, ref["stats"]           # the scene really is deep (the reference stack holds 32)
    assert ref["stats"]["hits"] > 0


def test_outgrown_stack_with_several_instances(rt, orc, scenes, blob5k):
    """The timed kernels run the LDS-only stack and trace a ray whose stack outgrows it AGAIN on the general stack, from the first
    instance on (round 5).  A deep chain between two ordinary meshes, in all three orders, one of them scaled and rotated (its loop is
    the C++ one, the others' the hand-written one): the hit of an outgrown ray may come from an instance traced before or after the
    chain, and every plane must equal the oracle's -- the production kernel's image and hit ids included."""
    W, H = 96, 64
    K, pose = scenes.scaled_K(W), (0.0, -1.0, 0.0, 0, 0, 0)
    chain = sd.deep_stack_scene(26)
    chain_tris = chain.meshes[0]
    for order in ((0, 1, 2), (1, 0, 2), (1, 2, 0)):
        meshes = [chain_tris, ("obj", blob5k), ("obj", blob5k)]
        inst = [(0, 0, (0,) * 6, (1, 1, 1)), (1, 0, (0.4, 30.0, 0.2, 0, 0, 0), (1, 1, 1)), (2, 0, (-0.5, 3.0, -0.3, 0.3, 0.1, 0.2), (1.3, 0.8, 1.1))]
        desc = sd.SceneDesc([((0.2, 0.9, 0.4), None)], meshes, [inst[k] for k in order])
        img, ref = _compare(rt, orc, desc, W, H, K, scenes.D_REF, pose)
        assert ref["stats"]["max_stack"] >= 20 and ref["stats"]["hits"] > 0, ref["stats"]


def test_stack_depth_at_the_lds_boundary(rt, orc, scenes):
    """the kernels without the private overflow of the traversal stack (render_kernel<.., SPILL = false>) are chosen
    for trees of at most 17 levels -- 16 postponed nodes plus the sentinel fill exactly the 17 LDS rows.  Chains of 15 .. 20
    levels, whose central rays hold a postponed node on every level, cross that boundary: 16, 17 (the last LDS-only depth) and 18
    (the first with the overflow) must all be among them, and every plane must equal the oracle's."""
    W, H = 96, 64
    K, pose = scenes.scaled_K(W), (0.0, -1.0, 0.0, 0, 0, 0)
    levels, deepest = set(), {}
    for n in range(14, 22):
        desc = sd.deep_stack_scene(n)
        so = desc.build_oracle(orc)
        ref = so.render(W, H, K, scenes.D_REF, pose, threads=4)
        sp = desc.build_product(rt)
        sp.upload_to_device()
        cam = rt.Camera(W, H, K, scenes.D_REF)
        cam.set_pose(pose)
        depth = sp.info()["max_stack"]
        levels.add(depth)
        deepest[depth] = int(ref["stats"]["max_stack"])
        dbg, ids, img = rt.render_debug(sp, cam), rt.render_ids(sp, cam), rt.render(sp, cam)
        assert np.array_equal(img, ref["img"]) and np.array_equal(ids["img"], ref["img"]), n
        for name in PLANES:
            assert np.array_equal(dbg[name], ref[name]), (n, name)
        for name in ("hit_inst", "hit_tri"):
            assert np.array_equal(ids[name], ref[name]), (n, name)
        # the extension kernels share the stack
        cam.set_options(4, 1, 1)
        exr = so.render_ex(W, H, K, scenes.D_REF, pose, 4, 1, 1, threads=4)
        assert np.array_equal(rt.render_ex(sp, cam)["img"], exr["img"]), n
        so.close()
    assert {16, 17, 18} <= levels, levels
    assert deepest[17] >= 15 and deepest[18] >= 16, deepest      # the rays really fill the stack to (about) the tree's depth


def test_forced_overflow_stack_kernels_in_a_child_process():
    """RT_STACK_SPILL=1 selects the kernels WITH the private overflow for every scene (it is read once per process): the
    smoke scenes -- shallow trees that otherwise run the LDS-only kernels -- against the oracle in a child process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=root, env=dict(os.environ, RT_STACK_SPILL="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stderr[-2000:]


# ---- extension: spp / bounces / sun+shadow; semantics defined by the oracle (parity unpinned) ------

def _compare_ex(rt, orc, desc, W, H, K, pose, spp, bounces, lighting, threads=16):
    so = desc.build_oracle(orc)
    ref = so.render_ex(W, H, K, sd_D, pose, spp, bounces, lighting, threads=threads)
    sp = desc.build_product(rt)
    sp.upload_to_device()
    cam = rt.Camera(W, H, K, sd_D)
    cam.set_pose(pose)
    cam.set_options(spp, bounces, lighting)
    got = rt.render_ex(sp, cam)
    nbad = int((got["img"] != ref["img"]).any(axis=2).sum())
    assert nbad == 0, "%d pixels differ (spp %d bounces %d lighting %d)" % (nbad, spp, bounces, lighting)
    assert np.array_equal(got["total_pops"], ref["total_pops"])
    # Camera::render_scene with non-default options takes the same path
    assert np.array_equal(rt.render(sp, cam), ref["img"])
    so.close()
    return got, ref


sd_D = None


@pytest.fixture(autouse=True)
def _set_d(scenes):
    global sd_D
    sd_D = scenes.D_REF


def test_ex_degenerates_to_reference_frame(rt, orc, scenes, blob5k):
    """spp = 1, bounces = 0, lighting = 0 through rt_render_ex == rt_render == oracle reference frame."""
    m = sd.SHINY_CAMERA
    desc = sd.shiny_scene(scenes, blob5k)
    K = scenes.scaled_K(m["width"])
    got, _ = _compare_ex(rt, orc, desc, m["width"], m["height"], K, m["pose"], 1, 0, 0)
    so = desc.build_oracle(orc)
    ref = so.render(m["width"], m["height"], K, scenes.D_REF, m["pose"], threads=8)
    assert np.array_equal(got["img"], ref["img"]) and np.array_equal(got["total_pops"], ref["pops"])


@pytest.mark.parametrize("spp,bounces,lighting", [(1, 0, 1), (1, 3, 0), (4, 2, 1), (16, 8, 1)])
def test_ex_modes_match_oracle(rt, orc, scenes, blob5k, spp, bounces, lighting):
    m = sd.SHINY_CAMERA
    _compare_ex(rt, orc, sd.shiny_scene(scenes, blob5k), m["width"], m["height"], scenes.scaled_K(m["width"]), m["pose"],
                spp, bounces, lighting)


def test_ex_samples_in_chunks(rt, orc, scenes, blob5k, monkeypatch):
    """A frame whose sample data exceeds the scratch budget is rendered in chunks of sample indices (running sums kept
    between chunks): forced here with a budget of 3 samples for 10 spp -> chunks of 3, 3, 3, 1; same frame, same pops."""
    m = sd.SHINY_CAMERA
    monkeypatch.setenv("RT_EX_SCRATCH_BYTES", str(3 * m["width"] * m["height"] * 16))
    _compare_ex(rt, orc, sd.shiny_scene(scenes, blob5k), m["width"], m["height"], scenes.scaled_K(m["width"]), m["pose"], 10, 3, 1)
    monkeypatch.setenv("RT_EX_SCRATCH_BYTES", "1")                      # degenerate budget: one sample per chunk
    _compare_ex(rt, orc, sd.shiny_scene(scenes, blob5k), m["width"], m["height"], scenes.scaled_K(m["width"]), m["pose"], 4, 2, 1)


@pytest.mark.parametrize("spp,bounces,lighting,size", [(3, 2, 1, (200, 120)), (2, 5, 0, (203, 117)), (5, 0, 1, (64, 48)), (6, 4, 1, (333, 190)),
                                                       (16, 0, 0, (131, 77)), (33, 1, 1, (97, 61)), (70, 2, 1, (64, 48)), (130, 0, 0, (45, 31))])
def test_ex_forms_are_bit_identical(rt, orc, scenes, blob5k, monkeypatch, spp, bounces, lighting, size):
    """The three ways the extension renderer maps (pixel, sample) pairs to lanes -- a pixel's samples in one wave with the sum
    taken there (the default from 4 samples on; more than 64 samples take several launches), one sample index per launch row with
    sample planes and a resolve pass (RT_EX_PIXEL_WAVES=0), and the wavefront form (RT_EX_WAVEFRONT=1: one cast per launch, live
    paths compacted between launches) -- give the same frame and the same
    node-pop totals, and both equal the oracle -- also with small queue groups (4 segments), with the samples in chunks
    of 2 and of 1 (the primary launch's workgroup then covers 2 or 4 pixel quads), and for a rank's stripes."""
    W, H = size
    desc = sd.shiny_scene(scenes, blob5k)
    K, pose = scenes.scaled_K(W), sd.SHINY_CAMERA["pose"]
    got, ref = _compare_ex(rt, orc, desc, W, H, K, pose, spp, bounces, lighting)
    sp = desc.build_product(rt)
    sp.upload_to_device()
    cam = rt.Camera(W, H, K, sd_D)
    cam.set_pose(pose)
    cam.set_options(spp, bounces, lighting)
    # (the default from 4 samples on: a pixel's samples share a wave and are summed there; RT_EX_PIXEL_WAVES=0: one sample index per
    # launch row, sample planes and a resolve pass)
    # (round 6: RT_EX_SPLIT=1 renders the camera ray of a path with bounces or lighting in a launch of its own, render_ex_kernel<..,
    # PHASE 1 / 2>; RT_EX_SPLIT_BYTES forces that form into chunks of 4 and of 12 workgroups)
    variants = [{"RT_EX_SPLIT": "1"}, {"RT_EX_SPLIT": "1", "RT_EX_SPLIT_BYTES": "8192"}, {"RT_EX_SPLIT": "1", "RT_EX_SPLIT_BYTES": str(12 * 2048)},
                {"RT_EX_PIXEL_WAVES": "0"}, {"RT_EX_PIXEL_WAVES": "0", "RT_EX_SCRATCH_BYTES": str(3 * W * H * 16)}, {"RT_EX_WAVEFRONT": "1"}, {"RT_EX_WAVEFRONT": "1", "RT_EX_GROUP": "4"},
                {"RT_EX_WAVEFRONT": "1", "RT_EX_SCRATCH_BYTES": str(2 * (W + 16) * (H + 16) * 380)},
                {"RT_EX_WAVEFRONT": "1", "RT_EX_SCRATCH_BYTES": "1"}]
    for env in variants:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        other = rt.render_ex(sp, cam)
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(other["img"], got["img"]), env
        assert np.array_equal(other["total_pops"], got["total_pops"]), env
    # a rank's stripes through the wavefront form
    monkeypatch.setenv("RT_EX_WAVEFRONT", "1")
    import importlib
    tiling = importlib.import_module("cuda-raytracing_amd.tiling")
    h = rt.libs()[0]
    world, stripe = 3, 16
    pitch = W * 3
    max_rows = max(tiling.stripe_rows(H, stripe, r, world) for r in range(world))
    gathered = rt.DeviceBuffer(nbytes=world * max_rows * pitch)
    for r in range(world):
        cam.render_scene_stripes(sp, gathered.ptr.value + r * max_rows * pitch, pitch, stripe, r, world, synchronize=True)
    out = rt.DeviceBuffer(width_bytes=W * 3, height=H)
    rt.check(h.rt_unstripe(gathered.ptr, pitch, max_rows * pitch, out.ptr, out.pitch, W, H, stripe, world, None))
    rt.check(h.rt_device_synchronize())
    assert np.array_equal(out.to_host().reshape(H, W, 3), ref["img"])


def test_c3_bunny_64spp_8bounces(rt, orc, scenes, blob70k):
    """BASELINE.json configs[2] shape: the 70k blob, 64 spp, 8 bounces, sun + shadow -- at 480x270 so that the oracle
    finishes in seconds (the semantics are the extension's, parity unpinned)."""
    W, H = 480, 270
    desc = sd.SceneDesc([((0.9, 0.5, 0.2), None, dict(roughness=0.05, metallic=0.4))], [("obj", blob70k)], [(0, 0, (0,) * 6, (1, 1, 1))])
    got, ref = _compare_ex(rt, orc, desc, W, H, scenes.scaled_K(W), scenes.C2_CAMERAS["mid"], 64, 8, 1, threads=32)
    assert ref["stats"]["rays"] > 64 * W * H


def test_c5_shape_tiled_64spp(rt, orc, scenes, blob5k):
    """BASELINE.json configs[4] shape (64 spp, frame tiled over 8 ranks, gathered and un-striped) at 320x180: the
    stitched extension frame equals the single-device extension frame and the oracle."""
    import ctypes as C
    import importlib
    tiling = importlib.import_module("cuda-raytracing_amd.tiling")
    h = rt.libs()[0]
    W, H, world, stripe = 320, 180, 8, 16
    desc = sd.SceneDesc([((0.9, 0.5, 0.2), None, dict(roughness=0.1, metallic=0.5))], [("obj", blob5k)], [(0, 0, (0,) * 6, (1, 1, 1))])
    K, pose = scenes.scaled_K(W), scenes.C2_CAMERAS["mid"]
    got, ref = _compare_ex(rt, orc, desc, W, H, K, pose, 64, 2, 1)
    sp = desc.build_product(rt)
    sp.upload_to_device()
    cam = rt.Camera(W, H, K, scenes.D_REF)
    cam.set_pose(pose)
    cam.set_options(64, 2, 1)
    pitch = W * 3
    max_rows = max(tiling.stripe_rows(H, stripe, r, world) for r in range(world))
    gathered = rt.DeviceBuffer(nbytes=world * max_rows * pitch)
    for r in range(world):
        cam.render_scene_stripes(sp, gathered.ptr.value + r * max_rows * pitch, pitch, stripe, r, world, synchronize=True)
    out = rt.DeviceBuffer(width_bytes=W * 3, height=H)
    rt.check(h.rt_unstripe(gathered.ptr, pitch, max_rows * pitch, out.ptr, out.pitch, W, H, stripe, world, None))
    rt.check(h.rt_device_synchronize())
    assert np.array_equal(out.to_host().reshape(H, W, 3), ref["img"])


def test_cpp_demo_application(rt, orc, scenes, blob5k, tmp_path):
    """examples/demo_main.cpp (the reference's main() flow written against the host C++ API) builds with plain g++,
    runs, and its out.png equals the same scene rendered through the oracle -- plus the FPS overlay of display_image."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "demo")
    pkg = os.path.join(root, "cuda-raytracing_amd")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    "-I" + os.path.join(root, "include"), "-I" + os.path.join(pkg, "csrc", "host"),
                    os.path.join(root, "examples", "demo_main.cpp"), "-L" + pkg, "-lrt_host", "-lrt_hip", "-Wl,-rpath," + pkg,
                    "-o", exe], check=True)
    tex = sd.checker_texture(48, 40, seed=4)
    ppm = str(tmp_path / "tex.ppm")
    with open(ppm, "wb") as f:
        f.write(b"P6\n48 40\n255\n" + tex[:, :, ::-1].tobytes())          # PPM is R,G,B; the material stores B,G,R
    png = str(tmp_path / "out.png")
    r = subprocess.run([exe, blob5k, png, "3", ppm], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Number of nodes" in r.stdout and "FPS" in r.stdout
    got = rt.read_image(png)
    desc = sd.SceneDesc([((0.9, 0.5, 0.2), None), ((1.0, 1.0, 1.0), tex)], [("obj", blob5k)],
                        [(0, 1, (0,) * 6, (1, 1, 1)), (0, 0, (-0.6, 1.48, 0.73, 0, 0, 0), (0.4, 0.4, 0.4))])
    so = desc.build_oracle(orc)
    ref = so.render(1920, 1080, scenes.K_1080, scenes.D_REF, (0.0, -1.6, 0.2, 0, 0, 0), planes=False, threads=16)["img"]
    # display_image's "FPS: ..." overlay sits in rows 9..29 from column 10 on; everything else is the rendered frame
    box = np.zeros(got.shape[:2], bool)
    box[9:30, 10:10 + 18 * 24] = True
    assert np.array_equal(got[~box], ref[~box])
    changed = (got != ref).any(-1)
    assert changed.any() and (got[changed] == (0, 255, 0)).all() and changed[9:30, 10:25].any()     # the green 'F'
    so.close()


def _same_tree(a, b):
    assert np.array_equal(a["child"], b["child"]) and np.array_equal(a["leaf_count"], b["leaf_count"])
    assert np.array_equal(a["leaf_idx"], b["leaf_idx"])
    assert np.array_equal(a["boxes"], b["boxes"])          # by value: atomics may pick -0.0 where the host fold picks +0.0
    assert np.array_equal(a["tris"].view(np.uint32), b["tris"].view(np.uint32))


@pytest.fixture(params=[None, "0", "7", "33"], ids=["small64", "levels_only", "small7", "small33"])
def bvh_small_limit(request):
    """RT_BVH_SMALL: subtrees of at most that many triangles are finished by one wave each (default 64); 0 sends every
    node through the level loop.  Both paths, and mixtures in between, must give the same tree."""
    if request.param is None:
        os.environ.pop("RT_BVH_SMALL", None)
    else:
        os.environ["RT_BVH_SMALL"] = request.param
    yield request.param
    os.environ.pop("RT_BVH_SMALL", None)


def test_gpu_bvh_build_matches_host_builder(rt, scenes, blob5k, blob70k, atrium, bvh_small_limit):
    """rt_bvh_build: the GPU build (level loop + one wave per small subtree) gives the host builder's
    (= the reference's) tree node for node -- children, leaf lists, bounds, pre-order numbering -- on OBJ meshes, soups,
    degenerate and deep inputs."""
    for path in (blob5k, blob70k, atrium):
        host = rt.Mesh.load_obj(path)
        dev = rt.Mesh.load_obj(path, gpu_build=True)
        assert dev.num_nodes == host.num_nodes and dev.max_level == host.max_level
        _same_tree(dev.dump(), host.dump())
    for seed, n in [(1, 0), (2, 1), (3, 2), (4, 3), (5, 64), (7, 63), (8, 65), (9, 128), (6, 1500)]:
        tris = sd.random_triangles(n, seed=seed) if n else np.zeros((0, 18), np.float32)
        _same_tree(rt.Mesh.from_triangles(tris, gpu_build=True).dump(), rt.Mesh.from_triangles(tris).dump())
    base = sd.random_triangles(6, seed=3, spread=0.5, size=0.6)
    dup = np.concatenate([np.repeat(base[:1], 40, axis=0), np.repeat(base[1:2], 33, axis=0), base[2:]])
    _same_tree(rt.Mesh.from_triangles(dup, gpu_build=True).dump(), rt.Mesh.from_triangles(dup).dump())
    # the partition path of meshes above 1 M triangles (device-wide library scan instead of the fused block scan)
    os.environ["RT_BVH_LIBRARY_SCAN"] = "1"
    try:
        _same_tree(rt.Mesh.load_obj(blob70k, gpu_build=True).dump(), rt.Mesh.load_obj(blob70k).dump())
    finally:
        del os.environ["RT_BVH_LIBRARY_SCAN"]
    # NaN and infinite coordinates: fminf / fmaxf skip NaN in the host builder's folds, the float atomics must too
    weird = sd.random_triangles(700, seed=12, spread=1.0, size=0.3)
    weird[5, 0] = np.nan; weird[17, 4] = np.inf; weird[40, 8] = -np.inf; weird[41, :9] = np.nan; weird[100, 2] = np.nan
    weird[300:310, 1] = np.inf; weird[400, 3:6] = -np.inf
    nans = sd.random_triangles(900, seed=13, spread=1.0, size=0.3)
    nans[::37, 0] = np.nan; nans[5::53, 4] = np.nan; nans[11::71, 6:9] = np.nan; nans[200, :9] = np.nan
    with np.errstate(all="ignore"):
        _same_tree(rt.Mesh.from_triangles(weird, gpu_build=True).dump(), rt.Mesh.from_triangles(weird).dump())
        host_nan = rt.Mesh.from_triangles(nans)
        assert host_nan.num_nodes > 100                        # NaN vertices alone do not stop the splitting
        a, b = rt.Mesh.from_triangles(nans, gpu_build=True).dump(), host_nan.dump()
        for k in ("child", "leaf_count", "leaf_idx"):
            assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(a["boxes"], b["boxes"], equal_nan=True)
    chain = sd.deep_stack_scene(28).meshes[0][1]
    d = rt.Mesh.from_triangles(chain, gpu_build=True)
    assert d.max_level == 28
    _same_tree(d.dump(), rt.Mesh.from_triangles(chain).dump())


def test_gpu_built_mesh_renders_identically(rt, orc, scenes, blob5k):
    """A scene whose mesh BVH was built on the GPU renders the oracle's frame and visit counts."""
    desc = sd.blob_scene(scenes, blob5k)
    so = desc.build_oracle(orc)
    W, H = 320, 180
    K, pose = scenes.scaled_K(W), scenes.C2_CAMERAS["mid"]
    ref = so.render(W, H, K, scenes.D_REF, pose, threads=8)
    sp = rt.Scene()
    sp.add_material(scenes.C2["albedo"])
    sp.add_mesh(rt.Mesh.load_obj(blob5k, gpu_build=True))
    sp.add_mesh_instance(0, 0)
    sp.upload_to_device()
    cam = rt.Camera(W, H, K, scenes.D_REF)
    cam.set_pose(pose)
    dbg = rt.render_debug(sp, cam)
    assert np.array_equal(dbg["img"], ref["img"]) and np.array_equal(dbg["pops"], ref["pops"]) and np.array_equal(dbg["hit_tri"], ref["hit_tri"])


def test_8k_frame_bands(rt, orc, scenes, blob70k):
    """BASELINE.json configs[4] resolution (7680x4320, 1 primary ray/pixel): three 24-row bands of the GPU frame against
    the oracle (RGB + hit ids + pops), and the striped 8-rank rendering of the full frame against the single launch."""
    import importlib
    tiling = importlib.import_module("cuda-raytracing_amd.tiling")
    h = rt.libs()[0]
    W, H = 7680, 4320
    desc = sd.blob_scene(scenes, blob70k)
    sp = desc.build_product(rt)
    sp.upload_to_device()
    K, pose = scenes.scaled_K(W), scenes.C2_CAMERAS["mid"]
    cam = rt.Camera(W, H, K, scenes.D_REF)
    cam.set_pose(pose)
    dbg = rt.render_debug(sp, cam)
    ids = rt.render_ids(sp, cam)                               # production kernel
    so = desc.build_oracle(orc)
    for y0 in (0, 2148, H - 24):
        ref = so.render(W, H, K, scenes.D_REF, pose, y0=y0, y1=y0 + 24, threads=24)
        for k in ("img", "hit_tri", "pops"):
            assert np.array_equal(dbg[k][y0:y0 + 24], ref[k][y0:y0 + 24]), (k, y0)
        for k in ("img", "hit_tri"):
            assert np.array_equal(ids[k][y0:y0 + 24], ref[k][y0:y0 + 24]), ("production", k, y0)
    so.close()
    world, stripe, pitch = 8, 16, W * 3
    max_rows = max(tiling.stripe_rows(H, stripe, r, world) for r in range(world))
    gathered = rt.DeviceBuffer(nbytes=world * max_rows * pitch)
    for r in range(world):
        cam.render_scene_stripes(sp, gathered.ptr.value + r * max_rows * pitch, pitch, stripe, r, world)
    out = rt.DeviceBuffer(width_bytes=pitch, height=H)
    rt.check(h.rt_unstripe(gathered.ptr, pitch, max_rows * pitch, out.ptr, out.pitch, W, H, stripe, world, None))
    rt.check(h.rt_device_synchronize())
    assert np.array_equal(out.to_host().reshape(H, W, 3), dbg["img"])


# RT_FUZZ_SEEDS=n widens the two differential fuzz tests below (a one-off campaign on the GPU box; the suite runs 24 and 4);
# RT_FUZZ_FIRST=k starts at seed k (a second campaign over seeds the first did not see)
_FUZZ_FIRST = int(os.environ.get("RT_FUZZ_FIRST", 0))
@pytest.mark.parametrize("seed", range(_FUZZ_FIRST, _FUZZ_FIRST + int(os.environ.get("RT_FUZZ_SEEDS", 24))))
def test_fuzz_random_scenes(rt, orc, scenes, blob5k, seed):
    """Differential fuzzing: random soups / blob instances with random poses, non-uniform scales, random materials
    (albedo or texture), random cameras and odd frame sizes; all six parity planes and RGB against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    n_mesh = int(rng.integers(1, 4))
    meshes, materials, instances = [], [], []
    for _ in range(n_mesh):
        if rng.random() < 0.35:
            meshes.append(("obj", blob5k))
        else:
            meshes.append(("tris", sd.random_triangles(int(rng.integers(1, 400)), seed=int(rng.integers(1 << 30)),
                                                       spread=float(rng.uniform(0.3, 1.5)), size=float(rng.uniform(0.05, 0.6)))))
    for _ in range(int(rng.integers(1, 4))):
        tex = sd.checker_texture(int(rng.integers(2, 70)), int(rng.integers(2, 50)), seed=int(rng.integers(1 << 30))) if rng.random() < 0.5 else None
        materials.append((tuple(rng.uniform(0, 1, 3)), tex))
    for _ in range(int(rng.integers(1, 6))):
        pose = tuple(np.concatenate([rng.uniform(-1.5, 1.5, 3), rng.uniform(-3.1, 3.1, 3)]))
        scale = tuple(rng.uniform(0.3, 1.8, 3)) if rng.random() < 0.7 else (1.0, 1.0, 1.0)
        instances.append((int(rng.integers(n_mesh)), int(rng.integers(len(materials))), pose, scale))
    # (from seed 34000 on: a fresh generator, so that the scenes of the earlier campaigns stay what they were.  A third of the scenes
    # get one instance whose transform is the identity -- the case the hand-written traversal loop takes)
    if seed >= 34000 or seed < 12:
        extra = np.random.default_rng(5000 + seed)
        if extra.random() < (0.34 if seed >= 34000 else 0.5):
            k = int(extra.integers(len(instances)))
            instances[k] = (instances[k][0], instances[k][1], (0.0,) * 6, (1.0, 1.0, 1.0))
    # (from seed 64000 on, round 6: 40 % of the scenes get one instance that is TRANSLATED only -- unit scale, no rotation: the shape of
    # the reference's own scene (kernel.cu:209-240) and the case the hand-written loop answers with `pt - inv_pose` in its candidate block)
    if seed >= 64000 or 12 <= seed < 24:
        extra = np.random.default_rng(9000 + seed)
        if extra.random() < 0.4:
            k = int(extra.integers(len(instances)))
            instances[k] = (instances[k][0], instances[k][1], tuple(instances[k][2][:3]) + (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    W, H = int(rng.integers(20, 200)), int(rng.integers(20, 140))
    cam_pose = tuple(np.concatenate([rng.uniform(-1, 1, 1), rng.uniform(-5, -2, 1), rng.uniform(-1, 1, 1), rng.uniform(-0.4, 0.4, 3)]))
    # (every third scene takes its trees from the GPU builder)
    _compare(rt, orc, sd.SceneDesc(materials, meshes, instances), W, H, scenes.scaled_K(W), scenes.D_REF, cam_pose, gpu_build=seed % 3 == 2)


@pytest.mark.parametrize("seed", range(_FUZZ_FIRST, _FUZZ_FIRST + max(4, int(os.environ.get("RT_FUZZ_SEEDS", 4)) // 3)))
def test_fuzz_extension_modes(rt, orc, scenes, blob5k, seed, monkeypatch):
    """Random spp / bounces / lighting, random metallic / roughness, rotated and scaled instances: extension kernel vs oracle,
    and its wavefront form (RT_EX_WAVEFRONT=1) vs both."""
    rng = np.random.default_rng(7000 + seed)
    mats = [(tuple(rng.uniform(0.1, 1, 3)), sd.checker_texture(16, 12, seed=seed) if rng.random() < 0.5 else None,
             dict(roughness=float(rng.choice([0.0, 0.05, 0.3])), metallic=float(rng.choice([0.0, 0.3, 0.8])))) for _ in range(3)]
    meshes = [("obj", blob5k), ("tris", sd.random_triangles(120, seed=40 + seed, spread=1.2, size=0.5))]
    inst = [(int(rng.integers(2)), int(rng.integers(3)), tuple(np.concatenate([rng.uniform(-1.2, 1.2, 3), rng.uniform(-1, 1, 3)])),
             tuple(rng.uniform(0.5, 1.4, 3))) for _ in range(3)]
    W, H = 96, 64
    opts = (int(rng.integers(1, 9)), int(rng.integers(0, 5)), int(rng.integers(0, 2)))
    desc = sd.SceneDesc(mats, meshes, inst)
    _compare_ex(rt, orc, desc, W, H, scenes.scaled_K(W), (0.2, -3.5, 0.5, 0.05, -0.1, 0.02), *opts)
    monkeypatch.setenv("RT_EX_SPLIT", "1")                                 # (the two-launch form of the bounce kernel)
    _compare_ex(rt, orc, desc, W, H, scenes.scaled_K(W), (0.2, -3.5, 0.5, 0.05, -0.1, 0.02), *opts)
    monkeypatch.delenv("RT_EX_SPLIT")
    monkeypatch.setenv("RT_EX_WAVEFRONT", "1")
    _compare_ex(rt, orc, desc, W, H, scenes.scaled_K(W), (0.2, -3.5, 0.5, 0.05, -0.1, 0.02), *opts)


def _adversarial_mesh(o, rng):
    """One mesh of a randomly chosen awkward kind, as [n,18] triangles (uv random).  Kinds: vertices on a coarse lattice (flat boxes,
    shared edges, coplanar and coincident triangles), zero-area triangles (NaN normals), piles of coincident triangles (leaves above
    30, equal-distance candidates), coordinates scaled to 1e6..1e18 or 1e-6..1e-20, needle-thin slivers (huge invDenom), a few
    non-finite vertices (a flagged mesh: generic loop only), or a plain soup."""
    kind = str(rng.choice(["lattice", "degenerate", "piles", "huge", "tiny", "sliver", "nonfinite", "soup", "soup"]))

    def tris_from(v):
        out = np.stack([o.tri_from_vertices(np.asarray(t, np.float32).ravel()) for t in v])
        out[:, 12:18] = rng.uniform(0, 1, (len(out), 6)).astype(np.float32)
        return out

    soup = sd.random_triangles(int(rng.integers(1, 60)), seed=int(rng.integers(1 << 30)), spread=float(rng.uniform(0.3, 1.2)), size=float(rng.uniform(0.1, 0.6)))
    if kind == "soup":
        return kind, soup
    if kind == "lattice":
        step = float(rng.choice([0.25, 0.5, 1.0]))
        n = int(rng.integers(4, 70))
        v = rng.integers(-3, 4, (n, 3, 3)).astype(np.float32) * np.float32(step)
        for i in range(n):
            if rng.random() < 0.6:                                       # the triangle lies in a plane x / y / z = const
                v[i, :, int(rng.integers(3))] = np.float32(rng.integers(-2, 3)) * np.float32(step)
        return kind, tris_from(v)
    if kind == "degenerate":
        n = int(rng.integers(6, 40))
        v = rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32)
        a, b = n // 3, 2 * n // 3
        v[:a, 1] = v[:a, 0]                                              # two equal vertices
        v[a:b, 2] = v[a:b, 0] + np.float32(rng.choice([2.0, 0.5, -1.0])) * (v[a:b, 1] - v[a:b, 0])   # collinear
        v[b:b + 2] = v[b:b + 2, :1]                                      # a point
        return kind, np.concatenate([tris_from(v), soup])
    if kind == "piles":
        base = sd.random_triangles(int(rng.integers(2, 6)), seed=int(rng.integers(1 << 30)), spread=0.5, size=0.6)
        reps = [int(rng.choice([1, 2, 5, 31, 34, 45])) for _ in base]
        return kind, np.concatenate([np.repeat(base[i:i + 1], r, axis=0) for i, r in enumerate(reps)])
    if kind in ("huge", "tiny"):
        e = float(rng.integers(6, 19)) if kind == "huge" else -float(rng.integers(6, 21))
        m = np.float32(10.0 ** e)
        v = (rng.uniform(-1, 1, (int(rng.integers(4, 30)), 3, 3)) * m).astype(np.float32)
        v[:, :, 1] += np.float32(2.0) * m                                # in front of a camera that looks along +y from near the origin
        return kind, (np.concatenate([tris_from(v), soup]) if rng.random() < 0.5 else tris_from(v))
    if kind == "sliver":
        n = int(rng.integers(4, 30))
        v = rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32)
        t = rng.uniform(0.1, 0.9, (n, 1)).astype(np.float32)
        v[:, 2] = v[:, 0] + t * (v[:, 1] - v[:, 0]) + rng.uniform(-1, 1, (n, 3)).astype(np.float32) * np.float32(10.0 ** -float(rng.integers(3, 8)))
        return kind, np.concatenate([tris_from(v), soup])
    bad = soup.copy()                                                    # nonfinite
    for i in rng.integers(0, len(bad), int(rng.integers(1, 4))):
        bad[i, int(rng.integers(9))] = np.float32(rng.choice([np.nan, np.inf, -np.inf]))
        bad[i, :12] = o.tri_from_vertices(bad[i, :9])[:12]
    return kind, bad


def _adversarial_scene(scenes, rng, shiny=False):
    """-> (SceneDesc, W, H, K, camera pose, description).  shiny: materials carry random roughness / metallic (extension modes)."""
    import orc as orc_mod
    o = orc_mod.oracle()
    kinds, meshes = [], []
    for _ in range(int(rng.integers(1, 4))):
        k, t = _adversarial_mesh(o, rng)
        kinds.append(k); meshes.append(("tris", t))
    materials = [(tuple(rng.uniform(0, 1, 3)), sd.checker_texture(int(rng.integers(2, 40)), int(rng.integers(2, 30)), seed=int(rng.integers(1 << 30))) if rng.random() < 0.5 else None)
                 for _ in range(int(rng.integers(1, 3)))]
    if shiny:
        materials = [m + (dict(roughness=float(rng.choice([0.0, 0.05, 0.3])), metallic=float(rng.choice([0.0, 0.3, 0.8]))),) for m in materials]
    half_pi = float(np.float32(np.pi / 2))
    instances = []
    for _ in range(int(rng.integers(1, 5))):
        form = str(rng.choice(["identity", "translated", "lattice_step", "signed_zero", "quarter_turn", "any"]))
        if form == "identity":
            pose = (0.0,) * 6
        elif form == "translated":
            pose = tuple(rng.uniform(-1.5, 1.5, 3)) + (0.0, 0.0, 0.0)
        elif form == "lattice_step":
            pose = tuple(float(v) * 0.25 for v in rng.integers(-6, 7, 3)) + (0.0, 0.0, 0.0)
        elif form == "signed_zero":
            pose = tuple(float(np.float32(-0.0)) if rng.random() < 0.5 else 0.0 for _ in range(6))
        elif form == "quarter_turn":
            pose = tuple(rng.uniform(-1, 1, 3)) + tuple(half_pi * float(v) for v in rng.integers(-2, 3, 3))
        else:
            pose = tuple(np.concatenate([rng.uniform(-1.5, 1.5, 3), rng.uniform(-3.1, 3.1, 3)]))
        sform = str(rng.choice(["unit", "unit", "pow2", "mirror", "large", "small", "any"]))
        scale = {"unit": (1.0, 1.0, 1.0),
                 "pow2": tuple(float(2.0 ** v) for v in rng.integers(-2, 3, 3)),
                 "mirror": tuple(float(v) for v in rng.choice([-1.0, 1.0, -0.5, 2.0], 3)),
                 "large": tuple(float(v) for v in rng.uniform(1e2, 1e4, 3)),
                 "small": tuple(float(v) for v in rng.uniform(1e-4, 1e-2, 3)),
                 "any": tuple(rng.uniform(0.3, 1.8, 3))}[sform]
        instances.append((int(rng.integers(len(meshes))), int(rng.integers(len(materials))), pose, scale))
    W, H = int(rng.integers(16, 97)), int(rng.integers(16, 65))
    K = scenes.scaled_K(W)
    view = str(rng.choice(["any", "axis", "on_lattice", "inside"]))
    if view == "any":
        cam_pose = tuple(np.concatenate([rng.uniform(-1, 1, 1), rng.uniform(-5, -2, 1), rng.uniform(-1, 1, 1), rng.uniform(-0.4, 0.4, 3)]))
    elif view == "inside":
        cam_pose = tuple(np.concatenate([rng.uniform(-0.5, 0.5, 3), rng.uniform(-3.1, 3.1, 3)]))
    else:
        # the principal point on a pixel centre and no rotation: the centre column / row of rays have direction components that are 0
        W, H = W & ~1, H & ~1
        K = (K[0], 0.0, W / 2.0, 0.0, K[4], H / 2.0, 0.0, 0.0, 1.0)
        cam_pose = ((0.0, -3.0, 0.0) if view == "axis" else tuple(float(v) * 0.25 for v in (rng.integers(-4, 5), rng.integers(-16, -7), rng.integers(-4, 5)))) + (0.0, 0.0, 0.0)
    info = "meshes %s instances %s view %s %s" % (kinds, [(i[0], i[2], i[3]) for i in instances], view, (W, H))
    return sd.SceneDesc(materials, meshes, instances), W, H, K, cam_pose, info


# RT_FUZZ_ADV_SEEDS=n / RT_FUZZ_ADV_FIRST=k: a campaign of the adversarial fuzz below (the suite runs 12 seeds)
_ADV_FIRST = int(os.environ.get("RT_FUZZ_ADV_FIRST", 0))
@pytest.mark.parametrize("seed", range(_ADV_FIRST, _ADV_FIRST + int(os.environ.get("RT_FUZZ_ADV_SEEDS", 12))))
def test_fuzz_adversarial_scenes(rt, orc, scenes, seed):
    """Differential fuzzing where the arithmetic is least comfortable (round 6, third session): test_degenerate_and_extreme_geometry
    and test_non_finite_vertices_... hold one fixed scene per kind, every one a single instance; here the kinds are drawn at random and
    COMBINED with the instance forms the hand-written loop distinguishes -- identity, translated only (also by lattice steps and signed
    zeros), quarter turns, any rotation; unit, power-of-two, mirrored (a negative component), very large and very small scales -- and
    with cameras whose central rays have direction components that are exactly zero (infinite inverses: the generic loop for those
    waves), that sit on the lattice the triangles use (rays along edges and in triangle planes), or inside the geometry.  Every plane
    and the RGB against the oracle, production and instrumented kernel, single frame and a batch of four through view records."""
    desc, W, H, K, cam_pose, info = _adversarial_scene(scenes, np.random.default_rng(31000 + seed))
    print("seed", seed, info)
    _compare(rt, orc, desc, W, H, K, scenes.D_REF, cam_pose, threads=4, gpu_build=seed % 3 == 2)


@pytest.mark.parametrize("seed", range(_ADV_FIRST, _ADV_FIRST + max(4, int(os.environ.get("RT_FUZZ_ADV_SEEDS", 12)) // 3)))
def test_fuzz_adversarial_extension_modes(rt, orc, scenes, seed, monkeypatch):
    """The same awkward scenes through the extension kernel (random spp / bounces / lighting, rough and metallic materials): secondary rays
    that start at hits on zero-area triangles (NaN normals), mirrored and extreme scales, origins on the lattice.  Semantics are this
    project's own: image and total pops against its oracle -- the default form, then one of the two opt-in forms
    (two launches, RT_EX_SPLIT=1, on even seeds; the wavefront form, RT_EX_WAVEFRONT=1, on odd ones)."""
    rng = np.random.default_rng(47000 + seed)
    desc, W, H, K, cam_pose, info = _adversarial_scene(scenes, rng, shiny=True)
    opts = (int(rng.integers(1, 9)), int(rng.integers(0, 5)), int(rng.integers(0, 2)))
    print("seed", seed, info, "spp / bounces / lighting", opts)
    _compare_ex(rt, orc, desc, W, H, K, cam_pose, *opts, threads=4)
    monkeypatch.setenv("RT_EX_SPLIT" if seed % 2 == 0 else "RT_EX_WAVEFRONT", "1")
    _compare_ex(rt, orc, desc, W, H, K, cam_pose, *opts, threads=4)


@pytest.mark.parametrize("seed", range(_ADV_FIRST, _ADV_FIRST + max(4, int(os.environ.get("RT_FUZZ_ADV_SEEDS", 12)) // 3)))
def test_fuzz_adversarial_refit_and_rebuild(rt, orc, scenes, seed):
    """An awkward scene whose first mesh is then (a) REFITTED to the triangles of another awkward mesh of the same count (the tree keeps its
    topology, every node gets the bounds of whatever its triangles became: flat, infinite, NaN -- the mesh flag of the octant loops is
    decided anew), (b) REBUILT on the device from a third one with at most as many triangles, and whose instances (c) all get another awkward
    pose and scale: all planes against the oracle after each step (orc_mesh_refit on the same tree; a fresh oracle mesh for the rebuild)."""
    import orc as orc_mod
    o = orc_mod.oracle()
    rng = np.random.default_rng(59000 + seed)
    desc, W, H, K, cam_pose, info = _adversarial_scene(scenes, rng)
    n0 = len(desc.meshes[0][1])
    kind_b, b = _adversarial_mesh(o, rng)
    b = b[np.arange(n0) % len(b)]                               # the same count, in some order (a refit is any new set of n triangles)
    b_kept = b.copy()                                           # ... of which vertices and normals count: texture coordinates stay (rt_hip.h),
    b_kept[:, 12:18] = desc.meshes[0][1][:, 12:18]              # whatever the caller's triangles carry (here: other random values)
    kind_c, c = _adversarial_mesh(o, rng)
    c = c[:n0]
    print("seed", seed, info, "refit to", kind_b, "rebuild from", kind_c, len(c))
    cam = rt.Camera(W, H, K, scenes.D_REF)
    cam.set_pose(cam_pose)

    def check(sp, so, what):
        ref = so.render(W, H, K, scenes.D_REF, cam_pose, threads=4)
        dbg = rt.render_debug(sp, cam)
        for n in ("img",) + PLANES:
            assert np.array_equal(dbg[n], ref[n]), (what, n, int((dbg[n] != ref[n]).sum()))
        ids = rt.render_ids(sp, cam)
        assert np.array_equal(ids["img"], ref["img"]) and np.array_equal(ids["hit_tri"], ref["hit_tri"]) and np.array_equal(ids["hit_inst"], ref["hit_inst"]), what

    so = desc.build_oracle(orc)
    sp = desc.build_product(rt, gpu_build=seed % 2 == 1)
    sp.upload_to_device()
    check(sp, so, "as uploaded")
    sp.refit_mesh(0, b)
    o.mesh_refit(desc.oracle_meshes[0], b_kept)
    check(sp, so, "refitted")
    sp.upload_to_device()                                       # the host copy was refitted too, and kept its texture coordinates as well
    check(sp, so, "uploaded again after the refit")
    so.close()
    sp.rebuild_mesh(0, c)
    desc_c = sd.SceneDesc(desc.materials, [("tris", c)] + list(desc.meshes[1:]), desc.instances)
    so = desc_c.build_oracle(orc)
    check(sp, so, "rebuilt")
    # (c) every instance gets another awkward form (Scene::update_mesh_instance, Scene.cpp:67-74), alternately synchronising and ordered
    # on the default stream: the flags the loops branch on (identity / unit inverse pose) are decided anew per update
    other = _adversarial_scene(scenes, np.random.default_rng(61000 + seed))[0].instances
    for i, (mesh, mat, _, _) in enumerate(desc.instances):
        pose, scale = other[i % len(other)][2], other[i % len(other)][3]
        sp.update_mesh_instance(i, mesh, mat, pose, scale, stream=False if i % 2 == 0 else None)
        so.update_instance(i, mesh, mat, pose, scale)
    check(sp, so, "instances updated")
    so.close()


@pytest.mark.parametrize("seed", range(_ADV_FIRST, _ADV_FIRST + max(4, int(os.environ.get("RT_FUZZ_ADV_SEEDS", 12)) // 3)))
def test_fuzz_adversarial_api_sequences(rt, orc, scenes, seed):
    """Stateful differential fuzzing of the boundary: a random SEQUENCE of the calls that change an uploaded scene -- refit of a mesh
    (other texture coordinates passed on purpose: they stay), rebuild of a mesh on the device (fewer triangles, as many, or more than it
    was uploaded with: the scene is uploaded again), update of an instance (pose, scale, mesh and material; synchronising or ordered on
    the default stream), upload_to_device() again -- over awkward scenes, and after EVERY call a frame through a randomly chosen
    entry point against the oracle, which mirrors the sequence (orc_mesh_refit on the same tree, a fresh mesh for a rebuild): the
    instrumented kernel's planes, the production kernel's hit ids, a batch of four frames with two poses (view records), stripes of
    2-5 virtual ranks put back by rt_unstripe, two default-stream frames into two images (rt_render_overlapped)."""
    import ctypes as C
    import orc as orc_mod
    o = orc_mod.oracle()
    h = rt.libs()[0]
    rng = np.random.default_rng(67000 + seed)
    desc, W, H, K, cam_pose, info = _adversarial_scene(scenes, rng)
    pose2 = tuple(np.asarray(cam_pose, np.float64) + np.concatenate([rng.uniform(-0.2, 0.2, 3), rng.uniform(-0.1, 0.1, 3)]))
    # per mesh: its triangles, the oracle's mesh, how many triangles its part of the device arrays holds.  An upload never changes the
    # tree a mesh is rendered with: the host copy of a mesh that was rebuilt on the device and refitted since builds, when it is needed
    # at last, the tree of the rebuild refitted -- not a new one over the moved triangles (MeshPrimitive::refit / sync_tree), which
    # would report another one of several exactly coincident triangles
    meshes = [dict(tris=t.copy(), h=o.mesh_from_triangles(t), cap=len(t)) for _, t in desc.meshes]

    def uploaded_again():
        for m in meshes:
            m["cap"] = len(m["tris"])
    instances = [list(i) for i in desc.instances]
    mats = desc.materials
    sp = desc.build_product(rt, gpu_build=bool(rng.integers(2)))
    sp.upload_to_device()
    cam = rt.Camera(W, H, K, scenes.D_REF)
    log = [info]

    def oracle_frames(poses):
        so = orc_mod.OracleScene(o)
        for m in mats:
            so.add_material(m[0], m[1])
        for m in meshes:
            so.add_mesh(m["h"])
        for mesh, mat, pose, scale in instances:
            so.add_instance(mesh, mat, pose, scale)
        out = [so.render(W, H, K, scenes.D_REF, p, threads=4) for p in poses]
        so.close()
        return out

    def check(step):
        via = str(rng.choice(["planes", "ids", "batch", "stripes", "default_stream"]))
        log.append("check via " + via)
        what = "seed %d step %d: %s" % (seed, step, "; ".join(log))
        print(what.split("; ")[-2] if step else what, "|", log[-1])
        cam.set_pose(cam_pose)
        if via == "planes":
            ref, = oracle_frames([cam_pose])
            dbg = rt.render_debug(sp, cam)
            for n in ("img",) + PLANES:
                assert np.array_equal(dbg[n], ref[n]), (n, what)
        elif via == "ids":
            ref, = oracle_frames([cam_pose])
            ids = rt.render_ids(sp, cam)
            for n in ("img", "hit_tri", "hit_inst"):
                assert np.array_equal(ids[n], ref[n]), (n, what)
        elif via == "batch":
            ref = oracle_frames([cam_pose, pose2])
            bufs = [rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(4)]
            cam.render_scene_batch(sp, [cam_pose, pose2, pose2, cam_pose], [b.ptr for b in bufs], bufs[0].pitch, synchronize=True)
            for k, b in enumerate(bufs):
                assert np.array_equal(b.to_host().reshape(H, W, 3), ref[(0, 1, 1, 0)[k]]["img"]), ("frame %d" % k, what, sp.view_stats())
                b.free()
        elif via == "stripes":
            ref, = oracle_frames([cam_pose])
            nr, stripe = int(rng.integers(2, 6)), int(rng.choice([3, 4, 8, 16]))
            rows = []
            for r in range(nr):
                n = C.c_int32(0)
                rt.check(h.rt_stripe_rows(H, stripe, r, nr, C.byref(n)))
                rows.append(n.value)
            maxr, pitch = max(max(rows), 1), W * 3
            gathered = rt.DeviceBuffer(nbytes=nr * maxr * pitch)
            for r in range(nr):
                cam.render_scene_stripes(sp, gathered.ptr.value + r * maxr * pitch, pitch, stripe, r, nr, synchronize=True)
            out = rt.DeviceBuffer(width_bytes=W * 3, height=H)
            rt.check(h.rt_unstripe(gathered.ptr, pitch, maxr * pitch, out.ptr, out.pitch, W, H, stripe, nr, None))
            rt.check(h.rt_device_synchronize())
            assert np.array_equal(out.to_host().reshape(H, W, 3), ref["img"]), ("%d ranks, %d-row stripes" % (nr, stripe), what)
            gathered.free(); out.free()
        else:
            ref = oracle_frames([cam_pose, pose2])
            a, b = rt.DeviceBuffer(width_bytes=W * 3, height=H), rt.DeviceBuffer(width_bytes=W * 3, height=H)
            cam.render_scene(sp, a.ptr, a.pitch)                   # the reference's loop: two frames on the default stream, then a synchronise
            cam.set_pose(pose2)
            cam.render_scene(sp, b.ptr, b.pitch)
            rt.check(h.rt_device_synchronize())
            assert np.array_equal(a.to_host().reshape(H, W, 3), ref[0]["img"]) and np.array_equal(b.to_host().reshape(H, W, 3), ref[1]["img"]), what
            a.free(); b.free()

    check(0)
    for step in range(1, 7):
        op = str(rng.choice(["refit", "rebuild", "instance", "instance", "upload"]))
        if op == "refit":
            i = int(rng.integers(len(meshes)))
            n = len(meshes[i]["tris"])
            kind, t = _adversarial_mesh(o, rng)
            t = t[np.arange(n) % len(t)]
            kept = t.copy()
            kept[:, 12:18] = meshes[i]["tris"][:, 12:18]
            log.append("refit mesh %d to %s" % (i, kind))
            sp.refit_mesh(i, t)
            o.mesh_refit(meshes[i]["h"], kept)
            meshes[i]["tris"] = kept
        elif op == "rebuild":
            i = int(rng.integers(len(meshes)))
            kind, t = _adversarial_mesh(o, rng)
            n = len(meshes[i]["tris"])
            if rng.random() < 0.75:
                t = t[:max(1, int(rng.integers(1, n + 1)))]
            log.append("rebuild mesh %d from %s, %d -> %d triangles (room for %d)" % (i, kind, n, len(t), meshes[i]["cap"]))
            sp.rebuild_mesh(i, t)
            grows = len(t) > meshes[i]["cap"]
            meshes[i] = dict(tris=t.copy(), h=o.mesh_from_triangles(t), cap=meshes[i]["cap"])
            if grows:                                               # more triangles than its part of the arrays holds: the scene is uploaded again
                uploaded_again()
        elif op == "instance":
            i = int(rng.integers(len(instances)))
            src = _adversarial_scene(scenes, np.random.default_rng(int(rng.integers(1 << 30))))[0].instances[0]
            new = [int(rng.integers(len(meshes))), int(rng.integers(len(mats))), src[2], src[3]]
            log.append("instance %d -> mesh %d material %d pose %s scale %s" % (i, new[0], new[1], new[2], new[3]))
            sp.update_mesh_instance(i, *new, stream=False if rng.random() < 0.5 else None)
            instances[i] = new
        else:
            log.append("upload again")
            sp.upload_to_device()
            uploaded_again()
        check(step)


def test_million_triangle_mesh(rt, orc, scenes, tmp_path):
    """Scale check: a 999 680-triangle blob (1.9 M BVH nodes, 32 levels -- the builder's depth cap, so deep leaves hold
    several triangles and the traversal stack spills).  GPU-built tree == host-built tree; a 24-row band of the 1080p
    frame matches the oracle on all planes; the working set (157 MB) no longer fits the L2s."""
    p = str(tmp_path / "blob1m.obj")
    assert scenes.write_blob_obj(p, 710, 705) == 999680
    host = rt.Mesh.load_obj(p)
    dev = rt.Mesh.load_obj(p, gpu_build=True)
    assert host.num_nodes == dev.num_nodes and host.max_level == 32
    _same_tree(dev.dump(), host.dump())
    sp = rt.Scene()
    sp.add_material(scenes.C2["albedo"])
    sp.add_mesh(dev)
    sp.add_mesh_instance(0, 0)
    sp.upload_to_device()
    W, H = 1920, 1080
    K, pose = scenes.scaled_K(W), scenes.C2_CAMERAS["mid"]
    cam = rt.Camera(W, H, K, scenes.D_REF)
    cam.set_pose(pose)
    dbg = rt.render_debug(sp, cam)
    assert np.array_equal(rt.render(sp, cam), dbg["img"])
    o = orc.oracle()
    so = orc.OracleScene(o)
    so.add_material(scenes.C2["albedo"])
    so.add_mesh(o.obj_load(p))
    so.add_instance(0, 0)
    y0 = 520
    ref = so.render(W, H, K, scenes.D_REF, pose, y0=y0, y1=y0 + 24, threads=32)
    for k in ("img",) + PLANES:
        assert np.array_equal(dbg[k][y0:y0 + 24], ref[k][y0:y0 + 24]), k
    so.close()
