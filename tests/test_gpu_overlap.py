"""Camera::render_scene on the default stream without `synchronize` -- the reference's own call shape (two renders into two
images, then one cudaDeviceSynchronize: kernel.cu:277-279) -- goes through rt_render_overlapped: consecutive frames alternate
between two blocking streams the scene owns.  What a caller of the reference's API may rely on must still hold: the frame is
ordered like a default-stream launch against the caller's own default-stream work, later frames into the same image win,
scene updates issued between two frames separate them, and every frame equals the oracle's."""
import ctypes as C

import numpy as np
import pytest

import scene_defs as sd

pytestmark = pytest.mark.gpu


def _scene(rt, scenes, obj, instances=None):
    d = sd.SceneDesc([(scenes.C2["albedo"], None)], [("obj", obj)], instances or [(0, 0, (0,) * 6, (1, 1, 1))])
    sp = d.build_product(rt)
    sp.upload_to_device()
    return d, sp


def _poses(scenes, n, cam="mid"):
    base = scenes.C2_CAMERAS[cam]
    return [(base[0] + 0.01 * k, base[1], base[2] + 0.004 * k, base[3] + 0.003 * k, base[4], base[5]) for k in range(n)]


def test_reference_loop_two_images_per_synchronise(rt, orc, scenes, blob70k):
    """kernel.cu:275-279: pose, render into d_img, render into d_img2, synchronise -- 12 rounds, every frame its own pose;
    both images after every synchronise against rt_render_debug, the first and last pair against the oracle."""
    W, H = 960, 540
    K = scenes.scaled_K(W)
    d, sp = _scene(rt, scenes, blob70k)
    so = d.build_oracle(orc)
    cam = rt.Camera(W, H, K, scenes.D_REF)                      # no stream set: the default stream
    imgs = [rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(2)]
    poses = _poses(scenes, 24)
    hlib = rt.libs()[0]
    before = sp.overlap_stats()
    for r in range(12):
        for b in range(2):
            cam.set_pose(poses[2 * r + b])
            cam.render_scene(sp, imgs[b].ptr, imgs[b].pitch)
        rt.check(hlib.rt_device_synchronize())
        for b in range(2):
            got = imgs[b].to_host().reshape(H, W, 3)
            cam.set_pose(poses[2 * r + b])
            assert np.array_equal(got, rt.render_debug(sp, cam)["img"]), (r, b)
            if r in (0, 11):
                assert np.array_equal(got, so.render(W, H, K, scenes.D_REF, poses[2 * r + b], threads=8, planes=False)["img"]), (r, b)
    after = sp.overlap_stats()
    assert after[0] - before[0] == 24 and after[1] == before[1]      # all 24 frames took the overlapped path; none had to wait for the other stream
    so.close()


def test_same_image_twice_second_frame_wins(rt, scenes, blob70k):
    """Two asynchronous renders into ONE image must not run side by side: the later call's frame is what the image holds.  The
    first pose is the expensive one (full coverage), the second a sky-heavy one that would finish first if the two overlapped."""
    W, H = 1920, 1080
    K = scenes.scaled_K(W)
    _, sp = _scene(rt, scenes, blob70k)
    cam = rt.Camera(W, H, K, scenes.D_REF)
    img = rt.DeviceBuffer(width_bytes=W * 3, height=H)
    other = rt.DeviceBuffer(width_bytes=W * 3, height=H)
    slow, fast = scenes.C2_CAMERAS["mid"], scenes.C2_CAMERAS["far"]
    cam.set_pose(fast)
    want = rt.render_debug(sp, cam)["img"]
    for k in range(10):
        if k % 3 == 2:                                          # a frame elsewhere in between moves the alternation along
            cam.set_pose(slow)
            cam.render_scene(sp, other.ptr, other.pitch)
        cam.set_pose(slow)
        cam.render_scene(sp, img.ptr, img.pitch)
        cam.set_pose(fast)
        cam.render_scene(sp, img.ptr, img.pitch)
        assert np.array_equal(img.to_host().reshape(H, W, 3), want), k          # (to_host: a default-stream copy, no synchronise before it)


def test_overlapping_images_inside_one_allocation(rt, scenes, blob5k):
    """Images that share SOME rows (an application that renders into windows of one big buffer): whichever streams the
    earlier frames went to, a later frame is ordered after every earlier one it shares memory with."""
    W, H = 640, 360
    K = scenes.scaled_K(W)
    _, sp = _scene(rt, scenes, blob5k)
    cam = rt.Camera(W, H, K, scenes.D_REF)
    pitch = W * 3
    big = rt.DeviceBuffer(nbytes=pitch * H * 4)
    base = big.ptr.value
    poses = _poses(scenes, 40)
    frames = []
    for p in poses[:6]:
        cam.set_pose(p)
        frames.append(rt.render_debug(sp, cam)["img"])
    rng = np.random.default_rng(5)
    host = np.zeros((H * 4, W, 3), np.uint8)
    rt.check(rt.libs()[0].rt_memcpy_h2d(big.ptr, host.ctypes.data, host.nbytes, None))
    for k in range(60):
        row0 = int(rng.integers(0, 3 * H + 1))                  # any window of H rows, overlapping earlier ones at random
        f = int(rng.integers(0, 6))
        cam.set_pose(poses[f])
        cam.render_scene(sp, C.c_void_p(base + row0 * pitch), pitch)
        host[row0:row0 + H] = frames[f]                         # sequential semantics: later frames overwrite earlier ones
    assert np.array_equal(big.to_host().reshape(H * 4, W, 3), host)
    assert sp.overlap_stats()[0] >= 60


def test_default_stream_work_of_the_caller_is_ordered_with_the_frames(rt, scenes, blob70k):
    """render -> default-stream D2H copy (no synchronise) sees the frame; default-stream H2D fill -> render: the frame lands on
    top of the fill (its sky pixels too: every pixel is written), never under it."""
    W, H = 1920, 1080
    K = scenes.scaled_K(W)
    _, sp = _scene(rt, scenes, blob70k)
    cam = rt.Camera(W, H, K, scenes.D_REF)
    hlib = rt.libs()[0]
    imgs = [rt.DeviceBuffer(nbytes=W * 3 * H) for _ in range(2)]
    junk = np.full(W * 3 * H, 77, np.uint8)
    poses = _poses(scenes, 8)
    want = []
    for p in poses:
        cam.set_pose(p)
        want.append(rt.render_debug(sp, cam)["img"])
    for k, p in enumerate(poses):
        b = k & 1
        rt.check(hlib.rt_memcpy_h2d(imgs[b].ptr, junk.ctypes.data, junk.nbytes, None))     # default stream
        cam.set_pose(p)
        cam.render_scene(sp, imgs[b].ptr, W * 3)                                            # must come after the fill ...
        got = imgs[b].to_host()                                                             # ... and before this copy
        assert np.array_equal(got.reshape(H, W, 3), want[k]), k


def test_instance_update_between_two_frames(rt, orc, scenes, blob5k):
    """render A -> Scene::update_mesh_instance -> render B into the other image, no synchronise in between: A shows the old
    pose, B the new one (Scene.cpp:67-74 is a synchronous cudaMemcpy on the default stream; so is the stream-ordered form
    with a NULL stream)."""
    W, H = 1280, 720
    K = scenes.scaled_K(W)
    inst = [(0, 0, (0,) * 6, (1, 1, 1)), (0, 0, (0.9, 0.2, 0.1, 0.3, 0.0, 0.2), (0.6, 0.6, 0.6))]
    d, sp = _scene(rt, scenes, blob5k, inst)
    cam = rt.Camera(W, H, K, scenes.D_REF)
    cam.set_pose(scenes.C2_CAMERAS["far"])
    imgs = [rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(2)]
    moved = [(0.9 - 0.15 * k, 0.2, 0.1 + 0.05 * k, 0.3 + 0.1 * k, 0.0, 0.2) for k in range(1, 9)]
    want = []
    for pose in [inst[1][2]] + moved:
        sp.update_mesh_instance(1, 0, 0, pose, (0.6, 0.6, 0.6))
        want.append(rt.render_debug(sp, cam)["img"])
    assert not np.array_equal(want[0], want[1])
    sp.update_mesh_instance(1, 0, 0, inst[1][2], (0.6, 0.6, 0.6))
    for k, pose in enumerate(moved):
        a, b = imgs[k & 1], imgs[(k & 1) ^ 1]
        cam.render_scene(sp, a.ptr, a.pitch)                                    # the instance as it was
        sp.update_mesh_instance(1, 0, 0, pose, (0.6, 0.6, 0.6), stream=False if k % 2 == 0 else None)     # synchronous form / ordered on the NULL stream
        cam.render_scene(sp, b.ptr, b.pitch)                                    # the instance as it is now
        assert np.array_equal(a.to_host().reshape(H, W, 3), want[k]), ("before", k)
        assert np.array_equal(b.to_host().reshape(H, W, 3), want[k + 1]), ("after", k)
    # and the last frame against the oracle with the last pose
    od = sd.SceneDesc(d.materials, d.meshes, [inst[0], (0, 0, moved[-1], (0.6, 0.6, 0.6))])
    so = od.build_oracle(orc)
    assert np.array_equal(want[-1], so.render(W, H, K, scenes.D_REF, scenes.C2_CAMERAS["far"], threads=8, planes=False)["img"])
    so.close()


def test_synchronous_and_explicit_stream_calls_keep_their_path(rt, scenes, blob5k):
    """synchronize = true and a camera with its own stream do not take the overlapped path (Camera.cu:38-39 semantics as
    before), and mix with it: an overlapped frame, then a synchronous frame into the same image -- the synchronous one wins."""
    W, H = 640, 360
    K = scenes.scaled_K(W)
    _, sp = _scene(rt, scenes, blob5k)
    cam = rt.Camera(W, H, K, scenes.D_REF)
    img = rt.DeviceBuffer(width_bytes=W * 3, height=H)
    p0, p1 = _poses(scenes, 2)
    cam.set_pose(p1)
    want = rt.render_debug(sp, cam)["img"]
    n0 = sp.overlap_stats()[0]
    for _ in range(5):
        cam.set_pose(p0)
        cam.render_scene(sp, img.ptr, img.pitch)
        cam.set_pose(p1)
        cam.render_scene(sp, img.ptr, img.pitch, synchronize=True)
        assert np.array_equal(img.to_host().reshape(H, W, 3), want)
    assert sp.overlap_stats()[0] - n0 == 5


def test_refit_and_rebuild_between_two_frames(rt, orc, scenes, blob5k):
    """render A -> Scene::refit_mesh / Scene::rebuild_mesh on the default stream -> render B into the other image, no synchronise
    by the caller in between: A shows the mesh as it was, B as it is now (both calls order their kernels on the NULL stream, which
    waits for the frames in flight on the scene's two blocking streams and holds back the frames issued after it)."""
    import orc as orc_mod
    o = orc_mod.oracle()
    W, H = 1280, 720
    K, pose = scenes.scaled_K(W), scenes.C2_CAMERAS["mid"]
    rest = rt.Mesh.load_obj(blob5k).dump()["tris"].copy()

    def deformed(step):
        m = rest.copy()
        v = m[:, :9].reshape(-1, 3, 3)
        v[..., 2] += np.float32(0.08 * step) * np.sin(4.0 * v[..., 0] + step)
        for i in range(len(m)):
            m[i, :12] = o.tri_from_vertices(m[i, :9])[:12]
        return m

    sp = rt.Scene()
    sp.add_material(scenes.C2["albedo"])
    sp.add_mesh(rt.Mesh.from_triangles(rest, gpu_build=True))
    sp.add_mesh_instance(0, 0)
    sp.upload_to_device()
    cam = rt.Camera(W, H, K, scenes.D_REF)
    cam.set_pose(pose)
    imgs = [rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in range(2)]
    shapes = [rest] + [deformed(k) for k in range(1, 5)]
    want = []
    for k, m in enumerate(shapes):                               # reference frames: each shape rendered on its own, synchronously
        (sp.refit_mesh if k % 2 else sp.rebuild_mesh)(0, m)
        want.append(rt.render_debug(sp, cam)["img"])
    assert not np.array_equal(want[0], want[1])
    sp.rebuild_mesh(0, shapes[0])
    for k in range(1, len(shapes)):
        a, b = imgs[k & 1], imgs[(k & 1) ^ 1]
        cam.render_scene(sp, a.ptr, a.pitch)                                    # the mesh as it was
        (sp.refit_mesh if k % 2 else sp.rebuild_mesh)(0, shapes[k])             # refit: same topology; rebuild: a new tree
        cam.render_scene(sp, b.ptr, b.pitch)                                    # the mesh as it is now
        assert np.array_equal(a.to_host().reshape(H, W, 3), want[k - 1]), ("before", k)
        assert np.array_equal(b.to_host().reshape(H, W, 3), want[k]), ("after", k)
    so = sd.SceneDesc([(scenes.C2["albedo"], None)], [("tris", shapes[-1])], [(0, 0, (0,) * 6, (1, 1, 1))]).build_oracle(orc)
    assert np.array_equal(want[-1], so.render(W, H, K, scenes.D_REF, pose, threads=8, planes=False)["img"])
    so.close()
