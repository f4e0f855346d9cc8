"""Round 6: the reference's own scene shape (kernel.cu:166-240: two textured instances, the second translated) at full size against
the oracle, and WHICH traversal loop ran (rt_scene_loop_stats): the hand-written gfx950 loop (trace_loop_asm, rt_kernels.hip) must
carry the benchmark scenes, and the parity tests above pass on any loop -- so the choice is asserted here."""
import numpy as np
import pytest

import scene_defs as sd
from test_gpu_parity import _compare

pytestmark = pytest.mark.gpu


def _loops(rt, scenes, sp, W, H, poses):
    cam = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF)
    bufs = [rt.DeviceBuffer(width_bytes=W * 3, height=H) for _ in poses]
    st = sp.loop_stats(cam, poses, [b.ptr for b in bufs], bufs[0].pitch)
    imgs = [b.to_host().reshape(H, W, 3) for b in bufs]
    for b in bufs:
        b.free()
    return st, imgs


@pytest.mark.parametrize("baked", [False, True])
def test_demo_scene_full_size(rt, orc, scenes, demo_objs, baked):
    """bench.py --workload demo [--baked] at 1920x1080: RGB, hit ids and visit counts of every pixel against the oracle; both textures on screen."""
    w = scenes.DEMO
    img, ref = _compare(rt, orc, sd.demo_scene(scenes, demo_objs, baked), w["width"], w["height"], scenes.scaled_K(w["width"]), w["D"], w["cam_pose"], threads=16)
    assert set(np.unique(ref["hit_inst"])) >= {0, 1}                    # the area and the board are both hit
    assert len(np.unique(img.reshape(-1, 3), axis=0)) > 200


def test_posed_and_baked_demo_frames_agree_on_what_is_hit(rt, scenes, demo_objs):
    """The translated board and its baked twin cover the same pixels (the fp32 vertices differ in the last bit at most, so a handful of
    silhouette pixels may): the twin is a fair comparison for the timing of the translated instance."""
    w = scenes.DEMO
    W, H = w["width"], w["height"]
    out = []
    for baked in (False, True):
        sp = sd.demo_scene(scenes, demo_objs, baked).build_product(rt)
        sp.upload_to_device()
        cam = rt.Camera(W, H, scenes.scaled_K(W), w["D"])
        cam.set_pose(w["cam_pose"])
        out.append(rt.render_ids(sp, cam)["hit_inst"])
    assert int((out[0] != out[1]).sum()) < 200


def test_loop_stats_c2_runs_the_handwritten_loop(rt, scenes, blob70k):
    """C2 (one identity instance, 28-level tree: the optimistic stack): every wave x instance cast on trace_loop_asm, nothing re-traced;
    single frames and batches (view records) alike; the instrumented copy writes the production frame."""
    c = scenes.C2
    W, H = c["width"], c["height"]
    sp = sd.blob_scene(scenes, blob70k).build_product(rt)
    sp.upload_to_device()
    for cam_name in ("mid", "far"):
        pose = scenes.C2_CAMERAS[cam_name]
        for n in (1, 4):
            st, imgs = _loops(rt, scenes, sp, W, H, [pose] * n)
            assert st["waves"] == n * ((W + 7) // 8) * ((H + 7) // 8), st
            # (the waves that are not: 8x8-pixel tiles on the image's centre column or row, where a direction component changes sign within
            # the wave -- one tile column plus one tile row per frame take the compiler's generic loop: 375 of 32 400)
            assert st["asm_loop_frac"] >= 0.985 and st["cpp_octant"] == 0, st
            assert st["cpp_generic"] <= n * ((W + 7) // 8 + (H + 7) // 8 + 8), st
            assert st["asm_posed"] == 0 and st["retraced_lanes"] == 0 and st["deep"] == 0, st
        cam = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF)
        cam.set_pose(pose)
        assert np.array_equal(imgs[0], rt.render(sp, cam))


def test_loop_stats_demo_translated_instance_on_the_handwritten_loop(rt, scenes, demo_objs):
    """The demo's second instance is translated: round 5 sent it to the compiler's loop (the hand-written one was gated on identity
    instances); now both instances run trace_loop_asm, and a translation does not need the out-of-line transform."""
    w = scenes.DEMO
    W, H = w["width"], w["height"]
    sp = sd.demo_scene(scenes, demo_objs).build_product(rt)
    sp.upload_to_device()
    st, _ = _loops(rt, scenes, sp, W, H, [w["cam_pose"]] * 4)
    waves = 4 * ((W + 7) // 8) * ((H + 7) // 8)
    assert st["waves"] == waves and st["asm"] + st["cpp_octant"] + st["cpp_generic"] == 2 * waves, st
    assert st["asm_loop_frac"] >= 0.985 and st["asm_posed"] == 0 and st["cpp_octant"] == 0, st


def test_loop_stats_posed_and_exact_uv_instances(rt, orc, scenes, blob5k):
    """The multi-instance parity scene: rotated / scaled instances take the hand-written loop's out-of-line transform (asm_posed); a
    mesh in the exact-uv mode stays on the compiler's loops."""
    m = sd.MULTI_CAMERA
    W, H = m["width"], m["height"]
    desc = sd.multi_instance_scene(scenes, blob5k)
    sp = desc.build_product(rt)
    sp.upload_to_device()
    st, imgs = _loops(rt, scenes, sp, W, H, [m["pose"]])
    waves = ((W + 7) // 8) * ((H + 7) // 8)
    assert st["waves"] == waves and st["asm"] + st["cpp_octant"] + st["cpp_generic"] == 3 * waves, st
    assert st["asm_posed"] > 0.9 * st["asm"] and st["asm_loop_frac"] > 0.8, st
    so = desc.build_oracle(orc)
    assert np.array_equal(imgs[0], so.render(W, H, scenes.scaled_K(W), scenes.D_REF, m["pose"], threads=8)["img"])
    so.close()
    # exact-uv: a uv coordinate beyond 1e37 switches the mesh to the mode that interpolates uv per candidate (raycast.cu:96)
    tris = sd.random_triangles(200, seed=5, spread=0.8, size=0.3)
    tris[7, 12] = np.float32(3e38)
    sx = sd.SceneDesc([((0.9, 0.5, 0.2), None)], [("tris", tris)], [(0, 0, (0,) * 6, (1, 1, 1))]).build_product(rt)
    sx.upload_to_device()
    st, _ = _loops(rt, scenes, sx, W, H, [m["pose"]])
    assert st["asm"] == 0 and st["cpp_octant"] + st["cpp_generic"] == waves, st
