#!/usr/bin/env python3
"""Where the primary kernel's idle lanes come from, and what re-packing rays between the waves of a workgroup could gain at
most.  Runs on the GPU box: the visit-count planes come from the instrumented kernel (rt_render_debug).

A lane's loop iterations in render_kernel = its interior-node pops + one per triangle of every leaf it visits (an empty
leaf takes one).  A wave (8x8 pixels) iterates max-over-lanes times; a workgroup is 2x2 waves.
  E_tail          = sum of lane iterations / (64 x sum of wave iterations): what lanes that finished early leave idle
  perfect re-pack = wave iterations if the 256 rays of a workgroup could be re-packed into waves at no cost after EVERY
                    iteration: sum over t of ceil(active(t) / 64) -- the upper bound of any intra-workgroup tail re-packing
   python tools/lane_loss.py [width height]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
rt = importlib.import_module("cuda-raytracing_amd")
scenes = importlib.import_module("cuda-raytracing_amd.scenes")
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
mesh = rt.Mesh.load_obj(os.path.join(ROOT, ".scene_cache", "blob70k.obj"))
scene = rt.Scene(); scene.add_material(scenes.C2["albedo"]); scene.add_mesh(mesh); scene.add_mesh_instance(0, 0); scene.upload_to_device()
for name, pose in scenes.C2_CAMERAS.items():
    cam = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF); cam.set_pose(pose)
    r = rt.render_debug(scene, cam)
    pops, aabb, tris = (r[k].astype(np.int64) for k in ("pops", "aabb", "tris"))
    interior = aabb // 2
    leaf = pops - interior
    it = interior + np.maximum(tris, leaf)
    pad = np.zeros(((H + 15) // 16 * 16, (W + 15) // 16 * 16), np.int64); pad[:H, :W] = it
    wg = pad.reshape(pad.shape[0] // 16, 16, pad.shape[1] // 16, 16).transpose(0, 2, 1, 3).reshape(-1, 256)
    waves = wg.reshape(-1, 2, 8, 2, 8).transpose(0, 1, 3, 2, 4).reshape(-1, 4, 64)
    cur = waves.max(2).sum()
    ideal = sum((((wg > t).sum(1) + 63) // 64).sum() for t in range(int(wg.max())))
    print("%-4s per ray: interior pops %.2f, leaf pops %.2f, triangle tests %.2f -> %.1f iterations (%.0f %% at interior nodes); "
          "wave iterations %d; E_tail %.3f; perfect intra-workgroup re-pack %d wave iterations = %.3f of now (-%.1f %%)"
          % (name, interior.mean(), leaf.mean(), tris.mean(), it.mean(), 100.0 * interior.sum() / it.sum(), cur, it.sum() / (64.0 * cur),
             ideal, ideal / cur, 100.0 * (1 - ideal / cur)))
