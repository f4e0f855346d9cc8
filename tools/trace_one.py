#!/usr/bin/env python3
"""Diagnostic: render one frame per C2 camera with RT_TRACE_FILE set and print per-wave lifetimes, residency over time and,
with RT_TRACE_PROF=1 in the environment, per-phase cycle shares of the stamped kernel variant.
   python tools/trace_one.py <outdir> [spp,bounces,lighting]      (the optional triple traces the extension kernel)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
rt = importlib.import_module("cuda-raytracing_amd")
scenes = importlib.import_module("cuda-raytracing_amd.scenes")
out = sys.argv[1]
ex = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else None
mesh = rt.Mesh.load_obj(os.path.join(ROOT, ".scene_cache", "blob70k.obj"))
scene = rt.Scene(); scene.add_material(scenes.C2["albedo"], roughness=0.05 if ex else 0.0, metallic=0.4 if ex else 0.0); scene.add_mesh(mesh); scene.add_mesh_instance(0, 0); scene.upload_to_device()
W, H = 1920, 1080
img = rt.DeviceBuffer(width_bytes=W * 3, height=H)
for name, pose in scenes.C2_CAMERAS.items():
    cam = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF); cam.set_pose(pose)
    if ex: cam.set_options(*ex)
    for _ in range(1 if ex else 12): cam.render_scene(scene, img.ptr, img.pitch, synchronize=True)    # (12: the heavy-first order has settled)
    os.environ["RT_TRACE_FILE"] = os.path.join(out, "trace_%s.bin" % name)
    cam.render_scene(scene, img.ptr, img.pitch, synchronize=True)
    del os.environ["RT_TRACE_FILE"]
    t = np.fromfile(os.path.join(out, "trace_%s.bin" % name), np.uint64).reshape(-1, 16)
    t = t[t[:, 1] > 0]
    st, en = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
    t0 = st.min(); dur = (en - st) / 100.0; span = (en.max() - t0) / 100.0
    print(name, "waves", len(t), "span %.1f us" % span, "wave dur mean %.2f p50 %.2f p99 %.2f max %.2f us" % (dur.mean(), np.median(dur), np.percentile(dur, 99), dur.max()),
          "sum dur %.0f us -> avg concurrency %.0f" % (dur.sum(), dur.sum() / span))
    # concurrency over time in 10 slices
    edges = np.linspace(0, span, 11)
    conc = [(((st - t0) / 100.0 < b) & ((en - t0) / 100.0 > a)).sum() for a, b in zip(edges[:-1], edges[1:])]
    print("   waves alive per tenth:", conc)
    if t[:, 8].max() > 0:                                     # RT_TRACE_PROF=1: per-phase cycle stamps of the diagnostic kernel
        for w in np.argsort(dur)[-3:]:
            cp, cm, ci, cl, nit, nint, nleaf = [int(v) for v in t[w, 4:11]]
            print("   heavy wave: dur %.1f us iters %d (with interior lanes %d, with leaf lanes %d) cycles: pop %d fetch %d interior %d leaf %d  per-iter %.0f"
                  % (dur[w], nit, nint, nleaf, cp, cm, ci, cl, (cp + cm + ci + cl) / max(nit, 1)))
        g = t[:, 11:14].astype(np.float64).sum(0)               # wave iterations whose lanes hold 1 / 2 / 3-4 different entries
        n_all = float(t[:, 8].astype(np.float64).sum())
        print("   different entries per wave iteration: one %.3f, two %.3f, three or four %.3f, more %.3f"
              % (g[0] / n_all, g[1] / n_all, g[2] / n_all, 1.0 - g.sum() / n_all))
        tot = t[:, 4:11].astype(np.float64).sum(0)
        print("   all waves: iters %.3g (interior %.3g, leaf %.3g); cycle shares pop %.2f fetch %.2f interior %.2f leaf %.2f; cycles/iter %.0f"
              % (tot[4], tot[5], tot[6], *(tot[:4] / tot[:4].sum()), tot[:4].sum() / tot[4]))
    if ex:                                                    # how much of the wave time is spent by waves of which length
        order = np.argsort(dur); cs = np.cumsum(dur[order]) / dur.sum()
        print("   wave dur deciles (us):", np.percentile(dur, [10, 30, 50, 70, 90, 95, 99]).round(0), " share of wave-time in the longest 10%% of waves: %.2f" % (1 - cs[int(0.9 * len(dur))]))
    late = np.argsort(en)[-5:]
    print("   last finishers: start %s end %s dur %s tile %s" % (((st[late]-t0)/100.0).round(1), ((en[late]-t0)/100.0).round(1), dur[late].round(1), t[late,3]))
