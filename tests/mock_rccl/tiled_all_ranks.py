"""Child process of test_gpu_tiling.py::test_tiled_all_with_several_ranks_over_mock_rccl.  RT_RCCL_LIBRARY points rt_comm at
the in-process mock (mock_rccl.cpp), so rt_comm_init_all can create N > 1 ranks on the one GPU of the box, and
rt_render_tiled_all runs its real N-rank code: per-rank stripes, scratch sizing, the grouped gathers, the un-stripe on the
root.  Every frame must equal the plain single-GPU render of the same pose."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scene_defs as sd                                          # noqa: E402

rt = importlib.import_module("cuda-raytracing_amd")
scenes = importlib.import_module("cuda-raytracing_amd.scenes")


def main(blob):
    h = rt.libs()[0]
    v = C.c_int32(0)
    rt.check(h.rt_comm_available(C.byref(v)), "rt_comm_available")
    assert v.value == 99999, "the mock was not loaded"
    W, H = 322, 203
    cases = 0
    for n in (2, 3, 8):
        replicas = []
        for _ in range(n):
            sp = sd.shiny_scene(scenes, blob).build_product(rt)
            sp.upload_to_device()
            replicas.append(sp)
        comms = (C.c_void_p * n)()
        rt.check(h.rt_comm_init_all((C.c_int32 * n)(*([0] * n)), n, comms), "rt_comm_init_all")
        scn = (C.c_void_p * n)(*[sp.device_handle for sp in replicas])
        for r in range(n):
            rk, nr = C.c_int32(-1), C.c_int32(-1)
            rt.check(h.rt_comm_info(comms[r], C.byref(rk), C.byref(nr), None))
            assert (rk.value, nr.value) == (r, n)
        cam = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF)
        for k, (opts, stripe, root) in enumerate([((1, 0, 0), 16, 0), ((1, 0, 0), 7, n - 1), ((4, 2, 1), 16, n // 2), ((1, 0, 0), 64, 0)]):
            pose = list(sd.SHINY_CAMERA["pose"])
            pose[0] += 0.05 * k                                   # another frame every time: stale scratch cannot pass
            cam.set_pose(pose)
            cam.set_options(*opts)
            want = rt.render(replicas[0], cam)
            img = rt.DeviceBuffer(width_bytes=W * 3, height=H)
            p = cam.params()
            o = (C.c_int32 * 3)(*opts)
            rt.check(h.rt_render_tiled_all(scn, comms, n, C.byref(p), o, img.ptr, img.pitch, stripe, root, None, 1), "rt_render_tiled_all")
            got = img.to_host().reshape(H, W, 3)
            assert np.array_equal(got, want), (n, opts, stripe, root, int((got != want).any(axis=2).sum()))
            cases += 1
        # frames alternated between two streams on the same communicators, no host wait in between: every call orders itself
        # behind the previous call's use of the communicator's scratch (an event), so all frames must come out right
        if n == 3:
            import torch
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            cam.set_options(1, 0, 0)
            outs, wants = [], []
            for k in range(6):
                pose = list(sd.SHINY_CAMERA["pose"])
                pose[1] += 0.07 * k
                cam.set_pose(pose)
                wants.append(rt.render(replicas[0], cam))
                outs.append(rt.DeviceBuffer(width_bytes=W * 3, height=H))
                pk = cam.params()
                sarr = (C.c_void_p * n)(*([streams[k & 1].cuda_stream] * n))
                rt.check(h.rt_render_tiled_all(scn, comms, n, C.byref(pk), None, outs[k].ptr, outs[k].pitch, 16, k % n, sarr, 0), "rt_render_tiled_all (two streams)")
            torch.cuda.synchronize()
            for k in range(6):
                assert np.array_equal(outs[k].to_host().reshape(H, W, 3), wants[k]), ("two streams", k)
            cases += 1
        # argument checks of the N-rank form
        assert h.rt_render_tiled_all(scn, comms, n, C.byref(p), None, img.ptr, img.pitch, 16, n, None, 1) == -1       # no such root
        assert h.rt_render_tiled_all(scn, comms, n - 1, C.byref(p), None, img.ptr, img.pitch, 16, 0, None, 1) == -1   # not the whole communicator
        for r in range(n):
            rt.check(h.rt_comm_destroy(comms[r]))
    cases += rotating_groups(h, blob)
    cases += exchange_objects(h, blob)
    print("OK", cases)


def exchange_objects(h, blob):
    """The same two exchanges through the objects bench.py drives (tiling.RcclExchange on rt.Comm, torch device tensors, torch's
    current stream): N ranks' to_root() and rotating() calls inside one group, checked against what the exchange is defined
    to deliver -- gathered[r] = rank r's local buffer; received (source-major) = every source's block for this rank."""
    import torch
    tiling = importlib.import_module("cuda-raytracing_amd.tiling")
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device="cpu").manual_seed(5)
    cases = 0
    for n, count, max_rows, pitch in ((2, 4, 5, 96), (3, 5, 7, 33), (8, 3, 2, 48), (8, 32, 3, 24)):
        comms = rt.Comm.init_all([0] * n)
        ex = [tiling.RcclExchange(c) for c in comms]
        slots, counts, offsets, real = tiling.rotating_plan(count, n)
        local = [torch.randint(0, 256, (slots * max_rows, pitch), dtype=torch.uint8, generator=gen).to(dev) for _ in range(n)]
        # every frame of a group to rank `root`
        for root in (0, n - 1):
            gathered = [torch.zeros((n, slots * max_rows, pitch), dtype=torch.uint8, device=dev) if r == root else None for r in range(n)]
            rt.Comm.group_start()
            for r in range(n):
                ex[r].to_root(local[r], gathered[r], root)
            rt.Comm.group_end()
            torch.cuda.synchronize()
            for r in range(n):
                assert torch.equal(gathered[root][r], local[r]), ("to_root", n, root, r)
        # rotating root: slot s of every rank to the rank that assembles it
        received = [torch.zeros((n * counts[r] * max_rows, pitch), dtype=torch.uint8, device=dev) for r in range(n)]
        for rep in range(2):                                      # (the second call takes the cached plan)
            rt.Comm.group_start()
            for r in range(n):
                ex[r].rotating(local[r], received[r], count, max_rows)
            rt.Comm.group_end()
        torch.cuda.synchronize()
        for d in range(n):
            got = received[d].reshape(n, counts[d] * max_rows, pitch)
            for src in range(n):
                want = local[src][offsets[d] * max_rows:(offsets[d] + counts[d]) * max_rows]
                assert torch.equal(got[src], want), ("rotating", n, count, d, src)
        for c in comms:
            c.close()
        cases += 1
    return cases


def rotating_groups(h, blob):
    """The rotating-root exchange of a group of frames (bench.py's stream mode with N ranks): every rank renders its stripes of
    all frames of the group, ONE rt_all_to_all per rank moves frame slot s to its assembling rank, every rank un-stripes the
    frames it owns.  Plan and buffer layout are tiling.rotating_plan's, as tiling.RcclExchange passes them."""
    tiling = importlib.import_module("cuda-raytracing_amd.tiling")
    W, H, stripe = 322, 203, 16
    row = W * 3
    sz = C.c_size_t
    cases = 0
    for n, count in ((2, 4), (3, 5), (8, 3), (4, 1)):
        comms = (C.c_void_p * n)()
        rt.check(h.rt_comm_init_all((C.c_int32 * n)(*([0] * n)), n, comms), "rt_comm_init_all")
        sp = sd.shiny_scene(scenes, blob).build_product(rt)
        sp.upload_to_device()
        cam = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF)
        max_rows = tiling.stripe_rows(H, stripe, 0, n)
        slots, counts, offsets, real = tiling.rotating_plan(count, n)
        poses = []
        for f in range(count):
            pose = list(sd.SHINY_CAMERA["pose"])
            pose[0] += 0.04 * f + 0.01 * n
            poses.append(pose)
        local = [rt.DeviceBuffer(nbytes=slots * max_rows * row) for _ in range(n)]
        received = [rt.DeviceBuffer(nbytes=n * counts[r] * max_rows * row) for r in range(n)]
        for r in range(n):
            cam.render_scene_stripes_batch(sp, poses, tiling.batch_local_ptrs(local[r].ptr.value, count, max_rows, row), row, stripe, r, n)
        rt.check(h.rt_group_start(), "rt_group_start")
        for r in range(n):
            rt.check(h.rt_all_to_all(comms[r], local[r].ptr, (sz * n)(*[c * max_rows * row for c in counts]),
                                     (sz * n)(*[o * max_rows * row for o in offsets]), received[r].ptr,
                                     (sz * n)(*([counts[r] * max_rows * row] * n)), (sz * n)(*[q * counts[r] * max_rows * row for q in range(n)]), None),
                     "rt_all_to_all")
        rt.check(h.rt_group_end(), "rt_group_end")
        assert sum(real) == count
        for d in range(n):
            for j in range(real[d]):
                src, rank_stride = tiling.batch_unstripe_args(received[d].ptr.value, j, counts[d], max_rows, row)
                img = rt.DeviceBuffer(width_bytes=row, height=H)
                rt.check(h.rt_unstripe(src, row, rank_stride, img.ptr, img.pitch, W, H, stripe, n, None), "rt_unstripe")
                rt.check(h.rt_device_synchronize())
                cam.set_pose(poses[offsets[d] + j])
                want = rt.render(sp, cam)
                got = img.to_host().reshape(H, W, 3)
                assert np.array_equal(got, want), (n, count, d, j, int((got != want).any(axis=2).sum()))
        for r in range(n):
            rt.check(h.rt_comm_destroy(comms[r]))
        cases += 1
    return cases


if __name__ == "__main__":
    main(sys.argv[1])
