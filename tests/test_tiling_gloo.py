"""N > 1 path on CPU: two gloo ranks render their stripes (with the oracle standing in for the GPU
kernel -- the stripe bookkeeping and the gather protocol are what is under test), gather to rank 0
with the helper bench.py's rehearsal mode uses (tiling.TorchExchange; the product path makes the same two calls on
tiling.RcclExchange = rt_gather / rt_all_to_all of the C-ABI), un-stripe, and compare with the full frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, blob, H, W, stripe, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import orc
    import scene_defs as sd
    tiling = importlib.import_module("cuda-raytracing_amd.tiling")
    scenes = importlib.import_module("cuda-raytracing_amd.scenes")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = sd.blob_scene(scenes, blob).build_oracle(orc)
    K, D, pose = scenes.scaled_K(W), scenes.D_REF, scenes.C2_CAMERAS["mid"]
    rows = tiling.frame_rows_of(H, stripe, rank, world)
    max_rows = max(tiling.stripe_rows(H, stripe, r, world) for r in range(world))
    local = torch.zeros((max_rows, W * 3), dtype=torch.uint8)
    # each rank renders only its own stripes
    for a in range(0, len(rows), stripe):
        y0 = int(rows[a])
        y1 = int(rows[min(a + stripe, len(rows)) - 1]) + 1
        img = s.render(W, H, K, D, pose, y0=y0, y1=y1, planes=False)["img"]
        local[a:a + (y1 - y0)] = torch.from_numpy(img[y0:y1].reshape(y1 - y0, W * 3))
    gathered = torch.zeros((world, max_rows, W * 3), dtype=torch.uint8) if rank == 0 else None
    tiling.TorchExchange(rank, world).to_root(local, gathered, root=0)
    if rank == 0:
        frame = tiling.unstripe_host(gathered.numpy(), H, stripe, world)
        full = s.render(W, H, K, D, pose, planes=False)["img"].reshape(H, W * 3)
        np.save(out_path, np.array([int(np.array_equal(frame, full)), int((frame != full).sum())]))
    dist.barrier()
    dist.destroy_process_group()


def _pipeline_worker(rank, world, port, blob, H, W, stripe, nframes, out_path):
    """bench.py's N > 1 frame loop (tiling.StripePipeline) over gloo; every frame uses a different camera so a
    buffer mix-up between in-flight frames cannot go unnoticed."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import orc
    import scene_defs as sd
    tiling = importlib.import_module("cuda-raytracing_amd.tiling")
    scenes = importlib.import_module("cuda-raytracing_amd.scenes")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = sd.blob_scene(scenes, blob).build_oracle(orc)
    K, D = scenes.scaled_K(W), scenes.D_REF
    poses = [(0.02 * f, -1.6 - 0.1 * f, 0.2, 0.01 * f, 0, 0) for f in range(nframes)]
    rows = tiling.frame_rows_of(H, stripe, rank, world)
    max_rows = max(tiling.stripe_rows(H, stripe, r, world) for r in range(world))
    local = [torch.zeros((max_rows, W * 3), dtype=torch.uint8) for _ in range(2)]
    gathered = [torch.zeros((world, max_rows, W * 3), dtype=torch.uint8) if rank == 0 else None for _ in range(2)]
    frames, state = [], {"frame": 0}

    def render_fn(b):
        img = s.render(W, H, K, D, poses[state["frame"]], planes=False)["img"].reshape(H, W * 3)
        local[b][:len(rows)] = torch.from_numpy(img[rows])
        state["frame"] += 1

    def unstripe_fn(b):
        frames.append(tiling.unstripe_host(gathered[b].numpy(), H, stripe, world).copy())

    ex = tiling.TorchExchange(rank, world)
    pipe = tiling.StripePipeline(render_fn, lambda b: ex.to_root(local[b], gathered[b], 0), unstripe_fn, assembles=rank == 0)
    pipe.stage_timing(True)
    for i in range(nframes):
        pipe.step(i)
    pipe.drain()
    assert pipe.frames_done == nframes
    # what an N > 1 bench line says about where a rank's time went (bench.multi_rank_report prints these per rank)
    st = pipe.stage_times()
    assert st["groups_timed"] == nframes and st["render_ms_per_group"] > 0 and st["exchange_ms_per_group"] > 0
    assert st["wait_for_buffer_ms"] >= 0 and st["unstripe_ms_per_group"] >= 0 and (rank != 0 or st["unstripe_ms_per_group"] > 0)
    parts = st["wait_for_buffer_ms"] + st["render_ms_per_group"] + st["exchange_ms_per_group"] + st["unstripe_ms_per_group"]
    assert parts <= st["group_span_ms"] + 1e-3
    box = [None] * world
    dist.all_gather_object(box, dict(st, rank=rank))
    assert [b["rank"] for b in box] == list(range(world))
    if rank == 0:
        ok = len(frames) == nframes
        for f in range(nframes):
            full = s.render(W, H, K, D, poses[f], planes=False)["img"].reshape(H, W * 3)
            ok = ok and np.array_equal(frames[f], full)
        np.save(out_path, np.array([int(ok), len(frames)]))
    dist.barrier()
    dist.destroy_process_group()

def _rotating_worker(rank, world, port, blob, H, W, stripe, F, ngroups, last, out_path, owner_rotation=False):
    """bench.py's default exchange: groups of F frames, frame f assembled on the rank frames_per_rank deals it to, one
    all-to-all per group; `last` = size of a shorter final group (0 = none).  Every frame has its own camera."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import orc
    import scene_defs as sd
    tiling = importlib.import_module("cuda-raytracing_amd.tiling")
    scenes = importlib.import_module("cuda-raytracing_amd.scenes")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = sd.blob_scene(scenes, blob).build_oracle(orc)
    K, D = scenes.scaled_K(W), scenes.D_REF
    sizes = [F] * ngroups + ([last] if last else [])
    nframes = sum(sizes)
    poses = [(0.02 * f, -1.6 - 0.05 * f, 0.2, 0.01 * f, 0, 0) for f in range(nframes)]
    rows = tiling.frame_rows_of(H, stripe, rank, world)
    max_rows = max(tiling.stripe_rows(H, stripe, r, world) for r in range(world))
    slots_max, counts_max, _, _ = tiling.rotating_plan(F, world)
    local = [torch.zeros((slots_max * max_rows, W * 3), dtype=torch.uint8) for _ in range(2)]
    received = [torch.zeros((world * counts_max[rank] * max_rows, W * 3), dtype=torch.uint8) for _ in range(2)]
    mine, state, group_of = {}, {"frame": 0}, [None, None]

    def render_fn(b):
        g = group_of[b]
        for f in range(g[1]):
            img = s.render(W, H, K, D, poses[g[0] + f], planes=False)["img"].reshape(H, W * 3)
            # owner rotation (bench.py's default): frame f of the group is rendered as owner (rank + f) % world
            mine_rows = tiling.frame_rows_of(H, stripe, tiling.owner_of(rank, f, world), world) if owner_rotation else rows
            local[b][f * max_rows:f * max_rows + len(mine_rows)] = torch.from_numpy(img[mine_rows])

    ex = tiling.TorchExchange(rank, world)

    def exchange_fn(b):
        ex.rotating(local[b], received[b], group_of[b][1], max_rows)

    def unstripe_fn(b):
        first, count = group_of[b]
        _, counts, offsets, real = tiling.rotating_plan(count, world)
        c = counts[rank]
        got = received[b][:world * c * max_rows].numpy().reshape(world, c, max_rows, W * 3)
        for k in range(real[rank]):
            mine[first + offsets[rank] + k] = tiling.unstripe_host(got[:, k], H, stripe, world,
                                                                   frame_index=offsets[rank] + k if owner_rotation else None).copy()

    pipe = tiling.StripePipeline(render_fn, exchange_fn, unstripe_fn)
    first = 0
    for i, c in enumerate(sizes):
        group_of[i & 1] = (first, c)
        pipe.step(i)
        first += c
    pipe.drain()
    ok = True
    for f, frame in mine.items():
        full = s.render(W, H, K, D, poses[f], planes=False)["img"].reshape(H, W * 3)
        ok = ok and np.array_equal(frame, full)
    flags = torch.tensor([int(ok), len(mine)], dtype=torch.int64)
    dist.all_reduce(flags[:1], op=dist.ReduceOp.MIN)
    dist.all_reduce(flags[1:], op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out_path, flags.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def test_two_rank_frame_pipeline(blob5k, tmp_path):
    out = str(tmp_path / "pipe.npy")
    mp.spawn(_pipeline_worker, args=(2, _free_port(), blob5k, 60, 96, 8, 5, out), nprocs=2, join=True)
    ok, n = np.load(out)
    assert ok == 1 and n == 5


@pytest.mark.parametrize("world,F,ngroups,last", [(2, 4, 2, 3), (3, 4, 1, 1), (3, 2, 2, 0)])
def test_rotating_root_pipeline(blob5k, tmp_path, world, F, ngroups, last):
    """Every frame of every group is assembled exactly once, on some rank, and equals the single-process frame --
    including groups whose frame count is not a multiple of the world size, ranks that get no frame of a group and groups
    smaller than the world (the exchange then moves padding slots: no rank ever sends or receives an empty message)."""
    out = str(tmp_path / "rot.npy")
    mp.spawn(_rotating_worker, args=(world, _free_port(), blob5k, 40, 64, 8, F, ngroups, last, out), nprocs=world, join=True)
    ok, n = np.load(out)
    assert ok == 1 and n == F * ngroups + last


@pytest.mark.parametrize("world,F,ngroups,last", [(2, 4, 1, 3), (3, 5, 2, 1)])
def test_rotating_root_pipeline_with_rotating_stripe_owner(blob5k, tmp_path, world, F, ngroups, last):
    """The same pipeline with the stripe owner rotating over the frames of a group (rt_render_stripes_batch_rotating /
    rt_unstripe_batch_rotating; here their host mirrors tiling.owner_of / unstripe_host(frame_index=)): the rank that assembles
    frame f of a group must place rank r's block as owner (r + f) % world's rows."""
    out = str(tmp_path / "rot_owner.npy")
    mp.spawn(_rotating_worker, args=(world, _free_port(), blob5k, 41, 64, 8, F, ngroups, last, out, True), nprocs=world, join=True)
    ok, n = np.load(out)
    assert ok == 1 and n == F * ngroups + last


def test_owner_rotation_gives_equal_shares():
    import importlib
    sys.path.insert(0, ROOT)
    tiling = importlib.import_module("cuda-raytracing_amd.tiling")
    for H, stripe, world in ((1080, 16, 8), (1080, 16, 4), (2160, 16, 8), (133, 16, 3), (50, 7, 5)):
        own = [tiling.stripe_rows(H, stripe, r, world) for r in range(world)]
        assert sum(own) == H
        for r in range(world):
            over_a_group = sum(tiling.stripe_rows(H, stripe, tiling.owner_of(r, f, world), world) for f in range(world))
            assert over_a_group == H                                            # every rank: one frame's worth of rows per `world` frames
        assert [tiling.owner_of(r, 0, world) for r in range(world)] == list(range(world))
    assert max(tiling.stripe_rows(1080, 16, r, 8) for r in range(8)) == 144 and min(tiling.stripe_rows(1080, 16, r, 8) for r in range(8)) == 128


@pytest.mark.parametrize("world,stripe,H", [(2, 16, 90), (2, 7, 45)])
def test_two_rank_stripes_gather(blob5k, tmp_path, world, stripe, H):
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, port, blob5k, H, 160, stripe, out), nprocs=world, join=True)
    ok, nbad = np.load(out)
    assert ok == 1, "%d bytes differ" % nbad
