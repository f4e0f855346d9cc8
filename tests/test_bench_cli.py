"""bench.py's launcher logic without a GPU: a bare `python bench.py --gpus N` must start its ranks itself, and when they
cannot run (this container has no GPU) it must fail loudly -- non-zero exit, no result line -- rather than print a number."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=600, env=env)


def test_bare_multi_gpu_run_launches_ranks_and_fails_without_gpus():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this check is for the GPU-less container")
    r = _run("--gpus", "2", "--debug-backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "launching 2 ranks" in r.stderr and "needs a GPU" in r.stderr
    r = _run("--gpus", "2", "--steps", "2")                    # RCCL backend: refuses before launching anything
    assert r.returncode != 0 and r.stdout.strip() == "" and "GPU(s) visible" in r.stderr


def test_single_gpu_run_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this check is for the GPU-less container")
    r = _run("--steps", "2", "--warmup", "1")
    assert r.returncode != 0 and r.stdout.strip() == "" and "needs a GPU" in r.stderr


def test_timed_region_aggregation():
    """The one driver-timed number: a K-step region that fits one launch is measured nine times and the line is priced with
    the median (bench.aggregate_repeats); a longer run stays one measurement."""
    sys.path.insert(0, ROOT)
    import bench
    a = bench.aggregate_repeats([0.0030, 0.0027, 0.0100, 0.0026, 0.0028, 0.0027, 0.0029, 0.0027, 0.0031], 20)
    assert a["dt"] == 0.0028 and a["fields"]["repeats"] == 9
    assert a["fields"]["ms_per_step_min"] == 0.13 and a["fields"]["ms_per_step_max"] == 0.5
    assert "median of 9" in a["fields"]["timed_region"]
    b = bench.aggregate_repeats([0.5], 960)
    assert b["dt"] == 0.5 and b["fields"] == {"repeats": 1}
    c = bench.aggregate_repeats([0.004, 0.002], 10)                     # (an even count: mean of the middle two)
    assert abs(c["dt"] - 0.003) < 1e-12
    assert bench.REPEATS_SHORT == 9


def test_multi_rank_report_fields():
    """What an N > 1 result line says about its ranks (bench.multi_rank_report): the per-rank stage times as given, the
    stripe-share imbalance = max / mean of the ranks' render time, and the RCCL facts that can be known."""
    sys.path.insert(0, ROOT)
    import bench
    ranks = [dict(rank=0, groups_timed=4, wait_for_buffer_ms=0.1, render_ms_per_group=0.6, exchange_ms_per_group=0.2, unstripe_ms_per_group=0.05, group_span_ms=0.95),
             dict(rank=1, groups_timed=4, wait_for_buffer_ms=0.0, render_ms_per_group=0.4, exchange_ms_per_group=0.3, unstripe_ms_per_group=0.0, group_span_ms=0.7)]
    r = bench.multi_rank_report(ranks, None)
    assert r["per_rank"] == ranks and r["stripe_share_imbalance"] == 1.2
    assert set(r["rccl"]) >= {"version", "through", "NCCL_MAX_NCHANNELS"} and r["rccl"]["through"] == "torch.distributed"


def test_scaling_prediction_fields():
    """bench.py --predict-scaling: one N of the prediction from the one-GPU time and the virtual ranks' times (the slowest rank
    sets the pace of a real run)."""
    sys.path.insert(0, ROOT)
    import bench
    ranks = [0.50, 0.45, 0.47, 0.46, 0.45, 0.46, 0.47, 0.44]
    p = bench.scaling_prediction(3.6, ranks)
    assert p["n_gpus"] == 8 and p["one_gpu_ms"] == 3.6 and p["ideal_ms"] == 0.45 and p["render_ms_per_rank"] == ranks
    assert p["predicted_render_speedup"] == 7.2 and p["predicted_render_efficiency"] == 0.9
    assert p["stripe_share_imbalance"] == round(0.50 / (sum(ranks) / 8), 4)
    q = bench.scaling_prediction(2.0, [1.0, 1.0])
    assert q["predicted_render_speedup"] == 2.0 and q["stripe_share_imbalance"] == 1.0


def test_committed_scaling_prediction():
    """profiles/r06_predicted_scaling.json = `python bench.py --predict-scaling 2,4,8` on one MI355X with this round's kernels:
    every virtual rank of N timed through the entry points a real rank uses.  north_star asks for >= 7 x at 8 GPUs; the render
    side of the STREAM must leave room for the exchange.  The driver's shape (one 20-frame group per timed region) is rendered as
    a pipeline of sub-groups with the heavy-first order; its render side is bounded by the fixed cost of a region."""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_predicted_scaling.json")))
    assert d["mode"] == "predict-scaling" and d["n_gpus"] == 1 and len(d["code_hash"]) == 16 and d["stripe_owner_rotates_over_frames"] is True
    per_n = {p["n_gpus"]: p for p in d["stream"]["per_n"]}
    assert sorted(per_n) == [2, 4, 8]
    for n, p in per_n.items():
        assert len(p["render_ms_per_rank"]) == n and all(v > 0 for v in p["render_ms_per_rank"])
        assert abs(p["predicted_render_speedup"] - p["one_gpu_ms"] / max(p["render_ms_per_rank"])) < 2e-3
        assert 1.0 <= p["stripe_share_imbalance"] < 1.02                       # the stripe owner rotates over the frames: equal shares
        assert p["predicted_render_speedup"] <= n
    assert per_n[8]["predicted_render_speedup"] >= 7.0
    shape = {p["n_gpus"]: p for p in d["stream"]["driver_shape_per_n"]}
    one = {p["n_gpus"]: p for p in d["stream"]["driver_shape_one_launch_per_n"]}
    assert sorted(shape) == sorted(one) == [2, 4, 8]
    assert shape[8]["sub_groups"] == [[0, 10], [10, 10]] and shape[4]["sub_groups"] == shape[2]["sub_groups"] == [[0, 5], [5, 5], [10, 5], [15, 5]]
    # round 5's one natural-order launch predicted 6.20 x at N = 8: the heavy-first order of thin striped launches must have moved that
    assert one[8]["predicted_render_speedup"] >= 6.35 and shape[8]["predicted_render_speedup"] >= 6.0
    assert shape[8]["stripe_share_imbalance"] < 1.03


def _barrier_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(world))
    import time
    import numpy as np
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = bench.NodeBarrier(rank, world, on_cpu=True)
    assert b.slots is not None and "shared-memory" in b.kind
    # every round: one rank is late; nobody may leave the barrier before it has arrived
    log = []
    for rnd in range(20):
        if rank == rnd % world:
            time.sleep(0.01)
        arrived = time.monotonic()
        b.wait()
        log.append((arrived, time.monotonic()))
    np.save(os.path.join(out_dir, "b%d.npy" % rank), np.asarray(log))
    dist.barrier()
    b.close()
    dist.destroy_process_group()


def test_node_barrier_is_a_barrier(tmp_path):
    """bench.NodeBarrier (the shared-memory rendezvous that brackets the timed region of an N-rank run on one node): in every round
    every rank leaves the barrier after the LAST rank has arrived at it."""
    import socket
    import numpy as np
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    world = 3
    mp.spawn(_barrier_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    logs = np.stack([np.load(str(tmp_path / ("b%d.npy" % r))) for r in range(world)])     # [rank][round][arrived, left]
    last_arrival = logs[:, :, 0].max(axis=0)
    assert (logs[:, :, 1] >= last_arrival[None, :]).all()
    assert not any(n.startswith("rt_bench_barrier_") for n in os.listdir("/dev/shm"))      # rank 0 removed the page


def test_sub_groups_of_a_single_group_region():
    """tiling.sub_groups: how bench.py cuts the ONE group of a short timed region (the driver's `--steps 20`) into the pipeline steps
    of an N-rank run: every frame in exactly one sub-group, in order, at most four, as even as possible, none smaller than the rank
    count (a rotating-root exchange moves at least one frame slot per rank)."""
    import importlib
    sys.path.insert(0, ROOT)
    tiling = importlib.import_module("cuda-raytracing_amd.tiling")
    assert tiling.sub_groups(20, 8) == [(0, 10), (10, 10)]
    assert tiling.sub_groups(20, 4) == tiling.sub_groups(20, 2) == tiling.sub_groups(20, 1) == [(0, 5), (5, 5), (10, 5), (15, 5)]
    assert tiling.sub_groups(5, 8) == [(0, 5)] and tiling.sub_groups(1, 1) == [(0, 1)]
    for count in range(1, 33):
        for world in (1, 2, 3, 4, 6, 8):
            for parts in (1, 2, 4):
                sub = tiling.sub_groups(count, world, parts)
                assert 1 <= len(sub) <= parts and sum(c for _, c in sub) == count
                assert [f for f, _ in sub] == [sum(c for _, c in sub[:k]) for k in range(len(sub))]
                assert len(sub) == 1 or all(c >= world for _, c in sub)
                assert max(c for _, c in sub) - min(c for _, c in sub) <= 1
                # the exchange plan of every sub-group gives every rank something to send and to receive
                for _, c in sub:
                    slots, counts, offsets, real = tiling.rotating_plan(c, world)
                    assert slots >= world and all(n >= 1 for n in counts) and sum(real) == c
