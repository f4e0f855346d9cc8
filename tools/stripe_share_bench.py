#!/usr/bin/env python3
"""What one rank of an N-GPU run does between collectives, timed on one GPU: rank 0's stripes of F frames per launch,
launches issued back to back on one stream or alternating between two.   python tools/stripe_share_bench.py [N] [F] [stripe_rows] [rank,rank,...]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rt = importlib.import_module("cuda-raytracing_amd")
scenes = importlib.import_module("cuda-raytracing_amd.scenes")
tiling = importlib.import_module("cuda-raytracing_amd.tiling")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
W, H = 1920, 1080
stripe = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ranks = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
mesh = rt.Mesh.load_obj(os.path.join(ROOT, ".scene_cache", "blob70k.obj"))
scene = rt.Scene(); scene.add_material(scenes.C2["albedo"]); scene.add_mesh(mesh); scene.add_mesh_instance(0, 0); scene.upload_to_device()
max_rows = max(tiling.stripe_rows(H, stripe, r, N) for r in range(N))
pitch = W * 3
for cname, pose in [(c, p) for c, p in scenes.C2_CAMERAS.items() for _ in ranks]:
    rank = ranks.pop(0); ranks.append(rank)
    out = []
    for nstreams in (1, 2):
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        bufs = [torch.zeros((F * max_rows, pitch), dtype=torch.uint8, device="cuda") for _ in range(nstreams)]
        calls, cams = [], []
        for s, b in zip(streams, bufs):
            cam = rt.Camera(W, H, scenes.scaled_K(W), scenes.D_REF); cam.set_pose(pose); cam.set_stream(s.cuda_stream); cams.append(cam)
            calls.append(cam.prepared_batch(scene, [pose] * F, tiling.batch_local_ptrs(b.data_ptr(), F, max_rows, pitch), pitch, stripes=(stripe, rank, N)))
        for i in range(8): calls[i % nstreams]()
        torch.cuda.synchronize()
        L = 48
        t0 = time.perf_counter()
        for i in range(L): calls[i % nstreams]()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out.append(dt / L * 1e3)
    ideal = {"far": 0.0724, "mid": 0.1338, "near": 0.0936}[cname] * F / N
    print("%-5s rank %d of N=%d F=%d: ms per launch, 1 stream %.3f, 2 streams %.3f (1/N of the one-GPU kernel time: %.3f)" % (cname, rank, N, F, out[0], out[1], ideal), flush=True)
