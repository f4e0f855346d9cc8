cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gpu_b" 2>&1 | tail -3
python - <<'PY'
import importlib, sys, time
sys.path.insert(0, '.')
rt = importlib.import_module("cuda-raytracing_amd")
for name in ("blob70k", "atrium"):
    tris = rt.Mesh.load_obj(".scene_cache/%s.obj" % name).dump()["tris"]
    for gpu in (False, True, True, True):
        t = time.perf_counter(); m = rt.Mesh.from_triangles(tris, gpu_build=gpu); dt = time.perf_counter() - t
        print(name, len(tris), "tris:", "gpu " if gpu else "host", "BVH build %.1f ms" % (dt * 1e3), flush=True)
PY
