cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python - <<'PY'
import importlib, sys
sys.path.insert(0, '.')
rt = importlib.import_module("cuda-raytracing_amd"); scenes = importlib.import_module("cuda-raytracing_amd.scenes")
import ctypes as C
m = rt.Mesh.load_obj(".scene_cache/blob70k.obj"); s = rt.Scene(); s.add_material(scenes.C2["albedo"]); s.add_mesh(m); s.add_mesh_instance(0,0); s.upload_to_device()
cam = rt.Camera(1920,1080,scenes.scaled_K(1920),scenes.D_REF); cam.set_pose(scenes.C2_CAMERAS["mid"]); cam.set_options(4, 2, 1)
img = rt.DeviceBuffer(width_bytes=1920*3, height=1080); cam.render_scene(s, img.ptr, img.pitch, synchronize=True)
rt.check(rt.libs()[1].rth_save_png(b"gpurun_out/c2_mid_lit.png", img.ptr, 1920, 1080, img.pitch)); print("png written")
PY
