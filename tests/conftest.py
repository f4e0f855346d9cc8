import importlib
import os
import sys

import pytest

# View records (librt_hip.so, launches of four and more frames) ask for eight rays per interior record of a frame before they pay for
# their pre-pass; the parity tests render small frames and want the path exercised wherever the scene allows it (read once per process)
os.environ.setdefault("RT_VIEW_MIN_RAYS", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    """The product package with both libraries built (hipcc cross-compiles without a GPU)."""
    mod = importlib.import_module("cuda-raytracing_amd")
    mod.build()
    mod.libs()
    return mod


@pytest.fixture(scope="session")
def scenes():
    return importlib.import_module("cuda-raytracing_amd.scenes")


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.build_oracle()
    return _orc


@pytest.fixture(scope="session")
def oracle(orc):
    return orc.oracle()


@pytest.fixture(scope="session")
def cache_dir():
    d = os.path.join(ROOT, ".scene_cache")
    os.makedirs(d, exist_ok=True)
    return d


@pytest.fixture(scope="session")
def blob70k(scenes, cache_dir):
    p = os.path.join(cache_dir, "blob70k.obj")
    if not os.path.exists(p):
        scenes.write_blob_obj(p, 188, 187)
    return p


@pytest.fixture(scope="session")
def blob5k(scenes, cache_dir):
    p = os.path.join(cache_dir, "blob5k.obj")
    if not os.path.exists(p):
        scenes.write_blob_obj(p, 50, 51)
    return p


@pytest.fixture(scope="session")
def atrium(scenes, cache_dir):
    p = os.path.join(cache_dir, "atrium.obj")
    if not os.path.exists(p):
        scenes.write_atrium_obj(p)
    return p


@pytest.fixture(scope="session")
def atrium_c6(scenes, cache_dir):
    """bench.py --workload c6: the atrium generator at 4 073 472 triangles (296 MB of OBJ text, written in about 15 s)."""
    p = os.path.join(cache_dir, "atrium_c6.obj")
    if not os.path.exists(p):
        scenes.write_atrium_obj(p, **scenes.C6["atrium"])
    return p


@pytest.fixture(scope="session")
def demo_objs(scenes, cache_dir):
    """bench.py --workload demo: (area, board, board with the demo's translation baked in) -- stand-ins for kernel.cu:209-210"""
    ps = tuple(os.path.join(cache_dir, n) for n in ("demo_area.obj", "demo_board.obj", "demo_board_baked.obj"))
    if not all(os.path.exists(q) for q in ps):
        scenes.write_demo_objs(ps[0], ps[1])
        scenes.write_demo_objs(ps[0], ps[2], offset=scenes.DEMO["board_pose"][:3])
    return ps
