#!/usr/bin/env python3
"""Freezes the ORACLE's own per-pixel planes into tests/golden/pins.json ("oracle_planes_self_regression").

These hashes are NOT reference-derived: they are what oracle/rt_oracle.c produced when this script was last run, kept so
that the oracle and the HIP kernels (which are compared with each other on every run) cannot drift together unnoticed.
The reference-derived numbers for the same frames are the FNV frame hashes and the per-ray statistics of SURVEY.md
(pins.json "frames" / "survey_c2_stats"), asserted next to them in tests/test_oracle_pins.py.

    python tests/golden/make_plane_pins.py          # rewrites the section in place
"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
PLANES = ("hit_inst", "hit_tri", "pops", "aabb", "tris", "inside")


def plane_hashes(out):
    return {k: hashlib.sha256(np.ascontiguousarray(out[k], np.int32).tobytes()).hexdigest() for k in PLANES}


def cases(scenes, sd, blob70k, blob5k):
    """name -> (SceneDesc, width, height, K, pose)"""
    c1, c2 = scenes.C1, scenes.C2
    out = {"C1_256x256": (sd.c1_scene(scenes), c1["width"], c1["height"], c1["K"], c1["cam_pose"])}
    for cam, pose in scenes.C2_CAMERAS.items():
        out["C2_%s_1920x1080" % cam] = (sd.blob_scene(scenes, blob70k), c2["width"], c2["height"], scenes.scaled_K(c2["width"]), pose)
    m = sd.MULTI_CAMERA
    out["multi_instance_textured_320x200"] = (sd.multi_instance_scene(scenes, blob5k), m["width"], m["height"], scenes.scaled_K(m["width"]), m["pose"])
    return out


def main():
    import orc
    import scene_defs as sd
    scenes = importlib.import_module("cuda-raytracing_amd.scenes")
    orc.build_oracle()
    cache = os.path.join(ROOT, ".scene_cache")
    os.makedirs(cache, exist_ok=True)
    b70, b5 = os.path.join(cache, "blob70k.obj"), os.path.join(cache, "blob5k.obj")
    if not os.path.exists(b70):
        scenes.write_blob_obj(b70, 188, 187)
    if not os.path.exists(b5):
        scenes.write_blob_obj(b5, 50, 51)
    pins = {}
    for name, (desc, W, H, K, pose) in cases(scenes, sd, b70, b5).items():
        s = desc.build_oracle(orc)
        out = s.render(W, H, K, scenes.D_REF, pose, threads=os.cpu_count() or 1)
        pins[name] = dict(plane_hashes(out), img_sha256=hashlib.sha256(out["img"].tobytes()).hexdigest())
        s.close()
        print(name, pins[name]["hit_tri"][:16])
    path = os.path.join(HERE, "pins.json")
    doc = json.load(open(path))
    doc["oracle_planes_self_regression"] = {
        "note": "sha256 of the oracle's own int32 planes (and RGB bytes); written by tests/golden/make_plane_pins.py; NOT reference-derived",
        "frames": pins}
    json.dump(doc, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
