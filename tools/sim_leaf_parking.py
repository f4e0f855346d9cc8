#!/usr/bin/env python3
"""Offline study (CPU, oracle): would PARKING lanes at leaves pay?  TEST TOOLING, not product code.

render_kernel runs one loop iteration per interior node and per leaf triangle of a lane; an iteration of a wave executes the
interior block if any lane is at an interior node and the triangle block if any lane is at a triangle -- at whatever lane
count.  On c2-mid the triangle block is ~a third of the vector instructions at ~20 of 64 lanes.  A lane's sequence of steps is
fixed by the reference's visit order, but WHEN a lane takes its next step is free: a lane that arrives at a leaf could wait
until enough lanes of its wave have arrived too.  This script replays the per-ray step sequences of the oracle
(orc_trace_steps) for a sample of 8x8 tiles under
   baseline        every alive lane steps in every iteration
   park(T)         lanes at a triangle step wait unless at least T lanes are at one, or no lane is at an interior node
and prices a wave iteration as  C_fetch + C_int * [any interior lane] + C_tri * [any triangle lane]  (VALU instructions, from
the ISA of the shipped loop: 22 / 33 / 100).
   python tools/sim_leaf_parking.py [tiles] [camera]
(An analysis script like tools/fuzz_campaign.py: it reads the CHECKER's per-ray step sequences through tests/orc.py; nothing in the product imports it.)
"""
import importlib, os, sys
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
import scene_defs as sd
scenes = importlib.import_module("cuda-raytracing_amd.scenes")
C_FETCH, C_INT, C_TRI = 22, 33, 100


def lane_steps(o, s, W, H, K, D, pose, x, y, buf):
    n = o.lib.orc_trace_steps(s.h, W, H, K, D, pose, x, y, buf.ctypes.data_as(C.POINTER(C.c_uint8)), len(buf))
    seq = []
    for v in buf[:min(n, len(buf))]:
        if v == 0:
            seq.append(0)                                   # interior step
        else:
            seq.extend([1] * max(int(v) - 1, 1))            # one step per triangle of the leaf (an empty leaf takes one)
    return np.array(seq, np.int8)


def simulate(seqs, T):
    """seqs: list of per-lane step arrays (0 interior, 1 triangle).  T = 0: baseline.  -> (iterations, with interior, with triangle,
    lane-steps)"""
    pos = np.zeros(len(seqs), np.int64)
    lens = np.array([len(q) for q in seqs])
    it = n_int = n_tri = 0
    while True:
        alive = pos < lens
        if not alive.any():
            break
        kind = np.array([seqs[l][pos[l]] if alive[l] else -1 for l in range(len(seqs))])
        at_int, at_tri = kind == 0, kind == 1
        run_tri = at_tri.any() and (T == 0 or not at_int.any() or at_tri.sum() >= T)
        step = at_int | (at_tri & run_tri)
        pos[step] += 1
        it += 1
        n_int += bool(at_int.any())
        n_tri += bool(run_tri)
    return it, n_int, n_tri, int(lens.sum())


def main():
    ntiles = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    cam = sys.argv[2] if len(sys.argv) > 2 else "mid"
    orc.build_oracle()
    o = orc.oracle()
    o.lib.orc_trace_steps.restype = C.c_int
    o.lib.orc_trace_steps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                      C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int]
    blob = os.path.join(ROOT, ".scene_cache", "blob70k.obj")
    if not os.path.exists(blob):
        scenes.write_blob_obj(blob, 188, 187)
    s = sd.blob_scene(scenes, blob).build_oracle(orc)
    W, H = 1920, 1080
    fp = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))
    Ka, Da, Pa = (np.ascontiguousarray(v, np.float32) for v in (scenes.scaled_K(W), scenes.D_REF, scenes.C2_CAMERAS[cam]))
    rng = np.random.default_rng(3)
    tiles = rng.choice((W // 8) * (H // 8), ntiles, replace=False)
    buf = np.zeros(4096, np.uint8)
    waves = []
    for t in tiles:
        tx, ty = int(t) % (W // 8), int(t) // (W // 8)
        waves.append([lane_steps(o, s, W, H, fp(Ka), fp(Da), fp(Pa), tx * 8 + (l & 7), ty * 8 + (l >> 3), buf) for l in range(64)])
    print("camera %s, %d tiles of 8x8 pixels" % (cam, ntiles))
    base = None
    for T in (0, 8, 16, 24, 32, 48):
        tot = np.zeros(4, np.int64)
        for w in waves:
            tot += np.array(simulate(w, T))
        cost = C_FETCH * tot[0] + C_INT * tot[1] + C_TRI * tot[2]
        base = base or cost
        print("%-9s wave iterations %7d  with interior %7d  with triangle %7d  lanes/iteration %.1f  VALU cost %.3f of baseline"
              % ("baseline" if T == 0 else "park(%d)" % T, tot[0], tot[1], tot[2], tot[3] / tot[0], cost / base))


if __name__ == "__main__":
    main()
